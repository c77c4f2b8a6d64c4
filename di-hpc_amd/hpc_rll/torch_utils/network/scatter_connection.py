"""``hpc_rll.torch_utils.network.scatter_connection`` -- drop-in for the reference module of the same path
(/root/reference/hpc_rll/torch_utils/network/scatter_connection.py:7-87): ``ScatterConnection(B, M, N, H, W,
scatter_type)`` and ``forward(x, location)``.

``cover`` is deterministic here (the largest entity index at a cell wins, which is what the CPU oracle's
sequential ``scatter_`` does); the reference kernel is a last-writer-wins race.
The autograd node is ``hpc_torch_utils_network.scatter_connection`` (compiled torch::autograd::Function)."""
import torch

import hpc_torch_utils_network


class ScatterConnection(torch.nn.Module):
    """Scatter entity embeddings x (B,M,N) to their (y,x) cell of a (B,N,H,W) feature map."""

    def __init__(self, B, M, N, H, W, scatter_type) -> None:
        super().__init__()
        self.B, self.M, self.N, self.H, self.W = B, M, N, H, W
        self.scatter_type = scatter_type
        assert self.scatter_type in ['cover', 'add']

    def forward(self, x: torch.Tensor, location: torch.Tensor) -> torch.Tensor:
        assert x.is_cuda
        assert location.is_cuda
        return hpc_torch_utils_network.scatter_connection(x, location, self.H, self.W, self.scatter_type)
