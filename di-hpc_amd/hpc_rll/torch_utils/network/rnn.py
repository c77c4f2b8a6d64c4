"""``hpc_rll.torch_utils.network.rnn`` -- drop-in for /root/reference/hpc_rll/torch_utils/network/rnn.py:10-183:
``LSTM(seq_len, batch_size, input_size, hidden_size, num_layers=1, norm_type='LN', dropout=0.)`` with the same flat
parameters (``wx``, ``wh``, ``bias``, ``ln_gamma``, ``ln_beta``; layouts rnn.py:107-115) and
``forward(inputs, prev_state) -> (output, [h, c])``.

Differences: gradients flowing in through the returned final states are propagated (the reference
zeroes them, lstm.cu:309-310); scratch lives in one workspace tensor allocated per call instead of ~20 module
buffers (one of them mis-sized, rnn.py:130); the dropout mask is a stateless hash of (seed, layer, element).
The autograd node is ``hpc_torch_utils_network.lstm`` (compiled torch::autograd::Function).
"""
import math

import torch
import torch.nn as nn

import hpc_torch_utils_network


class LSTM(nn.Module):
    """Multi-layer LSTM with LayerNorm on both gate pre-activations (gate order i, f, o, u)."""

    def __init__(self, seq_len, batch_size, input_size, hidden_size, num_layers=1, norm_type='LN', dropout=0.):
        super().__init__()
        assert norm_type in ['LN']
        self.seq_len = seq_len
        self.batch_size = batch_size
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.norm_type = norm_type
        self.dropout = dropout

        gain = math.sqrt(1. / self.hidden_size)
        dims = [input_size] + [hidden_size] * num_layers
        wx = torch.cat([torch.empty(dims[l] * hidden_size * 4).uniform_(-gain, gain) for l in range(num_layers)])
        wh = torch.cat([torch.empty(hidden_size * hidden_size * 4).uniform_(-gain, gain) for _ in range(num_layers)])
        bias = torch.cat([torch.empty(hidden_size * 4).uniform_(-gain, gain) for _ in range(num_layers)])
        self.register_parameter('wx', nn.Parameter(wx))
        self.register_parameter('wh', nn.Parameter(wh))
        self.register_parameter('bias', nn.Parameter(bias))
        self.register_parameter('ln_gamma', nn.Parameter(torch.ones(num_layers, hidden_size * 4 * 2)))
        self.register_parameter('ln_beta', nn.Parameter(torch.zeros(num_layers, hidden_size * 4 * 2)))

    def forward(self, inputs, prev_state):
        """inputs (S,B,input_size); prev_state None or (h0, c0) each (num_layers,B,H) -> (y (S,B,H), [h, c])."""
        assert inputs.is_cuda
        if prev_state is None:
            zeros = torch.zeros(self.num_layers, inputs.shape[1], self.hidden_size, dtype=inputs.dtype,
                                device=inputs.device)
            prev_state = (zeros, zeros)
        h0, c0 = prev_state
        assert h0.is_cuda
        assert c0.is_cuda
        p = self.dropout if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0
        y, h, c = hpc_torch_utils_network.lstm(inputs, self.wx, self.wh, self.bias, self.ln_gamma, self.ln_beta, h0, c0,
                                               p, seed)
        return y, [h, c]
