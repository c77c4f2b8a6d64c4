"""``hpc_rll.torch_utils.network.rnn`` -- drop-in for /root/reference/hpc_rll/torch_utils/network/rnn.py:10-183:
``LSTM(seq_len, batch_size, input_size, hidden_size, num_layers=1, norm_type='LN', dropout=0.)`` with the same flat
parameters (``wx``, ``wh``, ``bias``, ``ln_gamma``, ``ln_beta``; layouts rnn.py:107-115) and
``forward(inputs, prev_state) -> (output, [h, c])``.

Differences: gradients flowing in through the returned final states are propagated (the reference
zeroes them, lstm.cu:309-310); scratch lives in one workspace tensor allocated per call instead of ~20 module
buffers (one of them mis-sized, rnn.py:130) -- a checkpoint saved by the reference module still loads (its scratch
buffers are dropped), a checkpoint saved here holds the five parameters only; the dropout mask is a stateless hash of
(seed, layer, element), and dropout is applied in training mode only (the reference applies it whenever ``dropout``
> 0, also under ``eval()``).
The autograd node is ``hpc_torch_utils_network.lstm`` (compiled torch::autograd::Function).

Three batch regimes run PERSISTENT kernels whose workgroups exchange data and must be co-resident on the GPU: B <= 4
(per-layer kernels / layer wavefront), 5 <= B <= 256 (mid-batch kernels, ``csrc/lstm_mid.hpp``) and B >= 4096 (row-block
kernels, ``csrc/lstm_block.hpp``).  If ANOTHER PROCESS holds the compute units for seconds, such a kernel gives up and
reports it asynchronously: the next LSTM call of this process warns, re-runs itself on the step kernels and keeps using
them; results produced in between are invalid.  On GPUs shared between processes (several actors per GPU) either
construct the module with ``check_persistent=True`` (synchronises after every forward that RAN a persistent kernel --
whatever the batch size --, checks, and recomputes on the spot: an actor reads its outputs right away anyhow) or export
``HPC_RLL_LSTM_PERSIST=0`` (step kernels from the start).
"""
import math
import os
import warnings

import torch
import torch.nn as nn

import hpc_torch_utils_network

if os.environ.get("HPC_RLL_LSTM_PERSIST") == "0":      # GPUs shared between processes: no co-residency assumptions
    hpc_torch_utils_network.tune_set(3, 0)

# scratch buffers the REFERENCE module registers (rnn.py:117-141) and therefore writes into its state_dict
_REFERENCE_SCRATCH = ('xbuf', 'hbuf', 'ifog', 'hn', 'cn', 'ym', 'ln_in', 'ln_mean', 'ln_rstd', 'dropout_mask', 'dgate', 'dx',
                      'dwx', 'dwh', 'dbias', 'd_ln_gamma', 'd_ln_beta')


class LSTM(nn.Module):
    """Multi-layer LSTM with LayerNorm on both gate pre-activations (gate order i, f, o, u)."""

    def __init__(self, seq_len, batch_size, input_size, hidden_size, num_layers=1, norm_type='LN', dropout=0., *,
                 check_persistent=False):
        super().__init__()
        assert norm_type in ['LN']
        self.check_persistent = check_persistent
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
        """inputs (S,B,input_size); prev_state None or (h0, c0) each (num_layers,B,H) -> (y (S,B,H), [h, c]).
        y is always the caller's own (S,B,H) tensor.  When a graph is recorded it is also the last layer's saved h
        sequence (written by the cells directly, no (S,B,H) copy): like the output of ``exp`` or ``sigmoid`` it may be
        modified in place only if no backward through this LSTM call follows (autograd's version counter raises
        otherwise); holding ``y.detach()`` keeps S*B*H floats alive, not the op's workspace."""
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
        args = (inputs, self.wx, self.wh, self.bias, self.ln_gamma, self.ln_beta, h0, c0, p, seed)
        y, h, c = hpc_torch_utils_network.lstm(*args)
        # paths 1, 2, 4, 5 of hpc_rll_lstm_last_forward_path (this thread's own call): kernels that need co-residency
        if self.check_persistent and hpc_torch_utils_network.lstm_last_forward_path() in (1, 2, 4, 5):
            # the persistent kernels report a co-residency timeout asynchronously: wait for THIS call and look
            torch.cuda.current_stream(inputs.device).synchronize()
            if hpc_torch_utils_network.async_error():
                hpc_torch_utils_network.clear_async_error()
                warnings.warn("hpc_rll LSTM: a persistent kernel timed out waiting for co-residency (another process "
                              "holds the GPU); this forward is recomputed on the step kernels, which are used from now on",
                              RuntimeWarning)
                y, h, c = hpc_torch_utils_network.lstm(*args)
        return y, [h, c]

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """Accept a checkpoint written by the reference module: drop its scratch buffers."""
        for name in _REFERENCE_SCRATCH:
            state_dict.pop(prefix + name, None)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
