"""CPU tier for the compiled extension modules (no GPU needed): every L2 function of the reference's pybind def list is
bound with the reference's name, rejects lists of the wrong length with a message that names the accepted forms, and
rejects host tensors loudly (there is no CPU / eager fallback) -- in the native AND in the reference's positional form."""
import pytest
import torch


def Z(*s, dtype=torch.float32):
    return torch.zeros(*s, dtype=dtype)


T, B, N = 3, 4, 5
A2 = Z(T, B, dtype=torch.int64)
A1 = Z(B, dtype=torch.int64)


def _cases():
    import hpc_rl_utils as U
    import hpc_torch_utils_network as NW
    import hpc_models as M
    vt_in = [Z(T, B, N), Z(T, B, N), A2, Z(T + 1, B), Z(T, B), None]
    ppo_in = [Z(B, N), Z(B, N), A1, Z(B), Z(B), Z(B), Z(B), None]
    q_in = [Z(B, N), Z(B, N), A1, A1, Z(2, B), Z(B), None]
    lstm_in = [Z(2, B, 3), Z(1, B, 2), Z(1, B, 2), Z(3 * 8), Z(2 * 8), Z(8), Z(1, 16), Z(1, 16)]
    return [
        (U.GaeForward, ([Z(T + 1, B), Z(T, B)], [Z(T, B)], 0.99, 0.97)),
        (U.GaeBackward, ([Z(T, B)], [Z(T + 1, B), None], 0.99, 0.97)),
        (U.TdLambdaForward, ([Z(T + 1, B), Z(T, B), None], [Z(1), Z(T, B)], 0.9, 0.8)),
        (U.TdLambdaBackward, ([Z(1), Z(T, B)], [Z(T + 1, B)])),
        (U.VTraceForward, (vt_in, [Z(3), Z(10)], 0.99, 0.95, 1.0, 1.0, 1.0)),                       # native
        (U.VTraceForward, (vt_in, [Z(T, B)] * 2 + [Z(T, B, N)] * 3 + [Z(T, B)] * 4 + [Z(1)] * 3, 0.99, 0.95, 1.0, 1.0, 1.0)),  # reference
        (U.VTraceBackward, ([Z(1), Z(1), Z(1), Z(T, B, N), A2, Z(10)], [Z(T, B, N), Z(T + 1, B)])),
        (U.UpgoForward, ([Z(T, B, N), Z(T, B), A2, Z(T, B), Z(T + 1, B)], [Z(1), Z(10)])),
        (U.UpgoForward, ([Z(T, B, N), Z(T, B), A2, Z(T, B), Z(T + 1, B)], [Z(T, B), Z(T, B), Z(1), Z(T, B, N)])),
        (U.UpgoBackward, ([Z(1), Z(T, B, N), A2, Z(10)], [Z(T, B, N)])),
        (U.PPOForward, (ppo_in, [Z(5), Z(10)], True, 0.2, 0.0)),
        (U.PPOForward, (ppo_in, [Z(B)] * 2 + [Z(B, N)] * 3 + [Z(B)] * 4 + [Z(1)] * 5, True, 0.2, 0.0)),
        (U.PPOBackward, ([Z(1), Z(1), Z(1), Z(B, N), A1, Z(10)], [Z(B, N), Z(B)])),
        (U.QNStepTdForward, (q_in, [Z(B), Z(1), Z(B)], 0.9)),
        (U.QNStepTdBackward, ([Z(1), Z(B), A1], [Z(B, N)])),
        (U.QNStepTdRescaleForward, (q_in, [Z(B), Z(1), Z(B)], 0.9)),
        (U.QNStepTdRescaleBackward, ([Z(1), Z(B), A1], [Z(B, N)])),
        (U.DistNStepTdForward, ([Z(B, N, 7), Z(B, N, 7), A1, A1, Z(2, B), Z(B), None], [Z(B), Z(1), Z(B + B * 7)], 0.9, -1.0, 1.0)),
        (U.DistNStepTdBackward, ([Z(1), Z(B, 7), A1], [Z(B, N, 7)])),
        (U.IQNNStepTDErrorForward, ([Z(6, B, N), Z(7, B, N), A1, A1, Z(2, B), Z(B), Z(6, B), None, None], [Z(1), Z(B), Z(B, 6)], 0.9, 1.0)),
        (U.IQNNStepTDErrorBackward, ([Z(1), Z(B, 6), Z(B), A1], [Z(6, B, N)])),                    # reference: with `weight`
        (U.QRDQNNStepTDErrorForward, ([Z(B, N, 6), Z(B, N, 6), A1, A1, Z(2, B), Z(B), None, None], [Z(1), Z(B)] + [Z(B, 6, 6)] * 2 + [Z(B, 6)], 0.9)),
        (U.QRDQNNStepTDErrorBackward, ([Z(1), Z(B, 6), A1], [Z(B, N, 6)])),
        (U.Pad1DForward, ([Z(3), Z(4)], 0)),
        (U.Pad2DForward, ([Z(3, 2), Z(4, 1)], 0)),
        (U.Pad3DForward, ([Z(3, 2, 2), Z(4, 1, 1)], 0)),
        (U.GroupPad1DForward, ([Z(3), Z(4)], [2], [4], [0, 0], [0, 2], 0)),
        (U.Unpad1DForward, (Z(2, 4), [3, 4])),
        (U.Unpad2DForward, (Z(2, 4, 2), [3, 2, 4, 1])),
        (U.Unpad3DForward, (Z(2, 4, 2, 2), [3, 2, 2, 4, 1, 1])),
        (U.pad1d_packed, (Z(7), torch.tensor([3, 4]), 4, 0)),
        (U.gae, (Z(T + 1, B), Z(T, B))),
        (U.td_lambda, (Z(T + 1, B), Z(T, B))),
        (U.vtrace, tuple(vt_in[:5])),
        (U.ppo, tuple(ppo_in[:7])),
        (NW.LstmForward, (lstm_in, [Z(2, B, 2), Z(1, B, 2), Z(1, B, 2), Z(10)], 0.0)),
        (NW.LstmForward, (lstm_in, [Z(2, B, 8), Z(B, 8), Z(2, 1, B, 2), Z(2, 1, B, 2), Z(1, 2, B, 8), Z(1, 2, B, 2), Z(1), Z(1), Z(1), Z(1)], 0.0)),
        (NW.ScatterConnectionForward, ([Z(2, 3, 4), Z(2, 3, 2, dtype=torch.int64)], [Z(2, 4, 5, 5)], "add")),
        (NW.ScatterConnectionBackward, ([Z(2, 4, 5, 5), Z(2, 3, 2, dtype=torch.int64)], [Z(2, 3, 4)])),
        (NW.lstm, tuple(lstm_in[:1] + lstm_in[3:8] + lstm_in[1:3])),
        (NW.scatter_connection, (Z(2, 3, 4), Z(2, 3, 2, dtype=torch.int64), 5, 5, "cover")),
        (M.actor_critic_update_ae, ([Z(2, 3, 4), Z(2, dtype=torch.int64), Z(2, dtype=torch.int64)], [Z(2, 4)])),
        (M.actor_critic_lstm_activation, ([Z(2, 8), Z(2, 8), Z(8)], [Z(2, 2), Z(2, 2)])),
        (M.actor_critic_pre_sample, ([Z(2, 3, 4), Z(1, 2, 4), Z(2, 3, dtype=torch.bool)], [Z(2, 3)])),
    ]


def test_every_entry_point_rejects_host_tensors():
    for fn, args in _cases():
        with pytest.raises(RuntimeError, match="GPU"):
            fn(*args)


def test_wrong_list_lengths_name_the_accepted_forms():
    import hpc_rl_utils as U
    import hpc_torch_utils_network as NW
    with pytest.raises(RuntimeError, match=r"2 \(native\) or 12 \(reference\)"):
        U.VTraceForward([Z(1)] * 6, [Z(1)] * 3, 0.99, 0.95, 1.0, 1.0, 1.0)
    with pytest.raises(RuntimeError, match=r"6 \(native\) or 9 \(reference\)"):
        U.PPOBackward([Z(1)] * 7, [None, None])
    with pytest.raises(RuntimeError, match=r"4 \(native\) or 10 \(reference\)"):
        NW.LstmForward([Z(1)] * 8, [Z(1)] * 5, 0.0)
    with pytest.raises(RuntimeError, match="expected 2 tensors"):
        U.GaeForward([Z(1)], [Z(1)], 0.99, 0.97)
    with pytest.raises(RuntimeError, match="scatter_type"):
        NW.scatter_connection(Z(1, 1, 1), Z(1, 1, 2, dtype=torch.int64), 2, 2, "max")


def test_reference_backward_without_forward_is_an_error_not_a_crash():
    """The reference-convention backward looks its state up under a module buffer; an unknown buffer is a RuntimeError."""
    import hpc_rl_utils as U
    with pytest.raises(RuntimeError, match="no forward state"):
        U.UpgoBackward([Z(1), Z(T, B, N), Z(T, B)], [Z(T, B, N)])
