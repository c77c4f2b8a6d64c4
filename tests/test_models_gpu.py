"""GPU parity tests for the hpc_models actor-critic helpers.  The reference has no `origin` module for
these; its test (tests/test_actor_critic.py) validates against inline torch expressions and torch.nn.LSTM -- the same
expressions are the oracle here (update_ae :23-26, nn.LSTM :124-154, pre_sample :259-264 with its own
np.allclose(rtol=1e-5, atol=1e-5) assert :273)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("B,E,D", [(8, 512, 256), (3, 7, 5), (100, 33, 1000)])
def test_update_ae(B, E, D):
    import hpc_models
    g = torch.Generator().manual_seed(B)
    key = torch.randn(B, E, D, generator=g)
    num = torch.randint(max(E - 2, 1), E, (B,), generator=g)
    sample = torch.stack([torch.randint(0, int(n) + 1, (1,), generator=g)[0] for n in num])   # may equal entity_num ("end")
    ae = torch.randn(B, D, generator=g)
    end = sample == num
    ref = ae + key[torch.arange(B), sample.clamp(max=E - 1)] * (~end).unsqueeze(1)           # tests/test_actor_critic.py:25
    d_ae = ae.to(DEV)
    hpc_models.actor_critic_update_ae([key.to(DEV), sample.to(DEV), num.to(DEV)], [d_ae])
    assert torch.equal(d_ae.cpu(), ref)                                                       # one fp32 add: bit exact


@pytest.mark.parametrize("B,I,H", [(8, 384, 384), (5, 12, 70), (64, 100, 1024)])
def test_lstm_activation(B, I, H):
    import hpc_models
    torch.manual_seed(B + H)
    lstm = torch.nn.LSTM(I, H, 1)
    x, h0, c0 = torch.randn(1, B, I), torch.randn(1, B, H), torch.randn(1, B, H)
    out, (hn, cn) = lstm(x, (h0, c0))
    ih = (x[0] @ lstm.weight_ih_l0.t()).detach()
    hh = (h0[0] @ lstm.weight_hh_l0.t()).detach()
    bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach()
    h, c = torch.empty(B, H, device=DEV), c0[0].clone().to(DEV)
    hpc_models.actor_critic_lstm_activation([ih.to(DEV), hh.to(DEV), bias.to(DEV)], [h, c])
    assert np.allclose(hn[0].detach().numpy(), h.cpu().numpy(), rtol=1e-5, atol=1e-5)
    assert np.allclose(cn[0].detach().numpy(), c.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,E,H", [(8, 512, 32), (3, 9, 100), (64, 200, 384)])
def test_pre_sample(B, E, H):
    import hpc_models
    g = torch.Generator().manual_seed(E)
    x = torch.randn(1, B, H, generator=g)
    key = torch.randn(B, E, H, generator=g)
    mask = torch.rand(B, E, generator=g) < 0.8
    ref = (x.permute(1, 0, 2) * key).sum(dim=2).masked_fill(~mask, -1e9).div(0.8)            # test_actor_critic.py:259-264
    out = torch.zeros(B, E, device=DEV)
    hpc_models.actor_critic_pre_sample([key.to(DEV), x.to(DEV), mask.to(DEV)], [out])
    assert np.allclose(ref.numpy(), out.cpu().numpy(), rtol=1e-5, atol=1e-5)                  # the reference's own assert


def test_models_golden_fixtures(golden):
    """tests/golden/models.npz: outputs of the reference TEST's own oracle expressions (torch_update_ae, torch.nn.LSTM,
    the pre-sample statements), executed from /root/reference by tests/golden/make_golden.py."""
    import hpc_models
    g = golden("models")
    for i, (B, E, D, _) in enumerate(g["ae_cases"]):
        ae = torch.from_numpy(g[f"ae{i}_ae"]).to(DEV)
        hpc_models.actor_critic_update_ae([torch.from_numpy(g[f"ae{i}_key"]).to(DEV), torch.from_numpy(g[f"ae{i}_sample"]).to(DEV),
                                           torch.from_numpy(g[f"ae{i}_num"]).to(DEV)], [ae])
        assert np.array_equal(ae.cpu().numpy(), g[f"ae{i}_out"])                              # one fp32 add: bit exact
    for i, (B, I, H, _) in enumerate(g["act_cases"]):
        h, c = torch.empty(int(B), int(H), device=DEV), torch.from_numpy(g[f"act{i}_c0"]).to(DEV)
        hpc_models.actor_critic_lstm_activation([torch.from_numpy(g[f"act{i}_{k}"]).to(DEV) for k in ("ih", "hh", "bias")], [h, c])
        assert np.allclose(g[f"act{i}_hn"], h.cpu().numpy(), rtol=1e-5, atol=1e-5)
        assert np.allclose(g[f"act{i}_cn"], c.cpu().numpy(), rtol=1e-5, atol=1e-5)
    for i, (B, E, H, _) in enumerate(g["pre_cases"]):
        out = torch.zeros(int(B), int(E), device=DEV)
        hpc_models.actor_critic_pre_sample([torch.from_numpy(g[f"pre{i}_key"]).to(DEV), torch.from_numpy(g[f"pre{i}_x"]).to(DEV),
                                            torch.from_numpy(g[f"pre{i}_mask"]).to(DEV)], [out])
        assert np.allclose(g[f"pre{i}_out"], out.cpu().numpy(), rtol=1e-5, atol=1e-5)        # the reference's own assert
