#!/usr/bin/env python3
"""Measured parity errors that the FIXED tolerances written in tests/test_full_size_gpu.py and tests/test_lstm_gpu.py
come from: every error is max|ref - got| / max|ref| of the tensor (a per-tensor
scale, so that O(1/(T*B)) gradients are not compared against an absolute floor).  fp64 oracle evaluated with eager
PyTorch-ROCm ops on the GPU; `drift32` is the same oracle evaluated in fp32 (what torch itself loses at that shape).

    python tests/tools/r02_parity_probe.py > gpurun_out/r02_parity_probe.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import ref_torch as R  # noqa: E402

DEV = torch.device("cuda:0")


def nerr(ref, got):
    ref, got = ref.double(), got.double()
    return ((ref - got).abs().max() / ref.abs().max().clamp(min=1e-300)).item()


def out(**kw):
    print(json.dumps(kw), flush=True)


def lstm_case(name, S, B, I, H, L, seed, checkpoint, backward=True):
    from hpc_rll.torch_utils.network.rnn import LSTM
    torch.manual_seed(seed)
    m = LSTM(S, B, I, H, L).to(DEV)
    x = torch.randn(S, B, I, device=DEV, requires_grad=True)
    h0 = torch.randn(L, B, H, device=DEV, requires_grad=True)
    c0 = torch.randn(L, B, H, device=DEV, requires_grad=True)
    gy = torch.randn(S, B, H, device=DEV)
    y, (hn, cn) = m(x, (h0, c0))
    if backward:
        ((y * gy).sum() + hn.sum() - cn.sum()).backward()
    dims = [I] + [H] * L
    offs_x = [0]
    for l in range(L):
        offs_x.append(offs_x[-1] + dims[l] * 4 * H)

    def params(dt):
        leaf = lambda t: t.detach().to(dt).requires_grad_(True)  # noqa: E731
        wx = [leaf(m.wx[offs_x[l]:offs_x[l + 1]].reshape(dims[l], 4 * H)) for l in range(L)]
        wh = [leaf(m.wh[l * H * 4 * H:(l + 1) * H * 4 * H].reshape(H, 4 * H)) for l in range(L)]
        return wx, wh, leaf(m.bias.reshape(L, 4 * H)), leaf(m.ln_gamma), leaf(m.ln_beta)

    res = {}
    refs = {}
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        leaf = lambda t: t.detach().to(dt).requires_grad_(True)  # noqa: E731
        ox, oh0, oc0 = leaf(x), leaf(h0), leaf(c0)
        wx, wh, ob, og, obe = params(dt)
        oy, ohn, ocn = R.lstm(ox, oh0, oc0, wx, wh, ob, og, obe, checkpoint_steps=checkpoint)
        t = {"y": oy.detach(), "hn": ohn.detach(), "cn": ocn.detach()}
        if backward:
            ((oy * gy.to(dt)).sum() + ohn.sum() - ocn.sum()).backward()
            t.update(dx=ox.grad, dh0=oh0.grad, dc0=oc0.grad, dwx=torch.cat([w.grad.reshape(-1) for w in wx]),
                     dwh=torch.cat([w.grad.reshape(-1) for w in wh]), dbias=ob.grad.reshape(-1), dgamma=og.grad, dbeta=obe.grad)
        refs[tag] = t
        del ox, oh0, oc0, wx, wh, ob, og, obe, oy, ohn, ocn
        torch.cuda.empty_cache()
    got = {"y": y.detach(), "hn": hn.detach(), "cn": cn.detach()}
    if backward:
        got.update(dx=x.grad, dh0=h0.grad, dc0=c0.grad, dwx=m.wx.grad, dwh=m.wh.grad, dbias=m.bias.grad.reshape(-1),
                   dgamma=m.ln_gamma.grad, dbeta=m.ln_beta.grad)
    for k in got:
        res[k] = {"hip": nerr(refs["f64"][k], got[k]), "drift32": nerr(refs["f64"][k], refs["f32"][k]),
                  "max_ref": refs["f64"][k].abs().max().item()}
    out(case=name, shape=dict(S=S, B=B, I=I, H=H, L=L), err=res)


def c3_case():
    from hpc_rll.rl_utils.td import TDLambda
    from hpc_rll.rl_utils.upgo import UPGO
    from hpc_rll.rl_utils.vtrace import VTrace
    T, B, N = 256, 16384, 128
    g = torch.Generator(device=DEV).manual_seed(0)
    target = torch.randn(T, B, N, device=DEV, generator=g)
    behaviour = torch.randn(T, B, N, device=DEV, generator=g)
    action = torch.randint(0, N, (T, B), device=DEV, generator=g)
    value = torch.randn(T + 1, B, device=DEV, generator=g)
    reward = torch.randn(T, B, device=DEV, generator=g)
    weight = torch.rand(T, B, device=DEV, generator=g)
    rho = torch.rand(T, B, device=DEV, generator=g)
    v = value.clone().requires_grad_(True)
    TDLambda(T, B)(v, reward, weight, 0.9, 0.8).backward()
    v64 = value.double().requires_grad_(True)
    R.td_lambda_error(v64, reward.double(), weight.double(), 0.9, 0.8).backward()
    out(case="c3 td_lambda grad_value", err=nerr(v64.grad, v.grad), max_ref=v64.grad.abs().max().item())
    to = target.clone().requires_grad_(True)
    v = value.clone().requires_grad_(True)
    sum(VTrace(T, B, N)(to, behaviour, action, v, reward)).backward()
    to64 = target.double().requires_grad_(True)
    v64 = value.double().requires_grad_(True)
    sum(R.vtrace_error(to64, behaviour.double(), action, v64, reward.double(), None)).backward()
    out(case="c3 vtrace", err_grad_value=nerr(v64.grad, v.grad), err_grad_target=nerr(to64.grad, to.grad),
        max_ref_value=v64.grad.abs().max().item(), max_ref_target=to64.grad.abs().max().item())
    del to64
    to = target.clone().requires_grad_(True)
    UPGO(T, B, N)(to, rho, action, reward, value).backward()
    to64 = target.double().requires_grad_(True)
    R.upgo_loss(to64, rho.double(), action, reward.double(), value.double()).backward()
    lam32 = (reward + value[1:]) >= value[:-1]
    lam64 = (reward.double() + value[1:].double()) >= value[:-1].double()
    flips = (lam32 != lam64)
    agree_cols = ~flips.any(0)
    e_all = nerr(to64.grad, to.grad)
    e_agree = nerr(to64.grad[:, agree_cols], to.grad[:, agree_cols])
    out(case="c3 upgo", flips=int(flips.sum().item()), cols_with_flip=int((~agree_cols).sum().item()),
        err_grad_all=e_all, err_grad_agreeing_cols=e_agree, max_ref=to64.grad.abs().max().item())


if __name__ == "__main__":
    which = sys.argv[1:] or ["c3", "test", "mid", "c4"]
    if "c3" in which:
        c3_case()
    if "test" in which:
        lstm_case("lstm reference test shape", 64, 3, 1792, 384, 3, 0, False)
    if "mid" in which:
        lstm_case("lstm S=32 B=64 H=256 L=2", 32, 64, 128, 256, 2, 1, False)
        lstm_case("lstm C4 widths S=4", 4, 4096, 1024, 1024, 1, 1, False)
    if "c4" in which:
        lstm_case("lstm C4 full S=128", 128, 4096, 1024, 1024, 1, 0, True)
