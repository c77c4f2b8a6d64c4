"""CPU oracle for the hpc_rll hot path -- TEST INFRASTRUCTURE ONLY.

This module is a from-the-maths restatement of what the
reference's pure-PyTorch ground truth ``hpc_rll.origin`` computes.  It is the
checker for the HIP kernels; it is never the thing shipped or measured.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  Nothing under ``di-hpc_amd/`` imports it.

Pinning: every function here is compared against ``/root/reference``'s
``hpc_rll.origin`` in the build container by ``tests/golden/make_golden.py``
and against the committed fixtures in ``tests/golden/*.npz`` by
``tests/test_oracle_golden.py`` (the reference ships no golden vectors of its
own -- SURVEY.md section 8c).

All functions are differentiable torch code (gradients come from autograd, so
the oracle's backward is independent of the analytic adjoints the kernels
implement), take any float dtype (tests feed float64 for a tight reference)
and run on CPU.  Each function cites the reference lines it follows.
"""
from __future__ import annotations

from collections import namedtuple
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------
# GAE  (reference: hpc_rll/origin/gae.py:28-37, gae_kernel.h:14-28)
# ---------------------------------------------------------------------------


def gae_denominators(T: int, lambda_: float) -> List[float]:
    """D_t = 1 + lambda * D_{t+1}, D_T = 0 (python floats, i.e. fp64, like the
    reference's ``denom`` at origin/gae.py:34)."""
    d = [0.0] * (T + 1)
    for t in range(T - 1, -1, -1):
        d[t] = 1.0 + lambda_ * d[t + 1]
    return d


def gae(value: torch.Tensor, reward: torch.Tensor, gamma: float = 0.99, lambda_: float = 0.97) -> torch.Tensor:
    """Truncation-normalised GAE: G_t = D_t*delta_t + gamma*lambda*G_{t+1}; adv_t = G_t / D_t.

    value (T+1,B), reward (T,B) -> adv (T,B).  origin/gae.py:28-37.
    """
    T = reward.shape[0]
    d = gae_denominators(T, lambda_)
    delta = reward + gamma * value[1:] - value[:-1]
    rows = [None] * T
    g = torch.zeros_like(reward[0])
    for t in range(T - 1, -1, -1):
        g = d[t] * delta[t] + (gamma * lambda_) * g
        rows[t] = g / d[t]
    return torch.stack(rows, 0)


def gae_backward(grad_adv: torch.Tensor, gamma: float, lambda_: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """Analytic adjoint of :func:`gae`.  Returns (grad_value (T+1,B), grad_reward (T,B)).

    h_t = g_t/D_t + gamma*lambda*h_{t-1};  dL/ddelta_t = D_t*h_t;
    dL/dr_t = dL/ddelta_t;  dL/dV_t = -dL/ddelta_t + gamma*dL/ddelta_{t-1}.
    (The reference module has no backward: rl_utils/gae.py:17-18.)
    """
    T, B = grad_adv.shape
    d = gae_denominators(T, lambda_)
    ddelta = torch.zeros_like(grad_adv)
    h = torch.zeros_like(grad_adv[0])
    for t in range(T):
        h = grad_adv[t] / d[t] + (gamma * lambda_) * h
        ddelta[t] = d[t] * h
    gv = torch.zeros(T + 1, B, dtype=grad_adv.dtype)
    gv[:T] -= ddelta
    gv[1:] += gamma * ddelta
    return gv, ddelta


# ---------------------------------------------------------------------------
# TD(lambda)  (origin/td.py:148-244; td_lambda_kernel.h:17-50)
# ---------------------------------------------------------------------------


def lambda_returns(value: torch.Tensor, reward: torch.Tensor, gamma: float, lambda_) -> torch.Tensor:
    """ret_{T-1} = r + gamma*V_T ; ret_t = r_t + gamma*(lam_t*ret_{t+1} + (1-lam_t)*V_{t+1}).

    ``lambda_`` may be a float or a (T,B) tensor (UPGO uses a 0/1 tensor).  origin/td.py:235-244.
    """
    T = reward.shape[0]
    rows = [None] * T
    rows[T - 1] = reward[T - 1] + gamma * value[T]
    for t in range(T - 2, -1, -1):
        lam = lambda_[t] if isinstance(lambda_, torch.Tensor) else lambda_
        rows[t] = reward[t] + gamma * (lam * rows[t + 1] + (1 - lam) * value[t + 1])
    return torch.stack(rows, 0)


def td_lambda_error(value, reward, weight=None, gamma: float = 0.9, lambda_: float = 0.8) -> torch.Tensor:
    """0.5 * mean(w * (ret - V[:-1])^2), no gradient through ret.  origin/td.py:168-176."""
    with torch.no_grad():
        ret = lambda_returns(value, reward, gamma, lambda_)
    w = torch.ones_like(reward) if weight is None else weight
    return 0.5 * (w * (ret - value[:-1]) ** 2).mean()


# ---------------------------------------------------------------------------
# categorical helpers shared by V-trace / UPGO / PPO
# ---------------------------------------------------------------------------


def _logp_and_entropy(logits: torch.Tensor, action: torch.Tensor):
    """log pi(a) and entropy of Categorical(logits=...).  Masked actions (logit = -inf) have probability 0 and
    contribute 0 to the entropy: torch.distributions.Categorical.entropy(), which hpc_rll.origin calls
    (origin/vtrace.py:76-79, origin/ppo.py:57-61), clamps log p to the most negative finite value before p*log p."""
    logp_all = F.log_softmax(logits, dim=-1)
    logp = logp_all.gather(-1, action.unsqueeze(-1)).squeeze(-1)
    ent = -(logp_all.exp() * logp_all.clamp(min=torch.finfo(logp_all.dtype).min)).sum(-1)
    return logp, ent


# ---------------------------------------------------------------------------
# V-trace  (origin/vtrace.py:5-111; vtrace_kernel.h)
# ---------------------------------------------------------------------------

vtrace_loss = namedtuple("vtrace_loss", ["policy_loss", "value_loss", "entropy_loss"])


def vtrace_error(target_output, behaviour_output, action, value, reward, weight=None,
                 gamma: float = 0.99, lambda_: float = 0.95,
                 rho_clip_ratio: float = 1.0, c_clip_ratio: float = 1.0, rho_pg_clip_ratio: float = 1.0):
    """IMPALA V-trace losses.  origin/vtrace.py:63-79 (+ :5-17 for the return scan)."""
    T = reward.shape[0]
    logp_t, ent = _logp_and_entropy(target_output, action)
    with torch.no_grad():
        logp_b, _ = _logp_and_entropy(behaviour_output, action)
        is_w = torch.exp(logp_t - logp_b)
        rho = is_w.clamp(max=rho_clip_ratio)
        cs = is_w.clamp(max=c_clip_ratio)
        delta = rho * (reward + gamma * value[1:] - value[:-1])
        item = torch.zeros_like(reward[0])
        vs = [None] * T
        for t in range(T - 1, -1, -1):
            item = delta[t] + gamma * lambda_ * cs[t] * item
            vs[t] = value[t] + item
        vs = torch.stack(vs, 0)
        vs_next = torch.cat([vs[1:], value[-1:]], 0)
        adv = is_w.clamp(max=rho_pg_clip_ratio) * (reward + gamma * vs_next - value[:-1])
    w = torch.ones_like(reward) if weight is None else weight
    pg = -(logp_t * adv * w).mean()
    vl = (w * (value[:-1] - vs) ** 2).mean()
    el = (ent * w).mean()
    return vtrace_loss(pg, vl, el)


# ---------------------------------------------------------------------------
# UPGO  (origin/upgo.py:7-70; upgo_kernel.h:17-36)
# ---------------------------------------------------------------------------


def upgo_returns(reward, value):
    """ret_t = r_t + (ret_{t+1} if r_{t+1}+V_{t+2} >= V_{t+1} else V_{t+1}); last step bootstraps V_T.
    origin/upgo.py:36-38."""
    lam = ((reward + value[1:]) >= value[:-1]).to(reward.dtype)
    lam = torch.cat([lam[1:], torch.ones_like(lam[-1:])], 0)
    return lambda_returns(value, reward, 1.0, lam)


def upgo_loss(target_output, rhos, action, reward, value) -> torch.Tensor:
    """-mean(rho*(ret - V[:-1]) * log pi(a)).  origin/upgo.py:64-70."""
    with torch.no_grad():
        adv = rhos * (upgo_returns(reward, value) - value[:-1])
    logp, _ = _logp_and_entropy(target_output, action)
    return -(adv * logp).mean()


# ---------------------------------------------------------------------------
# PPO  (origin/ppo.py:51-80; ppo_kernel.h:158-241)
# ---------------------------------------------------------------------------

ppo_loss = namedtuple("ppo_loss", ["policy_loss", "value_loss", "entropy_loss"])
ppo_info = namedtuple("ppo_info", ["approx_kl", "clipfrac"])


def ppo_error(logit_new, logit_old, action, value_new, value_old, adv, return_, weight=None,
              clip_ratio: float = 0.2, use_value_clip: bool = True, dual_clip: Optional[float] = None):
    w = torch.ones_like(adv) if weight is None else weight
    logp_new, ent = _logp_and_entropy(logit_new, action)
    logp_old, _ = _logp_and_entropy(logit_old, action)
    ratio = torch.exp(logp_new - logp_old)
    s1 = ratio * adv
    s2 = ratio.clamp(1 - clip_ratio, 1 + clip_ratio) * adv
    inner = torch.min(s1, s2)
    if dual_clip is not None:
        inner = torch.max(inner, dual_clip * adv)
    policy = (-inner * w).mean()
    if use_value_clip:
        vclip = value_old + (value_new - value_old).clamp(-clip_ratio, clip_ratio)
        v = torch.max((return_ - value_new) ** 2, (return_ - vclip) ** 2)
    else:
        v = (return_ - value_new) ** 2
    value = 0.5 * (v * w).mean()
    entropy = (ent * w).mean()
    with torch.no_grad():
        kl = (logp_old - logp_new).mean().item()
        frac = ((ratio > 1 + clip_ratio) | (ratio < 1 - clip_ratio)).to(adv.dtype).mean().item()
    return ppo_loss(policy, value, entropy), ppo_info(kl, frac)


# ---------------------------------------------------------------------------
# n-step TD family  (origin/td.py:9-22, 29-143, 252-517)
# ---------------------------------------------------------------------------


def _nstep_reward(reward: torch.Tensor, gamma: float) -> torch.Tensor:
    """R_b = sum_t gamma^t r[t,b]  (origin/td.py:349-352)."""
    n = reward.shape[0]
    f = torch.tensor([gamma ** i for i in range(n)], dtype=reward.dtype, device=reward.device)
    return (f.unsqueeze(1) * reward).sum(0)


def value_transform(x, eps: float = 1e-2):
    return torch.sign(x) * (torch.sqrt(torch.abs(x) + 1) - 1) + eps * x


def value_inv_transform(x, eps: float = 1e-2):
    return torch.sign(x) * (((torch.sqrt(1 + 4 * eps * (torch.abs(x) + 1 + eps)) - 1) / (2 * eps)) ** 2 - 1)


def q_nstep_td_error(q, next_n_q, action, next_n_action, reward, done, weight, gamma: float, rescale: bool = False):
    """(mean(w*(q[b,a]-tgt)^2), per-sample (q-tgt)^2).  origin/td.py:280-291 and :326-340 (rescale)."""
    nstep = reward.shape[0]
    B = action.shape[0]
    idx = torch.arange(B, device=q.device)
    qsa = q[idx, action]
    with torch.no_grad():
        tq = next_n_q[idx, next_n_action]
        if rescale:
            tq = value_inv_transform(tq)
        tgt = _nstep_reward(reward, gamma) + (gamma ** nstep) * tq * (1 - done)
        if rescale:
            tgt = value_transform(tgt)
    w = torch.ones_like(qsa) if weight is None else weight
    per = (qsa - tgt) ** 2
    return (per * w).mean(), per


def dist_nstep_td_error(dist, next_n_dist, action, next_n_action, reward, done, weight,
                        gamma: float, v_min: float, v_max: float, n_atom: int):
    """C51 projection + cross entropy.  origin/td.py:56-143.  Mass whose projected position is
    integral is dropped (l==u), exactly like the reference."""
    nstep = reward.shape[0]
    B = action.shape[0]
    idx = torch.arange(B)
    dt = dist.dtype
    support = torch.linspace(v_min, v_max, n_atom, dtype=dt)
    dz = (v_max - v_min) / (n_atom - 1)
    with torch.no_grad():
        R = _nstep_reward(reward, gamma).unsqueeze(-1)
        nd = next_n_dist[idx, next_n_action]
        tz = (R + (1 - done).unsqueeze(-1) * (gamma ** nstep) * support).clamp(v_min, v_max)
        b = (tz - v_min) / dz
        lo = b.floor().long()
        up = b.ceil().long()
        proj = torch.zeros_like(nd)
        proj.scatter_add_(1, lo, nd * (up.to(dt) - b))
        proj.scatter_add_(1, up, nd * (b - lo.to(dt)))
    logp = torch.log(dist[idx, action])
    w = torch.ones(B, dtype=dt) if weight is None else weight
    per = -(logp * proj).sum(-1)
    loss = (per * w).mean()
    return loss, per


def iqn_nstep_td_error(q, next_n_q, action, next_n_action, reward, done, replay_quantiles, weight,
                       gamma: float, kappa: float = 1.0, value_gamma=None):
    """q (tau,B,N), next_n_q (tau',B,N), replay_quantiles (tau,B).  origin/td.py:391-448."""
    tau, B, _ = q.shape
    tau_p = next_n_q.shape[0]
    nstep = reward.shape[0]
    idx = torch.arange(B)
    qsa = q[:, idx, action].transpose(0, 1)                      # (B,tau)
    with torch.no_grad():
        tq = next_n_q[:, idx, next_n_action].transpose(0, 1)     # (B,tau')
        vg = (gamma ** nstep) if value_gamma is None else value_gamma.unsqueeze(-1)
        tgt = _nstep_reward(reward, gamma).unsqueeze(-1) + vg * tq * (1 - done).unsqueeze(-1)
    e = tgt[:, :, None] - qsa[:, None, :]                       # (B,tau',tau)
    huber = torch.where(e.abs() <= kappa, 0.5 * e ** 2, kappa * (e.abs() - 0.5 * kappa))
    rq = replay_quantiles.reshape(tau, B).transpose(0, 1)[:, None, :]   # (B,1,tau)
    qh = (rq - (e < 0).to(q.dtype).detach()).abs() * huber / kappa
    per = qh.sum(2).mean(1)
    w = torch.ones(B, dtype=q.dtype) if weight is None else weight
    return (per * w).mean(), per


def qrdqn_nstep_td_error(q, next_n_q, action, next_n_action, reward, done, tau, weight,
                         gamma: float, value_gamma=None):
    """q (B,N,tau).  ``tau`` is whatever the caller passes (the reference test passes the integer
    count, tests/test_qrdqn_nstep_td_error.py:57).  origin/td.py:480-517."""
    B = action.shape[0]
    nstep = reward.shape[0]
    idx = torch.arange(B)
    qsa = q[idx, action, :].unsqueeze(2)                         # (B,tau,1)
    with torch.no_grad():
        tq = next_n_q[idx, next_n_action, :].unsqueeze(1)        # (B,1,tau)
        vg = (gamma ** nstep) if value_gamma is None else value_gamma.reshape(B, 1, 1)
        tgt = _nstep_reward(reward, gamma).reshape(B, 1, 1) + vg * tq * (1 - done).reshape(B, 1, 1)
    u = F.smooth_l1_loss(tgt.expand(B, qsa.shape[1], tq.shape[2]), qsa.expand(B, qsa.shape[1], tq.shape[2]),
                         reduction="none")
    ind = ((tgt - qsa).detach() <= 0).to(q.dtype)
    per = (u * (tau - ind).abs()).sum(-1).mean(1)
    w = torch.ones(B, dtype=q.dtype) if weight is None else weight
    return (per * w).mean(), per


# ---------------------------------------------------------------------------
# Padding  (origin/padding.py; integer / byte exact)
# ---------------------------------------------------------------------------


def pad(xs: Sequence[torch.Tensor], value: int = 0):
    """Ragged list of k-D tensors -> (new_x (n,max...), mask same dtype, shapes).  origin/padding.py:53-63,118-173."""
    shapes = [tuple(t.shape) for t in xs]
    nd = len(shapes[0])
    mx = [max(s[d] for s in shapes) for d in range(nd)]
    new_x = torch.full([len(xs)] + mx, value, dtype=xs[0].dtype)
    mask = torch.full([len(xs)] + mx, value, dtype=xs[0].dtype)
    for i, t in enumerate(xs):
        sl = (i,) + tuple(slice(0, s) for s in shapes[i])
        new_x[sl] = t
        mask[sl] = 1
    return new_x, mask, shapes


def unpad(x: torch.Tensor, shapes) -> List[torch.Tensor]:
    out = []
    for i, s in enumerate(shapes):
        sl = (i,) + tuple(slice(0, d) for d in s)
        out.append(x[sl].clone())
    return out


def oracle_split_group(numels: Sequence[int], group: int):
    """DP that minimises sum over groups of (max numel in group * group size) over contiguous
    splits of the (sorted) list.  Returns boundary positions [0, ..., n] (len group+1).
    origin/padding.py:11-50 (ties broken towards the smaller split point, like python's min over
    (cost, k) tuples)."""
    n, m = len(numels), group
    arr = [None] + list(numels)
    f = {(0, 0): (0, 0)}
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            best = None
            for k in range(0, i):
                if (k, j - 1) in f:
                    c = f[(k, j - 1)][0] + arr[i] * (i - k)
                    if best is None or (c, k) < best:
                        best = (c, k)
            if best is not None:
                f[(i, j)] = best
    pos, cnt, positions = n, m, [n]
    while pos > 0:
        pos = f[(pos, cnt)][1]
        cnt -= 1
        positions.append(pos)
    return positions[::-1]


# ---------------------------------------------------------------------------
# ScatterConnection  (origin/scatter_connection.py:49-65)
# ---------------------------------------------------------------------------


def scatter_connection(x: torch.Tensor, location: torch.Tensor, H: int, W: int, scatter_type: str) -> torch.Tensor:
    """x (B,M,N), location (B,M,2) int64 (y,x) -> (B,N,H,W).  'cover': largest m wins on collision
    (CPU ``scatter_`` is sequential); 'add': sum."""
    B, M, N = x.shape
    cell = location[..., 0] * W + location[..., 1]                  # (B,M)
    out = torch.zeros(B, H * W, N, dtype=x.dtype, device=x.device)
    if scatter_type == "add":
        out = out.scatter_add(1, cell.unsqueeze(-1).expand(B, M, N), x)
    elif scatter_type == "cover":
        # forward: sequential semantics, the largest m at a cell wins.
        # backward: torch's scatter_ backward is grad_src = grad_out.gather(index), i.e. EVERY
        # entity (also the overwritten ones) receives the gradient of its cell -- which is also what
        # the reference's backward kernel does (scatter_connection_kernel.h:91-106).  scatter_add
        # has exactly that backward, so use it as the straight-through gradient carrier.
        win = torch.full((B, H * W), -1, dtype=torch.long, device=x.device)
        rows = torch.arange(B, device=x.device)
        for m in range(M):
            win[rows, cell[:, m]] = m
        gathered = x.detach()[rows.unsqueeze(1), win.clamp(min=0)]
        fwd = torch.where((win >= 0).unsqueeze(-1), gathered, torch.zeros_like(gathered))
        carrier = out.scatter_add(1, cell.unsqueeze(-1).expand(B, M, N), x)
        out = fwd + (carrier - carrier.detach())
    else:
        raise ValueError(scatter_type)
    return out.reshape(B, H, W, N).permute(0, 3, 1, 2).contiguous()


# ---------------------------------------------------------------------------
# LayerNorm LSTM  (origin/rnn.py:193-248)
# ---------------------------------------------------------------------------


def _lstm_step(xs, h, c, wx, wh, gx, bx, gh, bh, b, eps):
    H4 = wh.shape[1]
    gate = F.layer_norm(xs @ wx, (H4,), gx, bx, eps) + F.layer_norm(h @ wh, (H4,), gh, bh, eps) + b
    i, f, o, u = gate.chunk(4, dim=1)
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(u)
    h = torch.sigmoid(o) * torch.tanh(c)
    return h, c


def lstm(x, h0, c0, wx: Sequence[torch.Tensor], wh: Sequence[torch.Tensor], bias, ln_gamma, ln_beta,
         eps: float = 1e-5, checkpoint_steps: bool = False):
    """x (S,B,in); h0,c0 (L,B,H); wx[l] (in_l,4H); wh[l] (H,4H); bias (L,4H);
    ln_gamma/ln_beta (L, 2*4H) = [x-half | h-half].  gate order i,f,o,u.  Returns y (S,B,H), h (L,B,H), c (L,B,H).
    ``checkpoint_steps``: recompute every time step in backward (torch.utils.checkpoint) instead of keeping its
    activations -- same arithmetic, lets the full configs[3] size run in fp64 on one GPU."""
    S = x.shape[0]
    L = len(wx)
    H4 = wh[0].shape[1]
    hs, cs = [], []
    inp = x
    for l in range(L):
        h, c = h0[l], c0[l]
        outs = []
        gx, bx = ln_gamma[l, :H4], ln_beta[l, :H4]
        gh, bh = ln_gamma[l, H4:], ln_beta[l, H4:]
        for s in range(S):
            if checkpoint_steps:
                from torch.utils.checkpoint import checkpoint
                h, c = checkpoint(_lstm_step, inp[s], h, c, wx[l], wh[l], gx, bx, gh, bh, bias[l], eps,
                                  use_reentrant=False)
            else:
                h, c = _lstm_step(inp[s], h, c, wx[l], wh[l], gx, bx, gh, bh, bias[l], eps)
            outs.append(h)
        inp = torch.stack(outs, 0)
        hs.append(h)
        cs.append(c)
    return inp, torch.stack(hs, 0), torch.stack(cs, 0)
