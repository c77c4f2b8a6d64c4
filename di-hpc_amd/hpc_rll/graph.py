"""``hpc_rll.graphed`` -- one hipGraph per training step of an hpc_rll module.

Where it matters: the latency regime.  One rank of an 8-GPU strong-scaling run of the headline workload (GAE,
T=1024, global B=65536) holds B=8192: two kernels of ~17.5 us each, while an eager ``module(...)`` + ``.backward()``
costs the host 30-50 us per step, most of it torch's autograd engine handing the graph task to its device thread and
back.  A captured step is ONE ``hipGraphLaunch`` whatever the module launches inside; the kernels,
their order and their results are exactly the eager ones.  The reference has nothing comparable (every call is three
eager launches on the legacy stream); this is the MI355X-side answer to "launch-bound inner loops go
into hipGraphs".

    step = hpc_rll.graphed(GAE(T, B), value, reward, 0.99, 0.97, grad_outputs=grad_adv)
    for batch in loader:
        value.copy_(batch.value); reward.copy_(batch.reward)      # the example tensors ARE the static buffers
        adv, (d_value, d_reward) = step()                           # one hipGraphLaunch: forward + backward

Constraints (those of stream capture): the module's forward must not synchronise with the host -- ``PPO`` returns
python floats by default (``.tolist()``, reference rl_utils/ppo.py:148); construct it with ``PPO(B, N, sync_info=False)``
(monitors stay device tensors) to capture it; shapes are fixed at capture time.
The extension's own caches are capture-safe (a cold ``gae_coef`` table is filled inside the capture).
"""
from typing import Any, List, Optional, Sequence, Tuple

import torch


def _tensors_of(obj: Any, out: List[torch.Tensor]) -> None:
    if isinstance(obj, torch.Tensor):
        out.append(obj)
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            _tensors_of(o, out)
    elif isinstance(obj, dict):
        for o in obj.values():
            _tensors_of(o, out)


class GraphedStep:
    """A captured forward(+backward) step.  Attributes:
      inputs   the positional arguments given to :func:`graphed` (tensors among them are the static input buffers);
      outputs  the module's return value as produced inside the capture (static tensors, overwritten by every replay);
      wrt      the tensors gradients are taken with respect to: the input tensors that require grad, then the module's
               parameters that require grad;
      grads    tuple aligned with ``wrt`` (None where the output does not depend on the tensor), overwritten by every
               replay -- gradients are NOT accumulated into ``.grad`` (copy or add them where the optimiser wants them).
    Calling the object replays the graph on the current stream and returns ``(outputs, grads)`` (``outputs`` alone for a
    forward-only capture)."""

    def __init__(self, graph, inputs, outputs, wrt, grads, backward):
        self.graph, self.inputs, self.outputs, self.wrt, self.grads, self.backward = graph, inputs, outputs, wrt, grads, backward

    def replay(self) -> None:
        self.graph.replay()

    def __call__(self):
        self.graph.replay()
        return (self.outputs, self.grads) if self.backward else self.outputs


def graphed(module, *example_inputs, grad_outputs: Optional[Sequence[torch.Tensor]] = None, backward: Optional[bool] = None,
            warmup: int = 3, pool=None, **kwargs) -> GraphedStep:
    """Capture ``out = module(*example_inputs, **kwargs)`` and, when something requires grad, the backward pass
    ``autograd.grad(out, wrt, grad_outputs)`` into ONE hipGraph.

    ``grad_outputs``: a tensor (or a sequence, one per output tensor) of upstream gradients, static like the inputs;
    default ones.  ``backward=False`` captures the forward only.  ``warmup`` eager iterations run first on a side stream
    (allocator / cache warm-up, as torch.cuda.graphs prescribes).  ``pool``: a ``torch.cuda.graph_pool_handle()`` to share
    memory between several captured steps."""
    tens: List[torch.Tensor] = []
    _tensors_of(example_inputs, tens)
    _tensors_of(kwargs, tens)
    assert tens and all(t.is_cuda for t in tens), "graphed(): the example inputs must be GPU tensors"
    params = [p for p in module.parameters() if p.requires_grad] if isinstance(module, torch.nn.Module) else []
    wrt = [t for t in tens if t.requires_grad] + params
    if backward is None:
        backward = bool(wrt)
    assert not backward or wrt, "graphed(backward=True): nothing requires grad"
    if isinstance(grad_outputs, torch.Tensor):
        grad_outputs = [grad_outputs]

    def fwd_bwd():
        out = module(*example_inputs, **kwargs)
        if not backward:
            return out, ()
        outs: List[torch.Tensor] = []
        _tensors_of(out, outs)
        outs = [o for o in outs if o.requires_grad]
        assert outs, "graphed(): no output depends on a tensor that requires grad"
        gos = list(grad_outputs) if grad_outputs is not None else [torch.ones_like(o) for o in outs]
        assert len(gos) == len(outs), f"grad_outputs: {len(gos)} given for {len(outs)} differentiable outputs"
        return out, torch.autograd.grad(outs, wrt, gos, allow_unused=True)

    dev = tens[0].device
    with torch.cuda.device(dev):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool):
            out, grads = fwd_bwd()
    return GraphedStep(g, example_inputs, out, tuple(wrt), tuple(grads), backward)


class GraphedSteps:
    """n forward(+backward) steps -- n micro-batches, each on its OWN static input / upstream-gradient buffers -- captured
    into ONE hipGraph (:func:`graphed_steps`).  ``outputs[i]`` / ``grads[i]`` are step i's (as in :class:`GraphedStep`);
    ``wrt[i]`` the tensors its gradients are taken with respect to.  Calling the object replays all n steps with one
    ``hipGraphLaunch`` and returns ``(outputs, grads)``."""

    def __init__(self, graph, inputs, outputs, wrt, grads):
        self.graph, self.inputs, self.outputs, self.wrt, self.grads = graph, inputs, outputs, wrt, grads
        self.steps = len(inputs)

    def replay(self) -> None:
        self.graph.replay()

    def __call__(self):
        self.graph.replay()
        return self.outputs, self.grads


def graphed_steps(module, step_inputs: Sequence[Sequence[Any]], grad_outputs: Optional[Sequence[Any]] = None, warmup: int = 3,
                  pool=None) -> GraphedSteps:
    """Capture ``len(step_inputs)`` consecutive training steps of ``module`` into ONE hipGraph: step i is
    ``out_i = module(*step_inputs[i])`` followed by ``autograd.grad(out_i, wrt_i, grad_outputs[i])``, in order, exactly the
    kernels of i eager steps.  For the launch-latency regime with gradient accumulation over micro-batches (or several
    rollout shards per optimiser step): the host pays one ``hipGraphLaunch`` per n steps and the GPU-side boundary between
    two graph launches (a few microseconds, tests/tools/r05_gae_gap_probe.py) once per n steps instead of once per step.
    Every step has its own static buffers (refill them between replays); ``grad_outputs[i]`` is a tensor or a sequence per
    differentiable output of step i (default ones)."""
    n = len(step_inputs)
    assert n >= 1
    params = [p for p in module.parameters() if p.requires_grad] if isinstance(module, torch.nn.Module) else []
    wrts, flat = [], []
    for args in step_inputs:
        tens: List[torch.Tensor] = []
        _tensors_of(args, tens)
        assert tens and all(t.is_cuda for t in tens), "graphed_steps(): the step inputs must be GPU tensors"
        wrts.append([t for t in tens if t.requires_grad] + params)
        flat += tens
    assert all(wrts), "graphed_steps(): every step needs something that requires grad"

    def all_steps():
        outs_all, grads_all = [], []
        for i, args in enumerate(step_inputs):
            out = module(*args)
            outs: List[torch.Tensor] = []
            _tensors_of(out, outs)
            outs = [o for o in outs if o.requires_grad]
            assert outs, "graphed_steps(): no output depends on a tensor that requires grad"
            go = None if grad_outputs is None else grad_outputs[i]
            if isinstance(go, torch.Tensor):
                go = [go]
            gos = list(go) if go is not None else [torch.ones_like(o) for o in outs]
            assert len(gos) == len(outs)
            outs_all.append(out)
            grads_all.append(torch.autograd.grad(outs, wrts[i], gos, allow_unused=True))
        return outs_all, grads_all

    dev = flat[0].device
    with torch.cuda.device(dev):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                all_steps()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool):
            outs_all, grads_all = all_steps()
    return GraphedSteps(g, list(step_inputs), outs_all, [tuple(w) for w in wrts], [tuple(x) for x in grads_all])
