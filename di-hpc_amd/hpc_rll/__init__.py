"""hpc_rll for MI355X: the reference's ``hpc_rll.rl_utils`` / ``hpc_rll.torch_utils`` Python API on top of
hand-written gfx950 HIP kernels (``libhpc_rll_hip.so``, C ABI in ``include/hpc_rll_hip.h``).

Like the reference (all of whose ``__init__.py`` files are empty) nothing is re-exported here: import
the leaf modules, e.g. ``from hpc_rll.rl_utils.gae import GAE``.
"""
__version__ = "0.1.0"


def graphed(module, *example_inputs, **kwargs):
    """One hipGraph per forward+backward step of an hpc_rll module (see ``hpc_rll/graph.py``); an addition to the
    reference's surface for the launch-latency regime (small per-GPU batches under strong scaling)."""
    from .graph import graphed as _graphed
    return _graphed(module, *example_inputs, **kwargs)


def graphed_steps(module, step_inputs, **kwargs):
    """n consecutive steps (micro-batches with their own static buffers) in ONE hipGraph (see ``hpc_rll/graph.py``)."""
    from .graph import graphed_steps as _graphed_steps
    return _graphed_steps(module, step_inputs, **kwargs)
