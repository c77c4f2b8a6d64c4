#!/usr/bin/env python3
"""Shim: the secondary-config suite (timing harness + the algorithmic byte / flop / instruction models of SURVEY.md 8d) lives in
/bench_suite.py at the repo root, next to bench.py, which owns it (the driver-run numbers must not depend
on test tooling).  `python tests/tools/bench_suite.py [c3|td|gemm|c4|c5|small|all]` and `import bench_suite` from this
directory keep working."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if __name__ == "__main__":
    runpy.run_path(os.path.join(ROOT, "bench_suite.py"), run_name="__main__")
else:
    import importlib.util
    _spec = importlib.util.spec_from_file_location("_root_bench_suite", os.path.join(ROOT, "bench_suite.py"))
    _m = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(_m)
    sys.modules[__name__] = _m      # `import bench_suite as S; S.dev = ...` acts on the real module
