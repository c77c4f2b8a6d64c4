#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_losses_gpu.py tests/test_full_size_gpu.py tests/test_fuzz_gpu.py tests/test_edge_cases_gpu.py -m gpu -x -q -k "td_lambda or tdlambda or c3 or fuzz_scan or TDLambda or td" 2>&1 | tail -4
timeout 300 python bench_suite.py c3 2>&1 | grep -i "td_lambda\|td-lambda\|TD" | head -5
