#!/bin/bash
# round 3, GPU call 5: TD-family tests after the kernel changes, starved-LSTM test, then the td / c3 suites
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo skip-pytest

timeout 600 python tests/tools/bench_suite.py td > gpurun_out/r03_suite_td.log 2>&1
echo "suite td rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03_suite_td.log | tail -8
timeout 600 python tests/tools/bench_suite.py c3 > gpurun_out/r03_suite_c3.log 2>&1
echo "suite c3 rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03_suite_c3.log | tail -6
