#!/bin/bash
# round 3, GPU call 5: TD-family tests after the kernel changes, starved-LSTM test, then the td / c3 suites
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_losses_gpu.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py tests/test_lstm_gpu.py tests/test_reference_lists_gpu.py -m gpu -q -p no:cacheprovider -k "td or dntd or iqn or qrdqn or onehot or starved or fold or reference" > gpurun_out/r03_pytest_td.log 2>&1
echo "pytest(td) rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03_pytest_td.log | tail -15
timeout 600 python tests/tools/bench_suite.py td > gpurun_out/r03_suite_td.log 2>&1
echo "suite td rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03_suite_td.log | tail -8
timeout 600 python tests/tools/bench_suite.py c3 > gpurun_out/r03_suite_c3.log 2>&1
echo "suite c3 rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03_suite_c3.log | tail -6
