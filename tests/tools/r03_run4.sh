#!/bin/bash
# round 3, GPU call 4: new tests (teacher-forced C4 LSTM, ADVICE fixes), wave-per-trajectory probe, fold litmus
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_lstm_gpu.py tests/test_edge_cases_gpu.py tests/test_reference_lists_gpu.py tests/test_gae_gpu.py -m gpu -q -p no:cacheprovider -s -k "teacher or starved or modified_in_place or reference or gae" > gpurun_out/r03_pytest_new.log 2>&1
echo "pytest(new) rc=$?"; grep -v "amdgpu.ids" gpurun_out/r03_pytest_new.log | tail -25
timeout 300 python tests/tools/r03_gae_wpt_probe.py > gpurun_out/r03_gae_wpt_probe.log 2>&1
echo "wpt rc=$?"; tail -12 gpurun_out/r03_gae_wpt_probe.log
timeout 600 python tests/tools/r03_fold_litmus.py > gpurun_out/r03_fold_litmus.log 2>&1
echo "litmus rc=$?"; tail -3 gpurun_out/r03_fold_litmus.log
