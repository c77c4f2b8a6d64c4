#!/bin/bash
# round 3, final check: the driver's own sequence -- GPU tier, smoke, bench -- plus the suite and the large-batch TD probes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r03_pytest_final.log 2>&1
echo "pytest rc=$?"; grep -v amdgpu gpurun_out/r03_pytest_final.log | tail -3
cp gpurun_out/parity_probe.json gpurun_out/r03_parity_probe.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 900 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/r03_bench_default.json
timeout 900 python tests/tools/bench_suite.py all > gpurun_out/r03_suite_all.log 2>&1
echo "suite rc=$?"; cp gpurun_out/suite_all.json gpurun_out/r03_suite_all.json
PROBE_B=262144,131072,65536,32768,16384 PROBE_SW=1,8,16,32,64 timeout 600 python tests/tools/r03_batch_probe.py > gpurun_out/r03_batch_probe.log 2>&1
echo "batch probe rc=$?"
timeout 600 python tests/tools/r03_td_ab.py > gpurun_out/r03_td_ab.log 2>&1
echo "td ab rc=$?"; grep -v amdgpu gpurun_out/r03_td_ab.log | tail -4 | cut -c1-300
