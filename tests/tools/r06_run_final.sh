#!/bin/bash
# round 6, final check: the driver's own sequence -- GPU tier, smoke, bench -- plus the rocprofv3 evidence of the headline and the suites
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r06_pytest_final.log 2>&1
echo "pytest rc=$?"; grep -v amdgpu gpurun_out/r06_pytest_final.log | tail -3
cp gpurun_out/parity_probe.json gpurun_out/r06_parity_probe.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
tests/tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1
echo "collect rc=$?"; cp gpurun_out/bench.log gpurun_out/r06_bench_n1.json
tests/tools/collect_suite_profiles.sh r06 c3 c4 c5 td small > gpurun_out/r06_collect_suite.log 2>&1
echo "suite profiles rc=$?"
timeout 900 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
echo "bench default rc=$?"; cut -c1-300 gpurun_out/r06_bench_default.json
ls gpurun_out | grep r06_ | head -40
