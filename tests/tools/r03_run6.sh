#!/bin/bash
# round 3, GPU call: whole GPU tier, full suite, bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03_pytest_all.log 2>&1
echo "pytest(all) rc=$?"; grep -v amdgpu gpurun_out/r03_pytest_all.log | tail -6
cp gpurun_out/parity_probe.json gpurun_out/r03_parity_probe.json 2>/dev/null
timeout 900 python tests/tools/bench_suite.py all > gpurun_out/r03_suite_all.log 2>&1
echo "suite rc=$?"; cp gpurun_out/suite_all.json gpurun_out/r03_suite_all.json
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_n1_c.json 2> gpurun_out/r03_bench_n1_c.err
echo "bench rc=$?"; cut -c1-600 gpurun_out/r03_bench_n1_c.json
