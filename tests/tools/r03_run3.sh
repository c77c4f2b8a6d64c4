#!/bin/bash
# round 3, GPU call 3: device grouped padding tests first (new code), then the whole GPU tier, then the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_pad_scatter_gpu.py -m gpu -q -p no:cacheprovider -x -k "packed_group" > gpurun_out/r03_pytest_pad.log 2>&1
echo "pytest(pad group) rc=$?"; tail -25 gpurun_out/r03_pytest_pad.log
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03_pytest_all.log 2>&1
echo "pytest(all) rc=$?"; tail -12 gpurun_out/r03_pytest_all.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_n1_b.json 2> gpurun_out/r03_bench_n1_b.err
echo "bench rc=$?"; cut -c1-1800 gpurun_out/r03_bench_n1_b.json; tail -3 gpurun_out/r03_bench_n1_b.err
