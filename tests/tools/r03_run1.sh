#!/bin/bash
# round 3, GPU call 1: gradient-error probe of the whole GPU tier, pipelined-GAE sweep, bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ls /sys/class/drm/*/device/pp_dpm_sclk 2>&1 | head -3
cat /sys/class/drm/card*/device/pp_dpm_sclk 2>&1 | head -12
HPC_RLL_GRAD_PROBE_ONLY=1 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03_pytest_probe.log 2>&1
echo "pytest(probe) rc=$?"; tail -15 gpurun_out/r03_pytest_probe.log
cp gpurun_out/parity_probe.json gpurun_out/r03_parity_probe_all.json 2>/dev/null
timeout 600 python tests/tools/r03_gae_pf_sweep.py > gpurun_out/r03_gae_pf_sweep.log 2>&1
echo "sweep rc=$?"; tail -70 gpurun_out/r03_gae_pf_sweep.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_n1_a.json 2> gpurun_out/r03_bench_n1_a.err
echo "bench rc=$?"; tail -c 6000 gpurun_out/r03_bench_n1_a.json; tail -5 gpurun_out/r03_bench_n1_a.err
