#!/bin/bash
# round 3, GPU call 2: strict GPU tier, rocprofv3 evidence of the headline (kernel trace + PMC), pipelined-GAE sweep at other shapes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r03_pytest_strict.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r03_pytest_strict.log
cp gpurun_out/parity_probe.json gpurun_out/r03_parity_probe.json 2>/dev/null
bash tests/tools/collect_profiles.sh r03 2>&1 | tail -12
for shape in "1024 131072" "1024 32768" "256 262144"; do
  set -- $shape
  TUNE_T=$1 TUNE_B=$2 NREP=16 timeout 600 python tests/tools/r03_gae_pf_sweep.py > gpurun_out/r03_gae_pf_sweep_$1x$2.log 2>&1
  echo "sweep $shape rc=$?"; grep -A8 "shipped auto" gpurun_out/r03_gae_pf_sweep_$1x$2.log | head -4
done
