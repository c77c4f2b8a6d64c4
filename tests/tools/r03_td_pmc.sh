#!/bin/bash
# VALU instruction counts and wave-time split of the large-batch C51 / QR-DQN forwards from hardware counters (own
# rocprofv3 runs, counters only): gpurun -- 'bash tests/tools/r03_td_pmc.sh'  -> gpurun_out/r03_td_pmc.txt
set -u
REPO=$(pwd)
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tdpmc_a /tmp/tdpmc_b
rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d /tmp/tdpmc_a -o pmc -- \
    python "$REPO/tests/tools/r03_td_pmc_probe.py" > "$REPO/gpurun_out/r03_td_pmc_a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d /tmp/tdpmc_b -o pmc -- \
    python "$REPO/tests/tools/r03_td_pmc_probe.py" > "$REPO/gpurun_out/r03_td_pmc_b.log" 2>&1
{
  echo "# rocprofv3 --pmc (two passes, counters only), tests/tools/r03_td_pmc_probe.py: B = 262144, N = 64, 51 atoms / tau = 32"
  echo "# last of three launches per kernel; counter runs serialise and instrument the dispatches: compare shares, not us"
  python "$REPO/tests/tools/r03_td_pmc_summary.py" $(find /tmp/tdpmc_a /tmp/tdpmc_b -name "*.db")
} > "$REPO/gpurun_out/r03_td_pmc.txt" 2>&1
cat "$REPO/gpurun_out/r03_td_pmc.txt" | cut -c1-400
tail -3 "$REPO/gpurun_out/r03_td_pmc_a.log" "$REPO/gpurun_out/r03_td_pmc_b.log" | cut -c1-300
