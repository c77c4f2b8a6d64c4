#!/bin/bash
# VALU instruction counts of the instruction-bound TD forwards (C51, QR-DQN, IQN) from hardware counters, for bench.py's
# `bound: "valu"` rows:  gpurun -- 'bash tests/tools/r04_td_valu.sh'  -> gpurun_out/td_valu.json (copy to profiles/)
set -u
REPO=$(pwd)
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tdvalu
rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE -d /tmp/tdvalu -o pmc -- python "$REPO/tests/tools/r04_td_valu_probe.py" \
    > "$REPO/gpurun_out/r04_td_valu.log" 2>&1
python "$REPO/tests/tools/r04_td_valu_summary.py" "$REPO/gpurun_out/td_valu.json" $(find /tmp/tdvalu -name "*.db")
tail -2 "$REPO/gpurun_out/r04_td_valu.log" | cut -c1-200
