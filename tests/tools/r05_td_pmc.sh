#!/bin/bash
# Where the cycles of the n-step TD forwards go (C51, QR-DQN, IQN at the suite shapes): SQ issue / wait / LDS counters of the last
# launch of each, in separate rocprofv3 --pmc passes (8 SQ slots per pass; no trace domains beside them).
#   gpurun -- 'bash tests/tools/r05_td_pmc.sh'  -> gpurun_out/r05_td_pmc.json (copy to profiles/)
set -u
REPO=$(pwd)
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tdpmc
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P3="SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
    i=$((i + 1))
    rocprofv3 --pmc $P -d /tmp/tdpmc/p$i -o pmc -- python "$REPO/tests/tools/r04_td_valu_probe.py" > "$REPO/gpurun_out/r05_td_pmc_$i.log" 2>&1
    tail -1 "$REPO/gpurun_out/r05_td_pmc_$i.log" | cut -c1-160
done
python "$REPO/tests/tools/r05_td_pmc_summary.py" "$REPO/gpurun_out/r05_td_pmc.json" $(find /tmp/tdpmc -name "*.db")
