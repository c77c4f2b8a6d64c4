#!/bin/bash
# MFMA pipe utilisation of the exact-fp32 GEMM from hardware counters (its own rocprofv3 run, counters only):
#   tests/tools/collect_gemm_pmc.sh r02     -> gpurun_out/prof_<tag>/gemm_pmc/ ; summary by summarize_gemm_pmc.py
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    -d "$OUT/gemm_pmc" -o gemm_pmc -- python "$REPO/tests/tools/gemm_pmc_probe.py" > "$OUT/gemm_pmc.log" 2>&1
cd "$REPO"
ls "$OUT/gemm_pmc"
