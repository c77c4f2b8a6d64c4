#!/bin/bash
# HBM traffic of the secondary kernels from hardware counters: tests/tools/pmc_suite.py under rocprofv3 --pmc FETCH_SIZE and
# --pmc WRITE_SIZE (separate passes, counters only), reduced ON the box to gpurun_out/<tag>_suite_pmc_traffic.txt.
#   gpurun -- 'tests/tools/collect_suite_pmc.sh r02'
set -u
TAG=${1:-r02}
REPO=$(pwd)
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$C
    rocprofv3 --pmc $C -d /tmp/pmc_$C -o pmc -- python "$REPO/tests/tools/pmc_suite.py" > "$REPO/gpurun_out/${TAG}_pmc_$C.log" 2>&1
done
python "$REPO/tests/tools/summarize_suite_pmc.py" "$(find /tmp/pmc_FETCH_SIZE -name '*.db' | head -1)" \
    "$(find /tmp/pmc_WRITE_SIZE -name '*.db' | head -1)" "$REPO/gpurun_out/${TAG}_suite_traffic.json" > "$REPO/gpurun_out/${TAG}_suite_pmc_traffic.txt"
tail -40 "$REPO/gpurun_out/${TAG}_suite_pmc_traffic.txt"
