#!/bin/bash
# Where the grouped packed pad of 2^20 rows spends its time: end-to-end wall time (tests/tools/r03_group_pad_time.py) and a
# rocprofv3 kernel trace of the same script.   gpurun -- 'bash tests/tools/r04_group_pad_profile.sh'
set -u
REPO=$(pwd)
mkdir -p "$REPO/gpurun_out"
python "$REPO/tests/tools/r03_group_pad_time.py" > "$REPO/gpurun_out/r04_group_pad_time.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp_prof
rocprofv3 --kernel-trace -d /tmp/gp_prof -o trace -- python "$REPO/tests/tools/r03_group_pad_time.py" > "$REPO/gpurun_out/r04_group_pad_under_rocprof.txt" 2>&1
DB=$(find /tmp/gp_prof -name "*.db" | head -1)
python "$REPO/tests/tools/summarize_suite_profile.py" "$DB" "$REPO/gpurun_out/r04_group_pad_kernel_stats.csv" > /dev/null
cat "$REPO/gpurun_out/r04_group_pad_time.txt"
cut -c1-150 "$REPO/gpurun_out/r04_group_pad_kernel_stats.csv" | head -24
