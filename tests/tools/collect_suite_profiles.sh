#!/bin/bash
# rocprofv3 --kernel-trace of the secondary-config suites (tests/tools/bench_suite.py c3 | c4 | c5 | small) on the GPU
# box; per-kernel statistics are reduced on the box (the rocpd databases stay there) and come back as
# gpurun_out/<tag>_suite_<x>_kernel_stats.csv, to be copied into profiles/.
#   gpurun -- 'tests/tools/collect_suite_profiles.sh r02 c3 c4 c5 small'
set -u
TAG=${1:-r02}; shift
REPO=$(pwd)
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for X in "$@"; do
    D=/tmp/prof_suite_$X
    rm -rf "$D"
    rocprofv3 --kernel-trace -d "$D" -o trace -- python "$REPO/tests/tools/bench_suite.py" "$X" > "$REPO/gpurun_out/${TAG}_suite_${X}_under_rocprof.txt" 2>&1
    DB=$(find "$D" -name "*.db" | head -1)
    python "$REPO/tests/tools/summarize_suite_profile.py" "$DB" "$REPO/gpurun_out/${TAG}_suite_${X}_kernel_stats.csv"
    rm -rf "$D"
done
