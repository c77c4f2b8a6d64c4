#!/bin/bash
# Collect the rocprofv3 evidence of the headline bench on the GPU box (run through gpurun from the repo root):
#   tests/tools/collect_profiles.sh r02
# 1. --kernel-trace --stats of `python bench.py` (the driver's own command line, CPU baseline and the extra scaling legs
#    skipped): per-kernel average durations, which must agree with the roofline.achieved of the bench line;
# 2. PMC passes, each in its OWN run with nothing else (FETCH_SIZE and WRITE_SIZE do not fit one pass): HBM bytes per
#    launch of the two GAE kernels + a calibration copy of known size (tests/tools/pmc_probe.py);
# 3. the plain bench line.   Summaries -> profiles/ by tests/tools/summarize_profiles.py <tag> (run afterwards, CPU side).
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -- python "$REPO/bench.py" --steps 200 --warmup 20 --skip-cpu-baseline \
    --no-scaling-detail --no-suite > "$OUT/trace.log" 2>&1
export PROBE_CONFIG_OUT="$OUT/gae_config.json"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- python "$REPO/tests/tools/pmc_probe.py" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o write -- python "$REPO/tests/tools/pmc_probe.py" > "$OUT/write.log" 2>&1
cd "$REPO"
python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err
find "$OUT" -name "*.csv" | head -20
tail -1 gpurun_out/bench.log | cut -c1-400
