#!/bin/bash
# Round 5, GPU call 1: scatter store-pattern ceiling, GAE per-rank step gaps, row-block backward variants, the changed tests.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== scatter sweep micro"; timeout 120 tests/tools/micro/scatter_sweep.bin > "$OUT/r05_scatter_sweep.txt" 2>&1; tail -45 "$OUT/r05_scatter_sweep.txt"
echo "== lstm block bwd variants"; timeout 400 python tests/tools/r05_lstm_block_bwd_ab.py > "$OUT/r05_lstm_block_bwd_ab.txt" 2>&1; tail -14 "$OUT/r05_lstm_block_bwd_ab.txt"
echo "== phases (var 0, var 7)"
for v in 0 4 7; do VARS=$v ROUNDS=1 HPC_RLL_LSTM_PROFILE=1 timeout 200 python tests/tools/r05_lstm_block_bwd_ab.py 2>&1 | grep -i "row-block bwd" | tail -4 | sed "s/^/var $v: /"; done > "$OUT/r05_lstm_block_bwd_phases.txt" 2>&1; cat "$OUT/r05_lstm_block_bwd_phases.txt"
echo "== gae gap probe"; timeout 300 python tests/tools/r05_gae_gap_probe.py > "$OUT/r05_gae_gap_probe.txt" 2>&1; cat "$OUT/r05_gae_gap_probe.txt"
(cd /tmp && export TMPDIR=/tmp && MODE=trace timeout 300 rocprofv3 --kernel-trace -d "$OUT/r05_gae_trace" -o trace -- python "$REPO/tests/tools/r05_gae_gap_probe.py" > "$OUT/r05_gae_trace.log" 2>&1)
DB=$(find "$OUT/r05_gae_trace" -name "*.db" | head -1)
echo "db: $DB"; MODE=gaps python tests/tools/r05_gae_gap_probe.py "$DB" > "$OUT/r05_gae_gaps.txt" 2>&1; cat "$OUT/r05_gae_gaps.txt"
rm -rf "$OUT/r05_gae_trace"
echo "== tests"
timeout 900 python -m pytest tests/test_lstm_gpu.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py -m gpu -x -q -k "lstm or c4" > "$OUT/r05_pytest_lstm.log" 2>&1; tail -15 "$OUT/r05_pytest_lstm.log"
