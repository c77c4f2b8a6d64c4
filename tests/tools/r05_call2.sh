#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== scatter ablations"; timeout 200 tests/tools/micro/scatter_sweep.bin > "$OUT/r05_scatter_sweep2.txt" 2>&1; grep -v "^S2\|^S3\|^C1\|^C4\|   C" "$OUT/r05_scatter_sweep2.txt"
echo "== lstm bwd DMA ablation (var 5 = 2 buffers, no fences, 4-row chunks)"
for abl in 0 1 2 3; do HPC_RLL_BLK_ABL=$abl VARS=5 ROUNDS=1 HPC_RLL_LSTM_PROFILE=1 timeout 200 python tests/tools/r05_lstm_block_bwd_ab.py 2>&1 | grep -i "row-block bwd\|round 0" | tail -3 | sed "s/^/abl $abl: /"; done > "$OUT/r05_lstm_block_bwd_abl.txt" 2>&1; cat "$OUT/r05_lstm_block_bwd_abl.txt"
echo "== gae gap probe"; timeout 300 python tests/tools/r05_gae_gap_probe.py 2>&1 | grep -v Warning | grep -v "return Variable" > "$OUT/r05_gae_gap_probe.txt"; cat "$OUT/r05_gae_gap_probe.txt"
(cd /tmp && export TMPDIR=/tmp && MODE=trace timeout 300 rocprofv3 --kernel-trace -d "$OUT/r05_gae_trace" -o trace -- python "$REPO/tests/tools/r05_gae_gap_probe.py" > "$OUT/r05_gae_trace.log" 2>&1)
DB=$(find "$OUT/r05_gae_trace" -name "*.db" | head -1)
MODE=gaps python tests/tools/r05_gae_gap_probe.py "$DB" > "$OUT/r05_gae_gaps.txt" 2>&1; cat "$OUT/r05_gae_gaps.txt"
rm -rf "$OUT/r05_gae_trace"
echo "== tests"
timeout 600 python -m pytest tests/test_lstm_gpu.py -m gpu -x -q -k "starved or check_persistent or row_block" > "$OUT/r05_pytest_lstm2.log" 2>&1; tail -8 "$OUT/r05_pytest_lstm2.log"
