#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
echo "== scatter variants"; timeout 200 tests/tools/micro/scatter_sweep.bin > "$OUT/r05_scatter_sweep3.txt" 2>&1; grep "L9\|576\|^S0 \|^L0" "$OUT/r05_scatter_sweep3.txt"
