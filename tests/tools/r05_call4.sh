#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
timeout 300 python tests/tools/r05_lstm_skew_probe.py 2>&1 | grep variant | tee "$OUT/r05_lstm_skew_probe.txt"
