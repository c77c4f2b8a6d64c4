#!/bin/bash
timeout 400 python tests/tools/r05_lstm_bwd_bn_probe.py 2>&1 | grep "key26\|identical\|Error\|error" 
for k in 9 25; do KEYS=$k ROUNDS=1 HPC_RLL_LSTM_PROFILE=1 timeout 200 python tests/tools/r05_lstm_bwd_bn_probe.py 2>&1 | grep "row-block bwd" | tail -2 | sed "s/^/key $k: /"; done
