#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_pad_scatter_gpu.py tests/test_edge_cases_gpu.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py -m gpu -x -q -k "pad or group or Pad or c5" 2>&1 | tail -4
bash tests/tools/r05_group_pad_profile.sh 2>&1 | tail -16
