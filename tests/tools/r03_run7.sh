cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_lstm_gpu.py tests/test_graphed_gpu.py tests/test_reference_lists_gpu.py tests/test_fuzz_gpu.py tests/test_dist.py tests/test_models_gpu.py -q -p no:cacheprovider -x -m gpu 2>&1 | grep -v amdgpu | tail -8
