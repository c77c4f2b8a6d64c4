cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_losses_gpu.py tests/test_fuzz_gpu.py tests/test_edge_cases_gpu.py -q -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -25
timeout 900 python -m pytest tests/test_full_size_gpu.py -q -p no:cacheprovider -k "c51 or qrdqn or td or dist" 2>&1 | grep -v amdgpu | tail -8
