cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_lstm_gpu.py -q -p no:cacheprovider -x 2>&1 | grep -v amdgpu | tail -12
timeout 900 python -m pytest tests/test_full_size_gpu.py -q -p no:cacheprovider -k "lstm" -x 2>&1 | grep -v amdgpu | tail -5
for k in 2 1 2 1; do echo "key25=$k"; HPC_RLL_TUNE=25:$k timeout 600 python tests/tools/bench_suite.py c4 2>&1 | grep -v amdgpu | tail -1 | cut -c1-330; done
