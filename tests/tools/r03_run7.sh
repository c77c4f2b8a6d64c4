cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
for L in tests/tools/micro/libhpc_rll_hip_v2.so ""; do
  echo "== lib ${L:-current}"
  HPC_RLL_LIB=$L PROBE_B=262144,131072,32768 PROBE_SW=0,8,32,64 python tests/tools/r03_batch_probe.py 2>&1 | grep qrdqn | cut -c1-48
done
done
