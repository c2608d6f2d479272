#!/bin/bash
# Round 3, call 4: p8 with the offset arithmetic inside its MFMA segments (V2), with / without s_setprio; pp64 without s_setprio; lr sweep of the trajectory tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r03_run4.txt
{
for b in p8_bench p8_bench_noprio p8_bench_p6noprio; do
  echo "== $b =="
  P8_RACE=3 timeout 900 ./build/$b
done
echo "== trajectory tests: lr sweep =="
for lr in 6.4e-3 1e-3 2e-4; do
  echo "--- CSS_TRAJ_LR=$lr"
  CSS_TRAJ_LR=$lr timeout 900 python -m pytest tests/test_bf16_trajectory_gpu.py -m gpu -q -s -k thirty 2>&1 | grep "^step\|30 steps\|passed\|failed\|Error"
done
for lr in 1e-5 1e-6; do
  echo "--- CSS_TRAJ_LR3=$lr"
  CSS_TRAJ_LR3=$lr timeout 900 python -m pytest tests/test_bf16_trajectory_gpu.py -m gpu -q -s -k three 2>&1 | grep "step\|cosine\|passed\|failed\|Error"
done
} > $O 2>&1
tail -5 $O
