#!/bin/bash
# round 4: the GPU suite on the final tree, once more on a fresh box (consecutive green runs); then the conv_ws write-pattern question:
# 256 -> 256 / 512 into 1024-wide rows against 256 -> 256 into 256-wide rows and 256 -> 1024 (scripts/ws_bench.hip shapes 10-12, 0)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=${1:-a}
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_final_suite_$TAG.txt 2>&1
echo "suite rc=$?" | tee -a gpurun_out/r04_final_suite_$TAG.txt
tail -3 gpurun_out/r04_final_suite_$TAG.txt
if [ "$TAG" = "a" ]; then
  O=gpurun_out/r04_ws_pitch.txt
  echo "== conv_ws_kernel: output pitch / panel count (plain epilogue), two repetitions ==" > $O
  for rep in 1 2; do for sh in 0 10 11 12; do WB_ONLY=$sh timeout 120 ./build/ws_bench 2>&1 | grep -E "plain" | cut -c1-150 >> $O; done; done
  cat $O
fi
