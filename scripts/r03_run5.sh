#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r03_run5.txt
{
for b in p8_bench p8_bench_noprio; do
  echo "== $b =="
  P8_RACE=3 timeout 900 ./build/$b | grep "plain\|gemm8p\|p8_bench"
done
} > $O 2>&1
tail -3 $O
