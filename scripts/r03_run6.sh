#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r03_run6.txt
{
for b in p8_bench p8_bench_nostore; do
  echo "== $b (P8_NOPRIO) =="
  P8_RACE=2 timeout 900 ./build/$b | grep "plain\|gemm8p\|p8_bench\|one workgroup"
done
} > $O 2>&1
tail -3 $O
