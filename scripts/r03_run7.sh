#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r03_run7.txt
{
echo "== p8_bench (P8_NOPRIO, final wait ahead of the last tile's stores) =="
P8_RACE=2 timeout 900 ./build/p8_bench | grep "plain\|gemm8p\|p8_bench\|one workgroup"
} > $O 2>&1
tail -3 $O
