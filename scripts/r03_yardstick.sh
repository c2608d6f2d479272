#!/bin/bash
# Round 3, first GPU call: the guide's 8-phase GEMM next to conv_igemm_pp64_kernel on the same GEMMs (same box, uniform random operands),
# the vector-memory-path micro-benchmark and the conv_ws2 prototype.  Output -> gpurun_out/r03_yardstick.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r03_yardstick.txt
{
echo "== gemm8p (yardstick) =="
timeout 600 ./build/gemm8p
echo "== pp64 on the same GEMMs (1x1 conv, 32 x 64^2 = 131072 rows; uniform operands) =="
CB_UNIFORM=1 CB_NOWGRAD=1 CB_SHAPE="32,64,64,1024,256,1,1,0,1;32,64,64,2304,256,1,1,0,1;32,64,64,4608,256,1,1,0,1;32,64,64,18432,256,1,1,0,1;32,64,64,1024,512,1,1,0,1;32,64,64,2304,512,1,1,0,1;32,64,64,4608,512,1,1,0,1;16,64,64,4096,4096,1,1,0,1" timeout 600 ./build/cb_new
echo "== conv table, uniform operands =="
CB_UNIFORM=1 timeout 600 ./build/cb_new
echo "== conv table, uniform operands, with statistics =="
CB_UNIFORM=1 CB_NOWGRAD=1 CB_STATS=1 timeout 600 ./build/cb_new
echo "== ta_path_bench =="
timeout 300 ./build/ta_path_bench
echo "== ws2_bench =="
timeout 300 ./build/ws2_bench
} > $O 2>&1
tail -5 $O
