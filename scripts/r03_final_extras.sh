#!/bin/bash
# Extra evidence on the final tree: the 8-phase kernel's parity / race screen / timing harness, and the c4 workload's kernel stats
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROOT=$(pwd)
P8_ROUNDS=3 timeout 600 ./build/p8_bench > gpurun_out/r03_p8_harness_final_tree.txt 2>&1
tail -5 gpurun_out/r03_p8_harness_final_tree.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c4
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o s -- python3 $ROOT/bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /tmp/prof_c4.log 2>&1
S=$(find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $ROOT/gpurun_out/r03_c4_kernel_stats.csv
head -8 $ROOT/gpurun_out/r03_c4_kernel_stats.csv | cut -c1-160
