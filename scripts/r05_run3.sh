#!/bin/bash
# round 5, GPU call 3: partition microbenchmark (one split per process, bounded), conv_ws4_kernel experiments, the calm-regime trajectory test
P=gpurun_out/r05_partition_bench2.txt; : > $P
for X in 192 160 128; do
  echo "=== PB_X=$X ===" >> $P
  PB_X=$X timeout 150 ./build/partition_bench >> $P 2>&1; echo "rc=$?" >> $P
done
tail -30 $P
bash scripts/r05_ws4_run1.sh
python -m pytest -m gpu -x -q -s tests/test_bf16_trajectory_gpu.py::test_bf16_tracks_fp32_in_a_calm_regime tests/test_conv_ws_gpu.py > gpurun_out/r05_run3_tests.txt 2>&1; tail -3 gpurun_out/r05_run3_tests.txt
