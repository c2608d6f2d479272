#!/bin/bash
# round 4, run 7: the GPU suite on the tree with the tiled pseudo-label kernel / register-run class sums / eight-member trajectory test; short bench + kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_run7_suite.txt 2>&1
echo "suite rc=$?" | tee -a gpurun_out/r04_run7_suite.txt
tail -4 gpurun_out/r04_run7_suite.txt
timeout 600 python -m pytest tests/test_bf16_trajectory_gpu.py -q -s -m gpu > gpurun_out/r04_run7_trajectory.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extra > /tmp/prof_stats.log 2>&1
S=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $GRAFT_REPO_ROOT/gpurun_out/r04_run7_kernel_stats.csv
grep -E "pseudo_label|class_sums|contrast_scan|maxpool_bwd|colsum|nchw_to_nhwc" $GRAFT_REPO_ROOT/gpurun_out/r04_run7_kernel_stats.csv | cut -c1-160
