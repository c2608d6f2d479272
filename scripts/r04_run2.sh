#!/bin/bash
# round 4, run 2: the GPU suite on the tree with the ADVICE fixes, then the default bench line and a kernel-stats profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_run2_suite.txt 2>&1
echo "suite rc=$?" | tee -a gpurun_out/r04_run2_suite.txt
timeout 900 python bench.py > gpurun_out/r04_run2_bench.json 2> gpurun_out/r04_run2_bench.err
echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extra > /tmp/prof_stats.log 2>&1
S=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $GRAFT_REPO_ROOT/gpurun_out/r04_run2_kernel_stats.csv
tail -3 $GRAFT_REPO_ROOT/gpurun_out/r04_run2_suite.txt; cut -c1-400 $GRAFT_REPO_ROOT/gpurun_out/r04_run2_bench.json
