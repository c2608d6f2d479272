#!/bin/bash
# round 4, run 3: who issues the small copies / per-layer layout calls of a step; the GPU suite on the pruned tree; the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python scripts/r04_copy_hunt.py > gpurun_out/r04_copy_hunt.txt 2>&1
echo "hunt rc=$?"
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_run3_suite.txt 2>&1
echo "suite rc=$?" | tee -a gpurun_out/r04_run3_suite.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04_run3_bench.json 2> gpurun_out/r04_run3_bench.err
echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extra > /tmp/prof_stats.log 2>&1
S=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $GRAFT_REPO_ROOT/gpurun_out/r04_run3_kernel_stats.csv
tail -3 $GRAFT_REPO_ROOT/gpurun_out/r04_run3_suite.txt; cut -c1-300 $GRAFT_REPO_ROOT/gpurun_out/r04_run3_bench.json
