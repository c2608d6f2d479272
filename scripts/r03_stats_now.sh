#!/bin/bash
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_stats
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /tmp/prof_stats.log 2>&1
S=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $ROOT/gpurun_out/r03_now_kernel_stats.csv
cd $ROOT
for f in 0 1 0 1; do
  CSS_FORCE_COLLECTIVES=$f timeout 300 python bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('force_collectives $f', d['value'], d['ms_per_step'], d.get('rccl'))
" >> gpurun_out/r03_force_coll_now.txt 2>&1
done
cat gpurun_out/r03_force_coll_now.txt
head -50 gpurun_out/r03_now_kernel_stats.csv | cut -c1-200
