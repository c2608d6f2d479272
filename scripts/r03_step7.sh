#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03_step7.txt
echo "== tests ==" > $OUT
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_train_step_gpu.py tests/test_dist_gpu.py tests/test_models_gpu.py tests/test_full_size_gpu.py -m gpu -q -x 2>&1 | tail -8 >> $OUT
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_stats
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /tmp/prof_stats.log 2>&1
S=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $ROOT/gpurun_out/r03_step7_kernel_stats.csv
cd $ROOT
echo "== stage-2 kernels ==" >> $OUT
grep "bn_reduce\|wgrad_slab" gpurun_out/r03_step7_kernel_stats.csv | cut -c1-60,200-330 >> $OUT
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['losses'])
" >> $OUT 2>&1
done
tail -30 $OUT
