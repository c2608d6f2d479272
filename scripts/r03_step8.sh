#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03_step8.txt
echo "== tests ==" > $OUT
timeout 1500 python -m pytest tests/test_losses_gpu.py tests/test_train_step_gpu.py "tests/test_kernel_switches_gpu.py::test_ce_gather_kernel_switch_is_a_shipped_configuration" -m gpu -q -x 2>&1 | tail -8 >> $OUT
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_stats
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /tmp/prof_stats.log 2>&1
S=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $ROOT/gpurun_out/r03_step8_kernel_stats.csv
cd $ROOT
echo "== CE kernels ==" >> $OUT
grep "ce_small\|ce_kernel" gpurun_out/r03_step8_kernel_stats.csv | sed 's/([^)]*)//' | cut -c1-200 >> $OUT
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['losses'])
" >> $OUT 2>&1
done
tail -30 $OUT
