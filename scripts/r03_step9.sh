#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03_step9.txt
echo "== tests (stem on the LDS-DMA kernel) ==" > $OUT
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_network_gpu.py tests/test_dist_gpu.py "tests/test_kernel_switches_gpu.py::test_eager_residual_gradient_switch_is_a_shipped_configuration" -m gpu -q -x 2>&1 | tail -6 >> $OUT
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export CSS_NO_STEM_DMA=1; else unset CSS_NO_STEM_DMA; fi
  timeout 300 python bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('no_stem_dma $v', d['value'], d['ms_per_step'], d['losses'])
" >> $OUT 2>&1
done
unset CSS_NO_STEM_DMA
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_stats
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /tmp/prof_stats.log 2>&1
S=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $ROOT/gpurun_out/r03_step9_kernel_stats.csv
cd $ROOT
grep "conv_igemm_dma_kernel\|conv_igemm_kernel\|conv_wgrad_kernel" gpurun_out/r03_step9_kernel_stats.csv | sed 's/([^)]*)//' | cut -c1-200 >> $OUT
tail -22 $OUT
