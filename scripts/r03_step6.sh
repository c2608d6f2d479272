#!/bin/bash
# round 3, step 6: masked residual gradient (css_conv2d_dgrad_add_masked) + row order of the batch-norm passes
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03_step6.txt
echo "== tests ==" > $OUT
timeout 1500 python -m pytest tests/test_dgrad_add_masked_gpu.py tests/test_bf16_trajectory_gpu.py tests/test_conv_ws_gpu.py tests/test_ops_gpu.py tests/test_train_step_gpu.py tests/test_kernel_switches_gpu.py -m gpu -q -x 2>&1 | tail -25 >> $OUT
echo "== lazy (default) vs eager residual-gradient mask ==" >> $OUT
for round in 1 2; do
  for e in 0 1; do
    CSS_BN_EAGER_DRES=$e timeout 300 python bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('eager_dres $e', d['value'], d['ms_per_step'], d['losses'], {n:(v.get('ms_per_step'), v.get('frac')) for n,v in k.items() if n.startswith('bn_') or 'dgrad' in n})
" >> $OUT 2>&1
  done
done
ORDERS="0 1 2 4 3 5" bash scripts/r03_bn_order.sh > /dev/null 2>&1
cat gpurun_out/r03_bn_pass_order.txt >> $OUT
tail -40 $OUT
