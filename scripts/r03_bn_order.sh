#!/bin/bash
# A/B of the row order of the batch-norm streaming passes (CSS_BN_PASS_ORDER bits: 1 bn_apply downwards, 2 bn_bwd_reduce, 4 bn_bwd_apply),
# alternating processes on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03_bn_pass_order.txt
echo "== CSS_BN_PASS_ORDER A/B (images/s, ms per step; kernels: bn_apply, bn_bwd_reduce, bn_bwd_apply ms per step) ==" > $OUT
for round in 1 2; do
  for o in ${ORDERS:-0 1 2 4 3 5}; do
    CSS_BN_PASS_ORDER=$o timeout 300 python bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('order $o', d['value'], d['ms_per_step'], {n:(v.get('ms_per_step'), v.get('frac')) for n,v in k.items() if n.startswith('bn_')})
" >> $OUT 2>&1
  done
done
cat $OUT
