#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03_bn_red_blocks.txt
echo "== CSS_BN_RED_BLOCKS (row blocks of bn_bwd_reduce / bn_stats) in the step ==" > $OUT
for r in 1 2; do for b in 1024 2048 4096 512; do
  CSS_BN_RED_BLOCKS=$b timeout 300 python bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('blocks $b', d['value'], d['ms_per_step'], {n:(v.get('ms_per_step'), v.get('frac')) for n,v in k.items() if n.startswith('bn_')})
" >> $OUT 2>&1
done; done
cat $OUT
