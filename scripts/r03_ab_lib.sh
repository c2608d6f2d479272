#!/bin/bash
# A/B of two builds of the library in alternating processes on one box: CSS_HIP_LIB=build/libcss_old.so against the tree's own
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03_ab_lib.txt
echo "== new (tree) vs old (build/libcss_old.so), bench --steps 10 --warmup 8 ==" > $OUT
for r in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export CSS_HIP_LIB=$PWD/build/libcss_old.so; else unset CSS_HIP_LIB; fi
    timeout 300 python bench.py --steps 10 --warmup 8 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('$v', d['value'], d['ms_per_step'], d['losses'], {n:(v.get('ms_per_step')) for n,v in k.items() if n in ('conv_ws_kernel','bn_apply','bn_bwd_apply','conv_igemm_pp_kernels','conv_fwd_all_kernels','conv_dgrad_all_kernels')})
" >> $OUT 2>&1
  done
done
unset CSS_HIP_LIB
cat $OUT
