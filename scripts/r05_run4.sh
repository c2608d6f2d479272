#!/bin/bash
# round 5, GPU call 4: the partitioned backward - correctness (bit identity), then A/B in alternating processes at c2 and c4
python -m pytest -m gpu -x -q -s tests/test_partition_gpu.py > gpurun_out/r05_run4_tests.txt 2>&1; tail -4 gpurun_out/r05_run4_tests.txt
O=gpurun_out/r05_partition_ab.txt; : > $O
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d['kernels']
    print(d['ms_per_step'], d['value'], {n: k[n]['ms_per_step'] for n in ('conv_wgrad_all_kernels','bn_bwd_apply','bn_bwd_reduce','conv_dgrad_all_kernels')})
except Exception as e:
    print('FAILED', e)
PY
}
for rep in 1 2; do
  for wl in c2 c4; do
    for v in 0 192 160; do
      CSS_BWD_PARTITION=$v timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_tmp.err
      echo -n "PARTITION=$v $wl: " >> $O; line gpurun_out/ab_tmp.json >> $O
    done
  done
done
cat $O
