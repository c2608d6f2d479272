#!/bin/bash
# round 4: A/B of two LDS stages for the 128x128 convolution kernel when there are more workgroups than CUs (CSS_SMALL_NST2), c2 and c4,
# alternating processes; the forced-collectives lines after the peer kernel took its own term from `local`
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r04_small_nst2_ab.txt; : > $O
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], {k: v["ms_per_step"] for k, v in j["kernels"].items() if k in ("conv_fwd_all_kernels", "conv_dgrad_all_kernels")})
PY
}
for rep in 1 2; do
  for v in 0 1; do
    for wl in c2 c4; do
      CSS_SMALL_NST2=$v timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_tmp.err
      echo -n "NST2=$v $wl: " >> $O; line gpurun_out/ab_tmp.json >> $O
    done
  done
done
cat $O
for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_tmp.err; echo -n "plain: "; line gpurun_out/ab_tmp.json
  CSS_FORCE_COLLECTIVES=1 CSS_SYNCBN=peer timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_tmp.err; echo -n "forced peer: "; line gpurun_out/ab_tmp.json
done
timeout 600 python -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "peer or rccl_collectives" 2>&1 | tail -2
