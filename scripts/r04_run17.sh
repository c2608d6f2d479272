#!/bin/bash
# round 4: A/B of two LDS stages (three workgroups per CU) for the 128x64 kernel of the Cout <= 64 layers (CSS_WGRAD_SMALL_BP32), c2 and c4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r04_wgrad_small_bp32_ab.txt; : > $O
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], {k: v["ms_per_step"] for k, v in j["kernels"].items() if k in ("conv_wgrad_all_kernels", "conv_fwd_all_kernels")})
PY
}
for rep in 1 2; do
  for v in 0 1; do
    for wl in c2 c4; do
      CSS_WGRAD_SMALL_BP32=$v timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_tmp.err
      echo -n "WGRAD_SMALL_BP32=$v $wl: " >> $O; line gpurun_out/ab_tmp.json >> $O
    done
  done
done
cat $O
CSS_WGRAD_SMALL_BP32=1 timeout 900 python -m pytest tests/test_conv_bench_scale_gpu.py tests/test_ops_gpu.py -k "wgrad or conv" -x -q -m gpu 2>&1 | tail -2
