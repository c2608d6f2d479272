#!/bin/bash
# round 4: A/B of two LDS stages (three workgroups per CU) for the 128x64 kernel of the Cout <= 64 layers (CSS_WGRAD_STREAM), c2 and c4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r04_wgrad_stream_ab.txt; : > $O
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], {k: v["ms_per_step"] for k, v in j["kernels"].items() if k in ("conv_wgrad_all_kernels", "bn_bwd_apply")})
PY
}
for rep in 1 2; do
  for v in 0 1; do
    for wl in c2 c4; do
      CSS_WGRAD_STREAM=$v timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_tmp.err
      echo -n "WGRAD_STREAM=$v $wl: " >> $O; line gpurun_out/ab_tmp.json >> $O
    done
  done
done
cat $O
CSS_WGRAD_STREAM=1 timeout 900 python -m pytest tests/test_determinism_gpu.py tests/test_train_step_gpu.py -x -q -m gpu 2>&1 | tail -2
