#!/bin/bash
# A/B of several environment SETTINGS (each a quoted string of VAR=value pairs, "" = defaults) in alternating processes on one box:
#   bash scripts/ab_envs.sh OUT REPS STEPS WORKLOAD "CSS_A=1" "CSS_A=1 CSS_B=3" ""
# Prints per (setting, repetition): ms per step, images/s and the per-class kernel times of the profiled step (as scripts/ab_env.sh).
O=$1; REPS=$2; STEPS=$3; WL=$4; shift 4
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out; : > $O
for rep in $(seq 1 $REPS); do
  for s in "$@"; do
    env $s timeout 900 python bench.py --workload $WL --no-cpu-baseline --no-extra --steps $STEPS --warmup $STEPS > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_tmp.err
    echo -n "[$s] $WL: " >> $O
    python - >> $O <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
    k = j["kernels"]
    print(j["ms_per_step"], j["value"], {n: k[n]["ms_per_step"] for n in ("conv_fwd_all_kernels", "conv_dgrad_all_kernels", "conv_wgrad_all_kernels", "conv_ws_kernel", "bn_apply",
                                                                        "bn_bwd_apply", "bn_bwd_reduce") if n in k})
except Exception as e:
    print("FAILED", e)
PY
  done
done
cat $O
