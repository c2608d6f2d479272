#!/bin/bash
# A/B of ONE environment switch of the library in alternating processes on one box (every per-step figure of profiles/*_ab.txt since round 3
# was taken this way; the one-off scripts/r0N_run*.sh of rounds 3-4 were copies of this loop - VERDICT r04 item 8).
#   bash scripts/ab_env.sh VAR "v1 v2 ..." ["c2 c4"] [steps] [reps] [out]      e.g.   bash scripts/ab_env.sh CSS_SMALL_NST2 "0 1" "c2 c4" 10 2
# Prints, per (value, workload, repetition): ms per step, images/s and the per-class kernel times of the profiled step.
VAR=$1; VALS=${2:-"0 1"}; WLS=${3:-"c2"}; STEPS=${4:-10}; REPS=${5:-2}
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
O=${6:-gpurun_out/ab_${VAR}.txt}; : > $O
line() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = j["kernels"]
    print(j["ms_per_step"], j["value"], {n: k[n]["ms_per_step"] for n in ("conv_igemm_p8_and_ws_kernels", "conv_ws_kernel", "conv_fwd_all_kernels", "conv_dgrad_all_kernels",
                                                                        "conv_wgrad_all_kernels", "bn_apply", "bn_bwd_apply", "bn_bwd_reduce") if n in k})
except Exception as e:
    print("FAILED", e)
PY
}
for rep in $(seq 1 $REPS); do
  for wl in $WLS; do
    for v in $VALS; do
      env "$VAR=$v" timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps $STEPS --warmup $STEPS > gpurun_out/ab_tmp.json 2>> gpurun_out/ab_tmp.err
      echo -n "$VAR=$v $wl: " >> $O; line gpurun_out/ab_tmp.json >> $O
    done
  done
done
cat $O
