#!/bin/bash
# round 4, run 4: weight-gradient plan A/B on the harness (r03 build vs this tree, alternating processes), the copy hunt, the peer-exchange
# and distributed tests, the conv / determinism tests on the new plan, a short bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r04_wgrad_plan_ab.txt
echo "== conv_bench wgrad (slab workspace + ordered reduce in both builds): r03 plan (whole rounds of every XCD) vs r04 plan (whole rounds of the chip, fewest slabs), alternating processes ==" > $O
for rep in 1 2; do
  for b in cb_r03 cb_r04; do
    for sh in 0 1 3 5 6; do
      echo -n "$b: " >> $O; CB_UNIFORM=1 CB_ONLY=$sh timeout 120 ./build/$b 2>&1 | grep wgrad >> $O
    done
  done
done
cat $O
timeout 600 python scripts/r04_copy_hunt.py > gpurun_out/r04_copy_hunt.txt 2>&1; echo "hunt rc=$?"
timeout 1800 python -m pytest tests/test_dist_gpu.py tests/test_conv_bench_scale_gpu.py tests/test_determinism_gpu.py tests/test_ops_gpu.py tests/test_kernel_switches_gpu.py -x -q -m gpu > gpurun_out/r04_run4_tests.txt 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r04_run4_tests.txt
tail -5 gpurun_out/r04_run4_tests.txt
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/r04_run4_bench_$i.json 2>> gpurun_out/r04_run4_bench.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/r04_run4_bench_$i.json").read().strip().splitlines()[-1])
print("bench", j["ms_per_step"], j["value"], {k:v["ms_per_step"] for k,v in j["kernels"].items() if "wgrad" in k})
PY
done
