#!/bin/bash
# Round 3, call 3: gemm8p / pp64 / p8 interleaved in ONE process on the same GEMMs; weight-gradient stagger A/B; new tests.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r03_run3.txt
{
echo "== p8_bench: GEMM shapes, three kernels interleaved =="
P8_FROM=10 P8_RACE=5 timeout 900 ./build/p8_bench
echo "== wgrad stagger A/B (alternating processes, per shape) =="
for sh in 0 1 3 5; do
  for v in stag nostag stag nostag; do
    if [ $v = nostag ]; then export CSS_WGRAD_NOSTAGGER=1; else unset CSS_WGRAD_NOSTAGGER; fi
    echo -n "$v: "; CB_UNIFORM=1 CB_ONLY=$sh timeout 300 ./build/cb_new | grep wgrad
  done
done
unset CSS_WGRAD_NOSTAGGER
echo "== tests =="
timeout 2400 python -m pytest tests/test_conv_bench_scale_gpu.py tests/test_loader_step_gpu.py tests/test_bf16_trajectory_gpu.py -m gpu -q -s 2>&1 | grep -v "^EMA\|^$" | tail -60
echo "== bench A/B: wgrad stagger =="
for v in stag nostag stag nostag; do
  if [ $v = nostag ]; then export CSS_WGRAD_NOSTAGGER=1; else unset CSS_WGRAD_NOSTAGGER; fi
  CSS_NO_P8_CONV=1 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/r03_wg_$v.json 2>> gpurun_out/r03_wg.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03_wg_$v.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], {k: (v["ms_per_step"], v["frac"]) for k, v in d["kernels"].items() if "wgrad" in k})
PY
done
} > $O 2>&1
tail -40 $O
