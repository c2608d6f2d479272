#!/bin/bash
# Round 3: conv_wgrad_p8_kernel against conv_wgrad_dma256_kernel (harness, alternating processes), parity tests, step-level A/B; trajectory test output
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r03_run10.txt
{
echo "== wgrad kernels (CSS_WGRAD_KERNEL: 0 lockstep, 1 half-step stagger, 2 two-phase p8), alternating processes =="
for sh in 0 1 3 5 6; do
  for v in 2 1 0 2 1 0; do
    echo -n "kernel $v: "; CSS_WGRAD_KERNEL=$v CB_UNIFORM=1 CB_ONLY=$sh timeout 300 ./build/cb_new | grep wgrad
  done
done
echo "== tests =="
timeout 2400 python -m pytest tests/test_conv_bench_scale_gpu.py tests/test_kernel_switches_gpu.py -m gpu -q -x 2>&1 | tail -5
timeout 900 python -m pytest tests/test_bf16_trajectory_gpu.py -m gpu -q -s 2>&1 | grep -v "^EMA" | tail -50
echo "== bench A/B: wgrad kernel 2 (p8) vs 1 =="
for v in 2 1 2 1; do
  CSS_WGRAD_KERNEL=$v python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/r03_wg2_$v.json 2>> gpurun_out/r03_wg2.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03_wg2_$v.json").read().strip().splitlines()[-1])
print("wgrad kernel $v", d["value"], d["ms_per_step"], {k: (v["ms_per_step"], v["frac"]) for k, v in d["kernels"].items() if "wgrad" in k})
PY
done
} > $O 2>&1
tail -12 $O
