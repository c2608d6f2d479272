#!/bin/bash
# Round 3: conv_igemm_p8_kernel, final form: full race screen, conv parity tests, step-level A/B against pp64 on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r03_run9.txt
{
echo "== p8_bench (50-launch race screen) =="
P8_RACE=50 timeout 1200 ./build/p8_bench
echo "== tests =="
timeout 2400 python -m pytest tests/test_conv_bench_scale_gpu.py tests/test_kernel_switches_gpu.py tests/test_conv_ws_gpu.py tests/test_blocks_gpu.py tests/test_network_gpu.py tests/test_bf16_trajectory_gpu.py tests/test_loader_step_gpu.py -m gpu -q -x 2>&1 | tail -8
echo "== bench A/B (same box): p8 (default) vs CSS_NO_P8_CONV=1 =="
for v in p8 pp64 p8 pp64; do
  if [ $v = pp64 ]; then export CSS_NO_P8_CONV=1; else unset CSS_NO_P8_CONV; fi
  python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/r03_ab2_$v.json 2>> gpurun_out/r03_ab2.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03_ab2_$v.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], {k: (v["ms_per_step"], v["frac"]) for k, v in d["kernels"].items() if k.startswith("conv")})
PY
done
} > $O 2>&1
tail -12 $O
