#!/bin/bash
# Round 3: conv_igemm_p8_kernel - parity / race screen / timing against pp64 (harness), bench-scale parity tests, step-level A/B on one box.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r03_p8.txt
{
echo "== p8_bench =="
timeout 900 ./build/p8_bench
echo "== tests: conv bench scale + switches + new loader test =="
timeout 1500 python -m pytest tests/test_conv_bench_scale_gpu.py tests/test_kernel_switches_gpu.py tests/test_loader_step_gpu.py -m gpu -x -q 2>&1 | tail -15
echo "== bench A/B (same box): p8 (default) vs CSS_NO_P8_CONV=1 =="
for v in p8 pp64 p8 pp64; do
  if [ $v = pp64 ]; then export CSS_NO_P8_CONV=1; else unset CSS_NO_P8_CONV; fi
  python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/r03_ab_$v.json 2>> gpurun_out/r03_ab.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03_ab_$v.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], {k: (v["ms_per_step"], v["frac"]) for k, v in d["kernels"].items() if k.startswith("conv")})
PY
done
} > $O 2>&1
tail -30 $O
