#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r03_run11.txt
{
echo "== stamps (prologue after the argument hoist) =="
timeout 200 ./build/p8_stamp
echo "== p8_bench =="
P8_RACE=20 timeout 1200 ./build/p8_bench | grep -v "one workgroup"
echo "== tests =="
timeout 2400 python -m pytest tests/test_conv_bench_scale_gpu.py tests/test_kernel_switches_gpu.py tests/test_blocks_gpu.py tests/test_bf16_trajectory_gpu.py -m gpu -q -x 2>&1 | tail -5
echo "== bench =="
for v in a b; do
  python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/r03_b11_$v.json 2>> gpurun_out/r03_b11.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03_b11_$v.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], {k: (v["ms_per_step"], v["frac"]) for k, v in d["kernels"].items()})
PY
done
} > $O 2>&1
tail -6 $O
