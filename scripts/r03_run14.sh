#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r03_run14.txt
{
echo "== tests =="
timeout 2400 python -m pytest tests/test_conv_bench_scale_gpu.py tests/test_kernel_switches_gpu.py tests/test_blocks_gpu.py tests/test_ops_gpu.py tests/test_network_gpu.py -m gpu -q -x 2>&1 | tail -8
echo "== bench A/B: two K groups in the leftover kernel (default) vs CSS_NO_SMALL_SPLITK=1 =="
for v in split one split one; do
  if [ $v = one ]; then export CSS_NO_SMALL_SPLITK=1; else unset CSS_NO_SMALL_SPLITK; fi
  python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/r03_b14_$v.json 2>> gpurun_out/r03_b14.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03_b14_$v.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], d["losses"], {k: (v["ms_per_step"], v["frac"]) for k, v in d["kernels"].items() if "other" in k or "all" in k})
PY
done
} > $O 2>&1
tail -12 $O | cut -c1-400
