#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r03_run13.txt
{
echo "== tests =="
timeout 2400 python -m pytest tests/test_blocks_gpu.py tests/test_conv_ws_gpu.py tests/test_ops_gpu.py tests/test_train_step_gpu.py tests/test_network_gpu.py -m gpu -q -x 2>&1 | tail -12
echo "== bench A/B: lazy residual gradient (default) vs CSS_NO_LAZY_RES=1 =="
for v in lazy mat lazy mat; do
  if [ $v = mat ]; then export CSS_NO_LAZY_RES=1; else unset CSS_NO_LAZY_RES; fi
  python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/r03_b13_$v.json 2>> gpurun_out/r03_b13.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03_b13_$v.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], d["losses"], {k: (v["ms_per_step"], v["frac"]) for k, v in d["kernels"].items() if k.startswith("bn") or "ws" in k})
PY
done
} > $O 2>&1
tail -14 $O | cut -c1-400
