#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r03_run12.txt
{
echo "== tests =="
timeout 2400 python -m pytest tests/test_ops_gpu.py tests/test_blocks_gpu.py tests/test_train_step_gpu.py tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -6
echo "== bench A/B: two-level stage 2 (default) vs CSS_BN_STAGE2_ONE_LEVEL=1 =="
for v in two one two one; do
  if [ $v = one ]; then export CSS_BN_STAGE2_ONE_LEVEL=1; else unset CSS_BN_STAGE2_ONE_LEVEL; fi
  python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 10 > gpurun_out/r03_b12_$v.json 2>> gpurun_out/r03_b12.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03_b12_$v.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], d["losses"])
PY
done
unset CSS_BN_STAGE2_ONE_LEVEL
echo "== kernel stats (rocprofv3) of 3 steps =="
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_prof12 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r03_prof12/*/*kernel_stats.csv | head -1); head -45 $f | cut -c1-200
} > $O 2>&1
tail -50 $O | cut -c1-220
