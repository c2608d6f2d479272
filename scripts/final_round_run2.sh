#!/bin/bash
# Round-end validation, second pass (kernels unchanged since scripts/final_round_run.sh collected the profiles): every -m gpu test, the
# default bench line (replaying the round's PMC traffic file), the 1-rank RCCL line, smoke().  Usage: bash scripts/final_round_run2.sh r02
TAG=${1:-r02}
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/${TAG}_gpu_tests.txt
tail -3 gpurun_out/${TAG}_gpu_tests.txt
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
cut -c1-330 gpurun_out/${TAG}_bench_line.json
CSS_FORCE_COLLECTIVES=1 python bench.py --no-cpu-baseline --no-extra > gpurun_out/${TAG}_force_coll_line.json 2>> gpurun_out/${TAG}_bench.err
cut -c1-330 gpurun_out/${TAG}_force_coll_line.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 > gpurun_out/${TAG}_smoke.txt
tail -3 gpurun_out/${TAG}_smoke.txt
# grid-size A/B of bn_apply (environment switch, same box)
for b in 256 8192; do
  CSS_BN_APPLY_BLOCKS=$b python bench.py --no-cpu-baseline --no-extra > gpurun_out/${TAG}_bnapply_$b.json 2>> gpurun_out/${TAG}_bench.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bnapply_$b.json").read().strip().splitlines()[-1])
print("CSS_BN_APPLY_BLOCKS=$b", d["value"], d["ms_per_step"], d["kernels"]["bn_apply"])
PY
done
