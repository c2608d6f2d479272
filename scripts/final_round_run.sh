#!/bin/bash
# Round-end validation on the GPU box: every -m gpu test, the round's profiles, the default bench line, the 769^2 lines, smoke().
# Usage: bash scripts/final_round_run.sh r02     (outputs under gpurun_out/, copied into profiles/ afterwards)
TAG=${1:-r02}
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/${TAG}_gpu_tests.txt
tail -3 gpurun_out/${TAG}_gpu_tests.txt
bash scripts/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
cut -c1-400 gpurun_out/${TAG}_bench_line.json
python bench.py --workload c4 --no-cpu-baseline --no-extra > gpurun_out/${TAG}_c4_bench_line.json 2>> gpurun_out/${TAG}_bench.err
cut -c1-300 gpurun_out/${TAG}_c4_bench_line.json
python bench.py --workload c5 --no-cpu-baseline --no-extra > gpurun_out/${TAG}_c5_bench_line.json 2>> gpurun_out/${TAG}_bench.err
cut -c1-300 gpurun_out/${TAG}_c5_bench_line.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 > gpurun_out/${TAG}_smoke.txt
tail -4 gpurun_out/${TAG}_smoke.txt
ls gpurun_out/${TAG}_*
