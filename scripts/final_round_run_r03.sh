#!/bin/bash
# Round-end evidence, collected from ONE source tree (scripts/source_hash.py digest stored beside it): the PMC / kernel-stats passes first
# (the default bench line replays the traffic files they produce), then the bench lines, smoke().  Usage: bash scripts/final_round_run_r03.sh r03
TAG=${1:-r03}
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROOT=$(pwd)
mkdir -p gpurun_out
rm -f gpurun_out/${TAG}_profile_errors.txt
bash scripts/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
cd $ROOT
# the bench reads profiles/<tag>_pmc_hbm_traffic_*.csv: hand it this run's files
cp gpurun_out/${TAG}_pmc_hbm_traffic_c2.csv gpurun_out/${TAG}_pmc_hbm_traffic_c4.csv profiles/ 2>/dev/null
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
cut -c1-400 gpurun_out/${TAG}_bench_line.json
python bench.py --workload c4 --no-cpu-baseline > gpurun_out/${TAG}_c4_bench_line.json 2>> gpurun_out/${TAG}_bench.err
cut -c1-300 gpurun_out/${TAG}_c4_bench_line.json
python bench.py --workload c5 --no-cpu-baseline > gpurun_out/${TAG}_c5_bench_line.json 2>> gpurun_out/${TAG}_bench.err
cut -c1-300 gpurun_out/${TAG}_c5_bench_line.json
CSS_FORCE_COLLECTIVES=1 python bench.py --no-cpu-baseline --no-extra > gpurun_out/${TAG}_force_coll_line.json 2>> gpurun_out/${TAG}_bench.err
cut -c1-300 gpurun_out/${TAG}_force_coll_line.json
python bench.py --no-cpu-baseline --no-extra > gpurun_out/${TAG}_bench_line_plain.json 2>> gpurun_out/${TAG}_bench.err
cut -c1-200 gpurun_out/${TAG}_bench_line_plain.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 > gpurun_out/${TAG}_smoke.txt
tail -3 gpurun_out/${TAG}_smoke.txt
ls -la gpurun_out/${TAG}_*
cat gpurun_out/${TAG}_profile_errors.txt 2>/dev/null
