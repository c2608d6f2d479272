#!/bin/bash
# Round-end evidence, collected from ONE source tree (scripts/source_hash.py digest stored beside it): the PMC / kernel-stats passes first
# (the default bench line replays the traffic files they produce), then the bench lines, the forced-collectives lines (RCCL and the peer
# exchange in loop-back), smoke(), and the whole GPU suite.  Usage: bash scripts/final_round_run.sh r05   (TAG = the round: every output is gpurun_out/<TAG>_*)
TAG=${1:-r05}
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
# what the collective CALLS of a step cost at one rank (no wire): no group / RCCL / peer exchange, twice, alternating
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-extra > gpurun_out/${TAG}_bench_line_plain_$rep.json 2>> gpurun_out/${TAG}_bench.err
  CSS_FORCE_COLLECTIVES=1 python bench.py --no-cpu-baseline --no-extra > gpurun_out/${TAG}_force_coll_line_$rep.json 2>> gpurun_out/${TAG}_bench.err
  CSS_FORCE_COLLECTIVES=1 CSS_SYNCBN=peer python bench.py --no-cpu-baseline --no-extra > gpurun_out/${TAG}_force_coll_peer_line_$rep.json 2>> gpurun_out/${TAG}_bench.err
done
python - <<PY
import json
for n in ("bench_line_plain_1", "force_coll_line_1", "force_coll_peer_line_1", "bench_line_plain_2", "force_coll_line_2", "force_coll_peer_line_2"):
    try:
        j = json.loads(open("gpurun_out/${TAG}_" + n + ".json").read().strip().splitlines()[-1])
        print(n, j["ms_per_step"], j["value"])
    except Exception as e:
        print(n, "failed", e)
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 > gpurun_out/${TAG}_smoke.txt
tail -3 gpurun_out/${TAG}_smoke.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_final_suite.txt 2>&1
echo "suite rc=$?" | tee -a gpurun_out/${TAG}_final_suite.txt
tail -3 gpurun_out/${TAG}_final_suite.txt
ls -la gpurun_out/${TAG}_* | head -40
cat gpurun_out/${TAG}_profile_errors.txt 2>/dev/null
