#!/bin/bash
# Round profiles of the default bench workload on the GPU box (each counter group is its own rocprofv3 run, every run bounded by
# `timeout`; FETCH_SIZE and WRITE_SIZE never share a pass; no trace domains next to --pmc).  Usage: scripts/collect_profiles.sh r02
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
# what these profiles were measured on (checked by tests/test_host_cpu.py against the working tree)
python3 $ROOT/scripts/source_hash.py $ROOT > $OUT/${TAG}_source_sha256.txt
B="python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra"
run() {   # name, counters...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 420 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/prof_$name -o p -- $B > /tmp/prof_$name.log 2>&1 || echo "$name: rocprofv3 failed or timed out" >> $OUT/${TAG}_profile_errors.txt
}
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
DB=$(find /tmp/prof_sq -name "*.db" | head -1); [ -n "$DB" ] && python3 $ROOT/scripts/pmc_sq_summary.py $DB 20 > $OUT/${TAG}_pmc_sq_mfma_busy.csv
run lds SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
DB=$(find /tmp/prof_lds -name "*.db" | head -1); [ -n "$DB" ] && python3 $ROOT/scripts/pmc_sq_summary.py $DB 20 > $OUT/${TAG}_pmc_sq_lds.csv
run fetch FETCH_SIZE
run write WRITE_SIZE
F=$(find /tmp/prof_fetch -name "*.db" | head -1); W=$(find /tmp/prof_write -name "*.db" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python3 $ROOT/scripts/pmc_summary.py $F $W > $OUT/${TAG}_pmc_hbm_traffic_c2.csv
# the same two traffic passes for the Cityscapes-shaped workload (extra.c4.roofline.traffic of the default bench line)
B4="python3 $ROOT/bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline --no-extra"
run4() {
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/prof_$name -o p -- $B4 > /tmp/prof_$name.log 2>&1 || echo "$name: rocprofv3 failed or timed out" >> $OUT/${TAG}_profile_errors.txt
}
run4 fetch4 FETCH_SIZE
run4 write4 WRITE_SIZE
F=$(find /tmp/prof_fetch4 -name "*.db" | head -1); W=$(find /tmp/prof_write4 -name "*.db" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python3 $ROOT/scripts/pmc_summary.py $F $W > $OUT/${TAG}_pmc_hbm_traffic_c4.csv
# per-kernel time: --kernel-trace --stats only (never next to --pmc), three steps
rm -rf /tmp/prof_stats
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /tmp/prof_stats.log 2>&1 || echo "stats: rocprofv3 failed or timed out" >> $OUT/${TAG}_profile_errors.txt
S=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/${TAG}_bench_kernel_stats.csv
# the same for the Cityscapes-shaped workload (the narrow-layer / deep-stem kernels at c4, from the final tree)
rm -rf /tmp/prof_stats4
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats4 -o s -- python3 $ROOT/bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /tmp/prof_stats4.log 2>&1 || echo "stats4: rocprofv3 failed or timed out" >> $OUT/${TAG}_profile_errors.txt
S=$(find /tmp/prof_stats4 -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/${TAG}_c4_kernel_stats.csv
ls -la $OUT/${TAG}_* 2>/dev/null
