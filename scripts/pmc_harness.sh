#!/bin/bash
# PMC passes over the conv harness (one shape per run): usage  scripts/pmc_harness.sh <shape index> <tag>   (on the GPU box)
# each counter group is its own rocprofv3 run (block slot limits; FETCH_SIZE / WRITE_SIZE never share a pass)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
IDX=$1; TAG=$2
export CB_ONLY=$IDX CB_NOWGRAD=1 CB_STATS=1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TA_TA_BUSY TA_BUFFER_TOTAL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES" \
           "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES" \
           "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 90 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_$i -o p -- $ROOT/build/cb_pp > /tmp/pmc_$i.log 2>&1 || echo "pass $i ($grp) failed or timed out" >> $ROOT/gpurun_out/pmc_$TAG.err
done
python3 - "$TAG" <<'PY' > $ROOT/gpurun_out/pmc_$TAG.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('/tmp/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        a = acc[k][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
for f in glob.glob('/tmp/pmc_1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        d = dur[k]; d[0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; d[1] += 1
for k in acc:
    print('==', k, 'launches', dur[k][1], 'avg_us %.1f' % (dur[k][0] / max(dur[k][1], 1)))
    for c, (v, n) in sorted(acc[k].items()):
        print('   %-34s per-launch %.4g' % (c, v / n))
PY
cat $ROOT/gpurun_out/pmc_$TAG.txt
