#!/bin/bash
# PMC passes over the conv_ws harness (on the GPU box): usage scripts/ws_pmc.sh <shape index> <tag>
# each counter group is its own rocprofv3 run (FETCH_SIZE / WRITE_SIZE never share a pass); FETCH_SIZE / WRITE_SIZE are in KB
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
IDX=$1; TAG=$2
export WB_ONLY=$IDX WB_REPS=5
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/wpmc_$i
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/wpmc_$i -o p -- $ROOT/build/ws_bench > /tmp/wpmc_$i.log 2>&1 || echo "pass $i ($grp) failed or timed out" >> $ROOT/gpurun_out/wpmc_$TAG.err
done
python3 - "$TAG" <<'PY' > $ROOT/gpurun_out/wpmc_$TAG.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('/tmp/wpmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:70]
        a = acc[k][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
for f in glob.glob('/tmp/wpmc_1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:70]
        d = dur[k]; d[0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; d[1] += 1
for k in acc:
    print('==', k, 'launches', dur[k][1], 'avg_us %.1f' % (dur[k][0] / max(dur[k][1], 1)))
    for c, (v, n) in sorted(acc[k].items()):
        print('   %-34s per-launch %.5g' % (c, v / n))
PY
cat $ROOT/gpurun_out/wpmc_$TAG.txt
