#!/bin/bash
# round 4, run 5: rocprof kernel stats of the step on the new weight-gradient plan; the teacher two-pass question measured on the harness
# (conv_ws statistics pass without stores + an addend-shaped second pass, against statistics + bn_apply today); the distributed tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r04_ws_two_pass.txt
echo "== conv_ws_kernel on the layer-3 / layer-2 / layer-1 / layer-4 conv3 shapes: shipped build (plain / stats / add), then -DWS_ABL_NOSTORE (the same launches with every output store dropped) ==" > $O
for rep in 1 2; do
  for b in ws_bench ws_bench_nostore; do
    for sh in 0 1 2 8; do
      echo "-- $b shape $sh" >> $O; WB_ONLY=$sh timeout 120 ./build/$b 2>&1 | grep -E "plain|stats|add" | cut -c1-150 >> $O
    done
  done
done
tail -30 $O
timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -m gpu > gpurun_out/r04_run5_tests.txt 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r04_run5_tests.txt
tail -4 gpurun_out/r04_run5_tests.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extra > /tmp/prof_stats.log 2>&1
S=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $GRAFT_REPO_ROOT/gpurun_out/r04_run5_kernel_stats.csv
T=$(find /tmp/prof_stats -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && python3 - "$T" > $GRAFT_REPO_ROOT/gpurun_out/r04_run5_wgrad_trace.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    n = r.get("Kernel_Name", "")
    if "wgrad" in n:
        g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0) or 0)) // max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1) or 1)), 1)
        d[(n[:40], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:42s} wgs {k[1]:5d}  n {len(v):4d}  avg {sum(v)/len(v):8.1f} us  total {sum(v)/1e3:7.2f} ms")
PY
head -20 $GRAFT_REPO_ROOT/gpurun_out/r04_run5_wgrad_trace.txt
