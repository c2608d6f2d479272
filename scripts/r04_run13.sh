#!/bin/bash
# round 4: per-kernel stats of the 769^2 workload, and of the forced-collectives step on both SyncBN paths (what is left of the +3.4 / +4.1 ms)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {  # name, env..., args
  local name=$1; shift
  rm -rf /tmp/ps_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$name -o s -- python3 $ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extra "$@" > /tmp/ps_$name.log 2>&1
  S=$(find /tmp/ps_$name -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $ROOT/gpurun_out/r04_${name}_kernel_stats.csv
}
run c4 --workload c4
export CSS_FORCE_COLLECTIVES=1
run force_rccl
export CSS_SYNCBN=peer
run force_peer
ls -la $ROOT/gpurun_out/r04_*_kernel_stats.csv
python3 - <<PY
import csv
for n in ("force_rccl", "force_peer"):
    rows = list(csv.DictReader(open("$ROOT/gpurun_out/r04_%s_kernel_stats.csv" % n)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / 5e6
    print(n, "kernel ms/step", round(tot, 2))
    for r in rows:
        nm = r["Name"]
        if any(k in nm for k in ("peer", "nccl", "rccl", "bn_finalize", "bn_reduce", "AllReduce", "all_reduce", "Broadcast")):
            print("   %6.1f/step  avg %7.1f us  %7.3f ms/step  %s" % (int(r["Calls"]) / 5, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 5e6, nm[:90]))
PY
