"""Per-kernel SQ counter summary from a rocprofv3 PMC pass (usage: pmc_sq_summary.py RESULTS.db [TOPN] > profiles/rNN_pmc_sq_*.csv).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
        --kernel-trace -d gpurun_out/r01_sq -o sq -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline

Every counter is summed over the launches of a kernel (and over the chip, as rocprofv3 reports it).  Derived columns:
  mfma_util_at_2p4GHz = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz): the counter ticks 32 cycles per
      v_mfma_f32_32x32x16_bf16 (MI355X_MICROARCH.md, cycle-constants table), i.e. 1024 busy-cycles per 2^25 FLOP, so this is
      MFMA FLOP/s over the 2.5 PFLOP/s dense bf16 peak, counted by the hardware (padding and tail MFMAs included);
  mfma_util_gui = the same against GRBM_GUI_ACTIVE x 1024 / n_xcc-sum (actual clocks; the normalisation is printed);
  wait_any / wait_inst / active = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint shares of
      a resident wave's time: parked on s_waitcnt/barrier, issue-stalled, issuing).
Durations are those of the PMC run itself (kernels are serialised and a little slower under counter collection).
"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
topn = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda key: [x for x in tabs if key in x][0]
pe, pi, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("kernel_dispatch"), t("kernel_symbol")
cols = [r[1] for r in db.execute(f"pragma table_info({pi})")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
pmc_name = {i: n for i, n in db.execute(f"select id, {name_col} from {pi}")}
dur = {}
for name, n, d in db.execute(f"select s.kernel_name, count(*), sum(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name"):
    dur[name] = (n, d)
vals = defaultdict(lambda: defaultdict(float))
q = (f"select s.kernel_name, p.pmc_id, sum(p.value) from {pe} p join {kd} d on p.event_id = d.event_id "
     f"join {ks} s on d.kernel_id = s.id group by s.kernel_name, p.pmc_id")
for name, pid, v in db.execute(q):
    vals[name][pmc_name.get(pid, str(pid))] += v
counters = sorted({c for v in vals.values() for c in v})
tot_gui = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in vals.values())
tot_ns = sum(dur[k][1] for k in vals if k in dur)
gui_per_ns = tot_gui / tot_ns if tot_ns else 0.0
print(f"# GRBM_GUI_ACTIVE per ns of kernel time over the whole run: {gui_per_ns:.3f} (= clock in GHz x number of XCC instances summed)")
hdr = ["kernel", "launches", "total_ms", "avg_us"] + counters + ["mfma_util_at_2p4GHz", "mfma_util_gui", "wait_any", "wait_inst", "active"]
print(",".join(hdr))
rows = sorted(vals.items(), key=lambda kv: -dur.get(kv[0], (0, 0))[1])[:topn]
for name, v in rows:
    n, d = dur[name]
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    wc = v.get("SQ_WAVE_CYCLES", 0.0)
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    nx = round(gui_per_ns / 2.4) if gui_per_ns > 3 else 1          # summed over XCCs or not
    u24 = busy / (1024 * d * 2.4) if d else 0.0
    ug = busy / (1024 * gui / max(nx, 1)) if gui else 0.0
    sh = lambda c: (v.get(c, 0.0) / wc) if wc else 0.0
    short = name.replace(".kd", "").replace('"', "'")[:90]
    print(",".join([f'"{short}"', str(n), f"{d / 1e6:.3f}", f"{d / n / 1e3:.1f}"] + [f"{v.get(c, 0.0):.0f}" for c in counters] +
                   [f"{u24:.4f}", f"{ug:.4f}", f"{sh('SQ_WAIT_ANY'):.3f}", f"{sh('SQ_WAIT_INST_ANY'):.3f}", f"{sh('SQ_ACTIVE_INST_ANY'):.3f}"]))
