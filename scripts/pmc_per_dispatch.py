"""Per-dispatch counter values from a rocprofv3 PMC pass (usage: pmc_per_dispatch.py RESULTS.db [NAME_SUBSTRING]): one line per kernel
launch in launch order - kernel, workgroups, duration (us) and every collected counter (FETCH_SIZE / WRITE_SIZE in KB)."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda key: [x for x in tabs if key in x][0]
pe, pi, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("kernel_dispatch"), t("kernel_symbol")
pmc_name = {i: n for i, n in db.execute(f"select id, name from {pi}")}
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
gx = "grid_size_x" if "grid_size_x" in cols else [c for c in cols if "grid" in c and "x" in c][0]
wx = "workgroup_size_x" if "workgroup_size_x" in cols else [c for c in cols if "workgroup" in c and "x" in c][0]
vals = defaultdict(dict)
for ev, pid, v in db.execute(f"select event_id, pmc_id, value from {pe}"):
    vals[ev][pmc_name.get(pid, str(pid))] = vals[ev].get(pmc_name.get(pid, str(pid)), 0.0) + v
for ev, name, start, end, g, w in db.execute(f"select d.event_id, s.kernel_name, d.start, d.end, d.{gx}, d.{wx} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"):
    if sub in name:
        c = " ".join(f"{k}={v:.0f}" for k, v in sorted(vals.get(ev, {}).items()))
        print(f"{name[:48]:48s} wgs={g // max(w, 1):6d} {(end - start) / 1e3:9.1f} us  {c}")
