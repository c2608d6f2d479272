"""Per-kernel ms/step from a rocprofv3 results .db (usage: prof_summary.py DB STEPS [TOPN])."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2]); topn = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for n, c, t in rows[:topn]:
    print(f"{t/steps/1e6:8.2f} ms/step {c/steps:7.1f}/step {t/tot*100:5.1f}% {n[:100]}")
print(f"total {tot/steps/1e6:.2f} ms/step")
