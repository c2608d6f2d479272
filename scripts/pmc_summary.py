"""profiles/*_pmc_hbm_traffic.csv from the two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r01_fetch -o fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r01_write -o write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python scripts/pmc_summary.py gpurun_out/r01_fetch/fetch_results.db gpurun_out/r01_write/write_results.db > profiles/r01_pmc_hbm_traffic.csv

Counter unit: KB.  read_MB_per_launch_corrected_x2 applies the gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE
tallies the 128-byte requests of wide coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte stores / atomics."""
import sqlite3
import sys


def per_kernel(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    pe = [t for t in tabs if "pmc_event" in t][0]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.kernel_name, count(*), sum(p.value) from {pe} p join {kd} d on p.event_id = d.event_id "
         f"join {ks} s on d.kernel_id = s.id group by s.kernel_name")
    return {n: (c, v) for n, c, v in db.execute(q)}


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
rows = []
for name, (n, kb) in fetch.items():
    wn, wkb = write.get(name, (n, 0.0))
    rows.append((kb + wkb, name, n, kb, wkb))
print("kernel,launches,FETCH_SIZE_KB_sum,WRITE_SIZE_KB_sum,read_MB_per_launch_corrected_x2,write_MB_per_launch")
for _, name, n, kb, wkb in sorted(rows, reverse=True)[:24]:
    short = name.replace(".kd", "")[:120]
    print(f"\"{short}\",{n},{kb:.0f},{wkb:.0f},{2 * kb / n / 1e3:.2f},{wkb / n / 1e3:.2f}")
