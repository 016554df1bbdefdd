"""usage: python tools/kernel_table.py <rocprofv3 results .db> [name filter]: per (kernel, grid size) launches / average / total time"""
import sqlite3, sys, numpy as np
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
agg = {}
for name, g, t in cur.execute(f"select s.kernel_name, d.grid_size_x, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id"):
    if flt in name:
        agg.setdefault((name.split('(')[0][:64], g), []).append(t / 1e3)
tot = sum(sum(v) for v in agg.values())
for (name, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-66s grid %8d  n %5d  avg %8.1f us  total %9.1f us" % (name, g, len(v), np.mean(v), np.sum(v)))
print("all kernels: %.1f us" % tot)
