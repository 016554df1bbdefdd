#!/bin/bash
# usage (GPU box): tools/l1_probe.sh  -> gpurun_out/l1_probe.txt: wall times of the one-call L1 seams + per-kernel durations by launch shape
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/l1p
rocprofv3 --kernel-trace -d /tmp/l1p -o l1 -- python3 $R/tools/l1_probe.py > $R/gpurun_out/l1_probe.txt 2> /tmp/l1p.err || { tail -5 /tmp/l1p.err; exit 1; }
python3 - >> $R/gpurun_out/l1_probe.txt <<'PY'
import sqlite3, glob, numpy as np
db = sqlite3.connect(glob.glob('/tmp/l1p/**/*.db', recursive=True)[0]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(cur.execute(f"select s.kernel_name, d.grid_size_x, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id"))
agg = {}
for name, g, t in rows:
    agg.setdefault((name.split('(')[0][:70], g), []).append(t / 1e3)
for (name, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-72s grid %8d  n %5d  avg %8.1f us  total %9.1f us" % (name, g, len(v), np.mean(v), np.sum(v)))
PY
cat $R/gpurun_out/l1_probe.txt | head -60
