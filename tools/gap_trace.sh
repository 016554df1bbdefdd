#!/bin/bash
# usage: tools/gap_trace.sh <bench args...>: kernel trace of a bench run; prints the longest kernels and the longest idle gaps between kernels
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gt
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/gt -o gt --output-format csv -- python3 $R/bench.py "$@" > /tmp/gt.log 2>&1 || { tail -5 /tmp/gt.log; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/gt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
print("kernels:", len(rows))
longest = sorted(rows, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)[:6]
for r in longest:
    print("long  %-50s at %9.2f ms  dur %8.3f ms" % (r["Kernel_Name"][:50], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
gaps = []
end = int(rows[0]["End_Timestamp"])
for a, b in zip(rows, rows[1:]):
    end = max(end, int(a["End_Timestamp"]))
    gaps.append((int(b["Start_Timestamp"]) - end, a, b))
for g, a, b in sorted(gaps, key=lambda x: -x[0])[:8]:
    print("gap %8.3f ms at %9.2f ms  after %-40s before %-40s" % (g / 1e6, (int(a["End_Timestamp"]) - t0) / 1e6, a["Kernel_Name"][:40], b["Kernel_Name"][:40]))
# one step in the middle of the run, kernel by kernel: start offset, duration, idle time before it
starts = [i for i, r in enumerate(rows) if "ev_minmax_init" in r["Kernel_Name"]]
if len(starts) > 6:
    i0, i1 = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
    base = int(rows[i0]["Start_Timestamp"]); end = base
    for r in rows[i0:i1 + 1]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.1f us  dur %8.1f  idle before %7.1f  %s" % ((st - base) / 1e3, (en - st) / 1e3, max(0, st - end) / 1e3, r["Kernel_Name"][:60]))
        end = max(end, en)
PY
