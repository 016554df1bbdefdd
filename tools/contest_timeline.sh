R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ct
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/ct -o ct --output-format csv -- $R/build/chain_latency > /tmp/ct.log 2>&1
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob("/tmp/ct/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:56]))
for f in glob.glob("/tmp/ct/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[:30]))
ev.sort()
groups, cur, last_end = [], [], None
for e in ev:
    if cur and e[0] - last_end > 25000: groups.append(cur); cur = []
    cur.append(e); last_end = e[1] if not cur[:-1] else max(last_end, e[1])
groups.append(cur)
gs = [g for g in groups if any("ev_focus_patch" in x[2] for x in g)]
g = gs[len(gs) // 2]
base = g[0][0]; end = base
for st, en, nm in g:
    print("%8.1f us  dur %7.1f  idle before %6.1f  %s" % ((st - base) / 1e3, (en - st) / 1e3, max(0, st - end) / 1e3, nm)); end = max(end, en)
print("span %.1f us busy %.1f us, %d ops" % ((end - base) / 1e3, sum(en - st for st, en, _ in g) / 1e3, len(g)))
PY
