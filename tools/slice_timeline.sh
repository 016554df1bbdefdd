#!/bin/bash
# usage (GPU box): tools/slice_timeline.sh -> gpurun_out/slice_timeline.txt: the kernels and copies of ONE eorb_ev_slice_extract call (tools/latency.cpp) on a time line:
# start offset, duration and the idle time in front of each (kernel + memory-copy trace of rocprofv3)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/stl
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/stl -o stl --output-format csv -- $R/build/latency > /tmp/stl.log 2>&1 || { tail -5 /tmp/stl.log; exit 1; }
python3 - > $R/gpurun_out/slice_timeline.txt <<'PY'
import csv, glob
ev = []
for f in glob.glob("/tmp/stl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
for f in glob.glob("/tmp/stl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", ""))[:40]))
ev.sort()
# calls = runs of kernels / copies with less than 8 us of idle time between them (a call's inputs arrive through a kernel or are read in
# place: no host-to-device copy marks its start any more; the host spends more than that between two calls);
# print a group that contains ev_pre_kernel<false> and octree but no klt (slice_extract, raw events)
groups, cur, last_end = [], [], None
for e in ev:
    if cur and e[0] - last_end > 8000: groups.append(cur); cur = []
    cur.append(e); last_end = e[1] if last_end is None or not cur[:-1] else max(last_end, e[1])
groups.append(cur)
sel = [g for g in groups if any("ev_pre_kernel" in x[2] for x in g) and any("octree" in x[2] for x in g) and not any("klt" in x[2] for x in g)]
trk = [g for g in groups if any("klt_track" in x[2] for x in g) and any("ev_pre_kernel" in x[2] for x in g)]
for name, gs in (("eorb_ev_slice_extract", sel), ("eorb_ev_slice_track", trk), ("W3 extract", [g for g in groups if any("describe_kernel<true>" in x[2] or "brief" in x[2] for x in g) and sum("pyr_resize" in x[2] for x in g) == 3]),
                 ("SearchForInitialization", [g for g in groups if any("win_cand_kernel<0>" in x[2] for x in g)]),
                 ("SearchByProjection", [g for g in groups if any("win_cand_kernel<1>" in x[2] for x in g)])):
    if not gs: continue
    g = gs[len(gs) // 2]
    base = g[0][0]; end = base
    print("---- %s: one call of %d in the trace ----" % (name, len(gs)))
    for st, en, nm in g:
        print("%8.1f us  dur %7.1f  idle before %6.1f  %s" % ((st - base) / 1e3, (en - st) / 1e3, max(0, st - end) / 1e3, nm))
        end = max(end, en)
    print("   span %.1f us, busy %.1f us" % ((end - base) / 1e3, sum(en - st for st, en, _ in g) / 1e3))
PY
cat $R/gpurun_out/slice_timeline.txt
