#!/bin/bash
# usage: tools/profile_round.sh <tag>  (on the GPU box)
#   -> gpurun_out/<tag>_{bench.json,bench_float.json,kernel_stats.csv,pmc_traffic.txt} for the default workload (w2) and
#      gpurun_out/<tag>_{w1,w3,w4}_{bench.json,kernel_stats.csv}
tag=$1
R=$PWD
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
python bench.py --input float --cpu-slices 0 > gpurun_out/${tag}_bench_float.json 2>> gpurun_out/${tag}_bench.err
for w in w1 w3 w4; do
  python bench.py --workload $w > gpurun_out/${tag}_${w}_bench.json 2>> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st /tmp/f1 /tmp/f2
rocprofv3 --kernel-trace --stats -d /tmp/st -o st --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-slices 0 --no-prof > /tmp/st.log 2>&1 || { tail -5 /tmp/st.log; exit 1; }
cp $(find /tmp/st -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_kernel_stats.csv
for w in w1 w3 w4; do
  rm -rf /tmp/st_$w
  rocprofv3 --kernel-trace --stats -d /tmp/st_$w -o st --output-format csv -- python3 $R/bench.py --workload $w --steps 5 --warmup 2 --cpu-slices 0 --no-prof --latency-calls 0 > /tmp/st_$w.log 2>&1 || { tail -5 /tmp/st_$w.log; exit 1; }
  cp $(find /tmp/st_$w -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_${w}_kernel_stats.csv
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/f1 -o f1 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-slices 0 --no-prof > /tmp/f1.log 2>&1 || { tail -5 /tmp/f1.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/f2 -o f2 --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-slices 0 --no-prof > /tmp/f2.log 2>&1 || { tail -5 /tmp/f2.log; exit 1; }
cd $R
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
out = open("gpurun_out/%s_pmc_traffic.txt" % tag, "w")
out.write("rocprofv3 --pmc <counter> --kernel-trace, separate passes; command: python3 bench.py --steps 3 --warmup 1 --cpu-slices 0 --no-prof\n")
out.write("workload per dispatch: 128 slices x 1,000,000 raw sensor events, 240x180; counter unit KB (x1024 = bytes)\n")
out.write("gfx950 note (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports wide coalesced reads by 2x; WRITE_SIZE is exact for streaming stores.\n")
for d, name in (("/tmp/f1", "FETCH_SIZE"), ("/tmp/f2", "WRITE_SIZE")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = r["Kernel_Name"][:70]
            acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        out.write("\n%s per dispatch (average):\n" % name)
        for k in sorted(acc, key=lambda k: -acc[k])[:14]:
            v = acc[k] / len(n[k])
            out.write("  %-72s dispatches=%-4d %12.1f KB = %9.2f MB\n" % (k, len(n[k]), v, v * 1024 / 1e6))
out.close()
print(open("gpurun_out/%s_pmc_traffic.txt" % tag).read())
PY
