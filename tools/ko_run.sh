#!/bin/bash
# usage: tools/ko_run.sh <lib> <tag> [ENV=VAL ...]: kernel stats of the default bench with an experiment build of the library
lib=$1; tag=$2; shift 2
R=$PWD
for kv in "$@"; do export "$kv"; done
export EORB_FE_LIB=$R/$lib
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ko_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ko_$tag -o st --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-slices 0 --no-prof > /tmp/ko_$tag.log 2>&1 || { tail -5 /tmp/ko_$tag.log; exit 1; }
grep -o '"value": [0-9.]*' /tmp/ko_$tag.log | tail -1
python3 - /tmp/ko_$tag $R/gpurun_out/ko_$tag.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
out = open(sys.argv[2], "w")
for r in list(csv.DictReader(open(f)))[:16]:
    line = "%-60s %4s %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3)
    print(line); out.write(line + "\n")
PY
