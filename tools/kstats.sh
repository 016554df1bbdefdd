#!/bin/bash
# usage: tools/kstats.sh <tag> <bench args...>  -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of that bench run)
tag=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats -d /tmp/ks_$tag -o st --output-format csv -- python3 $R/bench.py "$@" --cpu-slices 0 --no-prof > /tmp/ks_$tag.log 2>&1 || { tail -5 /tmp/ks_$tag.log; exit 1; }
cp $(find /tmp/ks_$tag -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_kernel_stats.csv
head -16 $R/gpurun_out/${tag}_kernel_stats.csv | cut -c1-170
