#!/bin/bash
# usage: tools/pmc_mem.sh <tag> <bench args...> ; memory-path counters of the accumulation kernels -> gpurun_out/pmc_mem_<tag>.txt
tag=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
# (TA_* counters made rocprofv3 hang on this pool: not collected; every pass runs under its own timeout)
PASSES=("TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
        "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
        "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD")
i=0
rm -f /tmp/pmcm_*.log $R/gpurun_out/pmc_mem_${tag}.progress
for P in "${PASSES[@]}"; do
  echo "pass $i: $P" >> $R/gpurun_out/pmc_mem_${tag}.progress
  rm -rf /tmp/pmcm_$i
  timeout -k 10 150 rocprofv3 --pmc $P --kernel-trace -d /tmp/pmcm_$i -o p --output-format csv -- python3 $R/bench.py "$@" --no-prof --cpu-slices 0 > /tmp/pmcm_$i.log 2>&1 || { echo "pass $i failed" >> $R/gpurun_out/pmc_mem_${tag}.progress; tail -3 /tmp/pmcm_$i.log >> $R/gpurun_out/pmc_mem_${tag}.progress; }
  i=$((i+1))
done
cd $R
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
out = open("gpurun_out/pmc_mem_%s.txt" % tag, "w")
for d in sorted(glob.glob("/tmp/pmcm_[0-9]*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:48]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k in acc:
            if "ev_gather" in k or "ev_scatter" in k or "ev_count" in k:
                out.write("%s dispatches=%d\n" % (k, len(n[k])))
                for c, v in sorted(acc[k].items()):
                    out.write("   %-36s %.5g per dispatch\n" % (c, v / len(n[k])))
out.close()
print(open("gpurun_out/pmc_mem_%s.txt" % tag).read())
PY
