#!/bin/bash
# usage: pmc.sh <tag> <bench args...> ; two SQ passes, per-kernel averages -> gpurun_out/pmc_<tag>.txt
tag=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
rm -rf /tmp/pmc1 /tmp/pmc2
rocprofv3 --pmc $P1 --kernel-trace -d /tmp/pmc1 -o p1 --output-format csv -- python3 $R/bench.py "$@" --no-prof --cpu-slices 0 > /tmp/p1.log 2>&1 || { tail -5 /tmp/p1.log; exit 1; }
rocprofv3 --pmc $P2 --kernel-trace -d /tmp/pmc2 -o p2 --output-format csv -- python3 $R/bench.py "$@" --no-prof --cpu-slices 0 > /tmp/p2.log 2>&1 || { tail -5 /tmp/p2.log; exit 1; }
cd $R
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
out = open("gpurun_out/pmc_%s.txt" % tag, "w")
for d in ("/tmp/pmc1", "/tmp/pmc2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k in acc:
            if "ev_gather" in k or "ev_scatter" in k or "ev_count" in k:
                out.write("%s dispatches=%d\n" % (k, len(n[k])))
                for c, v in sorted(acc[k].items()):
                    out.write("   %-24s %.4g per dispatch\n" % (c, v / len(n[k])))
out.close()
print(open("gpurun_out/pmc_%s.txt" % tag).read())
PY
