#!/bin/bash
# usage: tools/pmc_kernel.sh <tag> <kernel name substring> <bench args...>: issue-pipe utilisation of ONE kernel of a bench run
# (two rocprofv3 --pmc passes, the formulas of tools/roofline_inputs.py) -> gpurun_out/<tag>_pmc.txt
tag=$1; kern=$2; shift 2
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk1 /tmp/pk2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d /tmp/pk1 -o p --output-format csv -- python3 $R/bench.py "$@" > /tmp/pk1.log 2>&1 || { tail -3 /tmp/pk1.log; exit 1; }
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace -d /tmp/pk2 -o p --output-format csv -- python3 $R/bench.py "$@" > /tmp/pk2.log 2>&1 || { tail -3 /tmp/pk2.log; exit 1; }
cd $R
python3 - "$tag" "$kern" "$*" <<'PY'
import sys
sys.path.insert(0, "tools")
import roofline_inputs as ri
tag, kern, cmd = sys.argv[1:4]
a, b = ri.counters("/tmp/pk1"), ri.counters("/tmp/pk2")
out = open("gpurun_out/%s_pmc.txt" % tag, "w")
def w(s):
    print(s); out.write(s + "\n")
w("rocprofv3 --pmc (two passes) -- python3 bench.py %s" % cmd)
for k in a:
    if kern not in k:
        continue
    x, y = a[k], b.get(k, {})
    cyc = y.get("GRBM_GUI_ACTIVE", 0.0) / ri.N_XCD
    w(k.split("(")[0])
    if cyc > 0:
        w("  cycles %.0f; VALU %.3f  SALU %.3f  LDS %.3f  waiting %.3f  issue-stalled %.3f" % (
            cyc, x.get("SQ_INSTS_VALU", 0) * 4 / (cyc * ri.N_SIMD), x.get("SQ_INSTS_SALU", 0) / (cyc * ri.N_CU), y.get("SQ_LDS_IDX_ACTIVE", 0) / (cyc * ri.N_CU),
            x.get("SQ_WAIT_ANY", 0) / max(x.get("SQ_WAVE_CYCLES", 1), 1), x.get("SQ_WAIT_INST_ANY", 0) / max(x.get("SQ_WAVE_CYCLES", 1), 1)))
    w("  per launch: waves %.0f, VALU instructions %.0f, SALU %.0f, LDS %.0f, vector loads %.0f, scalar loads %.0f" % (
        y.get("SQ_WAVES", 0), x.get("SQ_INSTS_VALU", 0), x.get("SQ_INSTS_SALU", 0), x.get("SQ_INSTS_LDS", 0), y.get("SQ_INSTS_VMEM_RD", 0), y.get("SQ_INSTS_SMEM", 0)))
PY
