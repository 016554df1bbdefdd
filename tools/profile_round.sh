#!/bin/bash
# usage: tools/profile_round.sh <tag>  (on the GPU box)
#   -> gpurun_out/<tag>_{bench.json,bench_float.json,kernel_stats.csv,pmc_traffic.txt,roofline_inputs.json} for the default workload (w2),
#      gpurun_out/<tag>_{w1,w3,w4}_{bench.json,kernel_stats.csv}.  Copy what is to be judged into profiles/ (roofline_inputs.json under that
#      name: bench.py reads profiles/roofline_inputs.json).
tag=$1
R=$PWD
W2="--steps 3 --warmup 1 --cpu-slices 0 --no-prof"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st /tmp/f1 /tmp/f2 /tmp/q1 /tmp/q2
rocprofv3 --kernel-trace --stats -d /tmp/st -o st --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-slices 0 --no-prof > /tmp/st.log 2>&1 || { tail -5 /tmp/st.log; exit 1; }
cp $(find /tmp/st -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/f1 -o f1 --output-format csv -- python3 $R/bench.py $W2 > /tmp/f1.log 2>&1 || { tail -5 /tmp/f1.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/f2 -o f2 --output-format csv -- python3 $R/bench.py $W2 > /tmp/f2.log 2>&1 || { tail -5 /tmp/f2.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d /tmp/q1 -o q1 --output-format csv -- python3 $R/bench.py $W2 > /tmp/q1.log 2>&1 || { tail -5 /tmp/q1.log; exit 1; }
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/q2 -o q2 --output-format csv -- python3 $R/bench.py $W2 > /tmp/q2.log 2>&1 || { tail -5 /tmp/q2.log; exit 1; }
cd $R
rm -f gpurun_out/${tag}_roofline_inputs.json
python3 tools/roofline_inputs.py gpurun_out/${tag}_roofline_inputs.json w2:raw:128:1000000 /tmp/st /tmp/f1 /tmp/f2 /tmp/q1 /tmp/q2 -- python3 bench.py $W2 > gpurun_out/${tag}_pmc_summary.txt 2>&1 || { cat gpurun_out/${tag}_pmc_summary.txt; exit 1; }
cat gpurun_out/${tag}_pmc_summary.txt
mkdir -p profiles && cp gpurun_out/${tag}_roofline_inputs.json profiles/roofline_inputs.json      # (so that the bench lines below carry the measured traffic)
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
python bench.py --input float --cpu-slices 0 > gpurun_out/${tag}_bench_float.json 2>> gpurun_out/${tag}_bench.err
for w in w1 w3 w4; do
  python bench.py --workload $w > gpurun_out/${tag}_${w}_bench.json 2>> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
done
cd /tmp
for w in w1 w3 w4; do
  rm -rf /tmp/st_$w
  rocprofv3 --kernel-trace --stats -d /tmp/st_$w -o st --output-format csv -- python3 $R/bench.py --workload $w --steps 5 --warmup 2 --cpu-slices 0 --no-prof --latency-calls 0 > /tmp/st_$w.log 2>&1 || { tail -5 /tmp/st_$w.log; exit 1; }
  cp $(find /tmp/st_$w -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_${w}_kernel_stats.csv
done
cd $R
python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.load(open("gpurun_out/%s_roofline_inputs.json" % tag))
out = open("gpurun_out/%s_pmc_traffic.txt" % tag, "w")
out.write("rocprofv3 --pmc <counter> --kernel-trace, separate passes (tools/profile_round.sh); sources %s\n" % d["src_hash"])
out.write("gfx950 note (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports wide coalesced reads by 2x (doubled below); WRITE_SIZE is exact for streaming stores.\n")
for key, e in d["entries"].items():
    out.write("\n%s   (%s)\n" % (key, e["command"]))
    tot = 0.0
    for s, sc in e["scopes"].items():
        out.write("  scope %-12s traffic %9.1f MB per step, rocprof %7.3f ms\n" % (s, sc["traffic_bytes"] / 1e6, sc["rocprof_ms"]))
        tot += sc["traffic_bytes"] if s.startswith("ev_") else 0.0
        for k, kv in sc["kernels"].items():
            out.write("      %-44s %8.3f ms  fetch %9.1f MB  write %9.1f MB\n" % (k[:44], kv["avg_ms"], kv["fetch_bytes"] / 1e6, kv["write_bytes"] / 1e6))
        if "issue" in sc:
            i = sc["issue"]
            out.write("      issue (%s): VALU %.3f  LDS %.3f  waiting %.3f  issue-stalled %.3f\n" % (i["kernel"], i["valu_frac"], i["lds_frac"], i["wait_frac"], i["issue_stall_frac"]))
    out.write("  accumulation stage: %.1f MB per step\n" % (tot / 1e6))
out.close()
print(open("gpurun_out/%s_pmc_traffic.txt" % tag).read())
PY
