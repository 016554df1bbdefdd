#!/bin/bash
# usage: tools/profile_round.sh <tag> [workloads...]  (on the GPU box; default workloads: w2 w1 w3 w4 w1full)
#   per workload: rocprofv3 --kernel-trace --stats, then separate --pmc passes (FETCH_SIZE, WRITE_SIZE, two SQ passes) of the same bench
#   command -> gpurun_out/<tag>_<w>_kernel_stats.csv and one entry of gpurun_out/<tag>_roofline_inputs.json (also copied to
#   profiles/roofline_inputs.json: bench.py reads `roofline.traffic` / `roofline.issue` from there while the sources' hash matches),
#   then the bench lines themselves: gpurun_out/<tag>_bench.json (default command) and gpurun_out/<tag>_<w>_bench.json.
#   Progress goes to gpurun_out/<tag>_progress.txt.  Copy what is to be judged into profiles/.
tag=$1; shift
WL=${@:-w2 w1 w3 w4 w1full}
R=$PWD
cd /tmp && export TMPDIR=/tmp
PROG=$R/gpurun_out/${tag}_progress.txt; : > $PROG
rm -f $R/gpurun_out/${tag}_roofline_inputs.json
for w in $WL; do
  case $w in
    w2) key="w2:raw:128:1000000"; args="--no-side";;
    w1) key="w1:raw:1024:2000"; args="--workload w1 --latency-calls 0";;
    w3) key="w3:frames"; args="--workload w3";;
    w4) key="w4:frames"; args="--workload w4";;
    w1full) key="w1full:chain"; args="--workload w1full";;
  esac
  CMD="$args --steps 3 --warmup 1 --cpu-slices 0 --no-prof"
  for pass in st f1 f2 q1 q2; do
    rm -rf /tmp/${pass}_$w
    case $pass in
      st) opt="--kernel-trace --stats";;
      f1) opt="--pmc FETCH_SIZE --kernel-trace";;
      f2) opt="--pmc WRITE_SIZE --kernel-trace";;
      q1) opt="--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace";;
      q2) opt="--pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace";;
    esac
    echo "$(date +%T) $w $pass" >> $PROG
    timeout -k 10 400 rocprofv3 $opt -d /tmp/${pass}_$w -o $pass --output-format csv -- python3 $R/bench.py $CMD > /tmp/${pass}_$w.log 2>&1 || { echo "$w $pass failed" >> $PROG; tail -5 /tmp/${pass}_$w.log >> $PROG; exit 1; }
  done
  cp $(find /tmp/st_$w -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_${w}_kernel_stats.csv
  (cd $R && python3 tools/roofline_inputs.py gpurun_out/${tag}_roofline_inputs.json $key /tmp/st_$w /tmp/f1_$w /tmp/f2_$w /tmp/q1_$w /tmp/q2_$w -- python3 bench.py $CMD >> gpurun_out/${tag}_pmc_summary.txt 2>&1) || { echo "roofline_inputs $w failed" >> $PROG; exit 1; }
done
cd $R
mkdir -p profiles && cp gpurun_out/${tag}_roofline_inputs.json profiles/roofline_inputs.json      # (so that the bench lines below carry the measured traffic)
echo "$(date +%T) bench lines" >> $PROG
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err >> $PROG; exit 1; }
for w in $WL; do
  [ $w = w2 ] && continue
  python bench.py --workload $w > gpurun_out/${tag}_${w}_bench.json 2>> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err >> $PROG; exit 1; }
done
python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.load(open("gpurun_out/%s_roofline_inputs.json" % tag))
out = open("gpurun_out/%s_pmc_traffic.txt" % tag, "w")
out.write("rocprofv3 --pmc <counter> --kernel-trace, separate passes (tools/profile_round.sh); sources %s\n" % d["src_hash"])
out.write("gfx950 note (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports wide coalesced reads by 2x: doubled for the streaming kernels only (factor per kernel below); WRITE_SIZE is exact for streaming stores.\n")
for key, e in d["entries"].items():
    out.write("\n%s   (%s)\n" % (key, e["command"]))
    tot = 0.0
    for s, sc in e["scopes"].items():
        out.write("  scope %-16s traffic %9.1f MB per step, rocprof %7.3f ms\n" % (s, sc["traffic_bytes"] / 1e6, sc["rocprof_ms"]))
        tot += sc["traffic_bytes"] if s.startswith("ev_") else 0.0
        for k, kv in sc["kernels"].items():
            out.write("      %-44s %8.3f ms x %5d  fetch %9.1f MB (x%d)  write %9.1f MB\n" % (k[:44], kv["avg_ms"], kv["calls"], kv["fetch_bytes"] / 1e6, kv.get("fetch_factor", 2), kv["write_bytes"] / 1e6))
        if "issue" in sc:
            i = sc["issue"]
            out.write("      issue (%s): VALU %.3f  LDS %.3f  SALU %.3f  waiting %.3f  issue-stalled %.3f\n" % (i["kernel"], i["valu_frac"], i["lds_frac"], i.get("salu_frac", 0), i["wait_frac"], i["issue_stall_frac"]))
    out.write("  accumulation stage: %.1f MB per step\n" % (tot / 1e6))
out.close()
print(open("gpurun_out/%s_pmc_traffic.txt" % tag).read())
PY
echo "$(date +%T) done" >> $PROG
