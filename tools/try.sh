#!/bin/bash
EORB_GATHER_WIDE=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "raw or full_size or batch" > gpurun_out/try_testw.log 2>&1 || { tail -30 gpurun_out/try_testw.log; exit 1; }
tail -3 gpurun_out/try_testw.log
for cfg in "0 64" "1 64" "1 32" "1 128"; do
  set -- $cfg
  if [ $1 = 1 ]; then export EORB_GATHER_WIDE=1; else unset EORB_GATHER_WIDE; fi
  timeout -k 10 400 python bench.py --batch $2 --cpu-slices 0 --steps 10 > gpurun_out/try_w$1_b$2.json 2> gpurun_out/try_w$1_b$2.err || { tail -5 gpurun_out/try_w$1_b$2.err; exit 1; }
  python - gpurun_out/try_w$1_b$2.json $1 $2 <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("wide", sys.argv[2], "B", sys.argv[3], round(d["value"]), {k: round(v,3) for k,v in d["kernels_ms_per_step"].items() if k.startswith("ev_g")})
PY
done
