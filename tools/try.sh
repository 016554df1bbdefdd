#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ev2im or raw or full_size or batch or mci or golden" > gpurun_out/try_test.log 2>&1 || { tail -30 gpurun_out/try_test.log; exit 1; }
tail -3 gpurun_out/try_test.log
EORB_GATHER_NC=4 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "raw or full_size or batch" > gpurun_out/try_test4.log 2>&1 || { tail -30 gpurun_out/try_test4.log; exit 1; }
tail -3 gpurun_out/try_test4.log
for cfg in "2 64" "4 64" "2 8" "4 8" "2 16" "4 16" "2 32" "4 32"; do
  set -- $cfg
  EORB_GATHER_NC=$1 timeout -k 10 300 python bench.py --batch $2 --cpu-slices 0 --steps 10 > gpurun_out/try_nc$1_b$2.json 2> gpurun_out/try_nc$1_b$2.err || { tail -5 gpurun_out/try_nc$1_b$2.err; exit 1; }
  python - gpurun_out/try_nc$1_b$2.json $1 $2 <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print("NC", sys.argv[2], "B", sys.argv[3], round(d["value"]), {k: round(v,3) for k,v in d["kernels_ms_per_step"].items() if k.startswith("ev_g")})
PY
done
