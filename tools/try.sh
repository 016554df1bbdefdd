#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ev2im or raw or full_size or batch or mci or golden" > gpurun_out/try_test.log 2>&1 || { tail -30 gpurun_out/try_test.log; exit 1; }
tail -3 gpurun_out/try_test.log
for cfg in "raw 64" "raw 8" "raw 16"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --input $1 --batch $2 --cpu-slices 0 --steps 10 > gpurun_out/try_$1_b$2.json 2> gpurun_out/try_$1_b$2.err || { tail -5 gpurun_out/try_$1_b$2.err; exit 1; }
  python - gpurun_out/try_$1_b$2.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"]), {k: round(v,3) for k,v in d["kernels_ms_per_step"].items() if k.startswith("ev_")})
PY
done
