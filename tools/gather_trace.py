"""Per-wave timeline of the heaviest tile's ev_gather workgroup (64 batches).  Needs a library built with
-DEORB_TRACE -DEORB_DIAG (EORB_FE_LIB=<that .so> python tools/gather_trace.py [--raw]); the product build has no trace code."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from eorb_slam_amd import frontend, synth, _lib
L = _lib.lib()
W, H, B, N = 240, 180, 8, 1000000
raw = "--raw" in sys.argv
pairs = [synth.shapes_events(N, W, H, seed=2 + b, motion=0.5, undistort=True, return_raw=True) for b in range(B)]
fb = frontend.FrontEndBatch(W, H, 1.0, False, max_batch=B, max_events=N)
c = fb.ctx
if raw:
    mx, my = synth.undistort_lut(W, H); frontend.EvImConverter.set_undistort_maps(mx, my, True, ctx=c)
    blob = np.concatenate([p[1] for p in pairs])
else:
    blob = np.concatenate([frontend.pack_events(p[0]) for p in pairs])
d = c.dev_alloc(blob.nbytes); c.upload(d, blob)
offs = np.arange(B + 1, dtype=np.int64) * N
for it in range(2):
    fb.run_dev(d, offs, raw=raw); c.sync()
out = (C.c_ulonglong * (16 * 64 * 8))()
L.eorb_trace_read.argtypes = [C.c_void_p, C.c_int]
print("trace:", L.eorb_trace_read(out, 16 * 64 * 8))
T = np.array(list(out), np.int64).reshape(16, 64, 8)
t0 = T[:8, :, 0]
names = ["adds", "setup", "val0", "val1", "val2", "val3", "val4", "val5"]
print("interval (top-of-loop to next top) per wave, median:", [int(np.median(np.diff(t0[w]))) for w in range(8)])
for w in range(8):
    rel = T[w, :, :] - T[w, :, 0:1]
    rel = np.where(T[w] > 0, rel, -1)
    med = [int(np.median(rel[:, k][rel[:, k] >= 0])) if (rel[:, k] >= 0).any() else -1 for k in range(8)]
    print(names[w], "stamps rel. to loop top (median):", med)
# a few raw batches for wave skew
base = T[:8, 10:14, 0].min(axis=0)
for b in range(10, 14):
    print("batch", b, {names[w]: (int(T[w, b, 0] - T[:8, b, 0].min()), int(T[w, b, 6] - T[:8, b, 0].min())) for w in range(8)})
