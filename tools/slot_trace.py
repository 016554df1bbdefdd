"""Schedule of the slot gather (ev_slots.hip): per-wave start / end / entries of one batch.
Needs the trace build: make -C eorb_slam_amd/csrc ../../build/libeorb_fe_trace.so; EORB_FE_LIB=build/libeorb_fe_trace.so python tools/slot_trace.py"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eorb_slam_amd import frontend as fe, synth, _lib
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=128); ap.add_argument("--events", type=int, default=1000000)
a = ap.parse_args()
L = _lib.lib()
W, H, B, N = 240, 180, a.batch, a.events
mx, my = synth.undistort_lut(W, H)
base = [synth.shapes_events(N, W, H, seed=2 + b, motion=0.5, undistort=True, return_raw=True)[1] for b in range(B)]
blob = np.concatenate(base)
fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=B, max_events=N)
c, cap = fb.ctx, fb.cap
fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=c)
d_ev = c.dev_alloc(blob.nbytes); c.upload(d_ev, blob)
d_img = c.dev_alloc(B * W * H); d_kp = c.dev_alloc(B * cap * 28); d_desc = c.dev_alloc(B * cap * 32)
d_n = c.dev_alloc(B * 4); d_m = c.dev_alloc(B * cap * 4); d_nm = c.dev_alloc(B * 4)
off = np.arange(B + 1, dtype=np.int64) * N
for it in range(3):
    fb.run_dev(d_ev, off, d_img, d_kp, d_desc, d_n, d_m, d_nm, raw=True)
c.sync()
MAXR = 1 << 17
out = (C.c_ulonglong * (6 * MAXR))()
L.eorb_slot_trace_read.restype = C.c_int; L.eorb_slot_trace_read.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
n = L.eorb_slot_trace_read(c.h, out, MAXR)
T = np.frombuffer(out, dtype=np.uint64)[:6 * n].reshape(n, 6).astype(np.int64)
T = T[T[:, 2] > 0]
t0 = T[:, 1].min(); st = (T[:, 1] - t0) / 100.0; en = (T[:, 2] - t0) / 100.0        # microseconds (100 MHz wall clock)
print("waves with records:", len(T), " kernel span: %.1f us" % en.max(), " listed entries: %.1f M, processed %.1f M" % (T[:, 3].sum() / 1e6, T[:, 5].sum() / 1e6))
dur = en - st
print("wave duration us: median %.1f  p90 %.1f  max %.1f" % (np.median(dur), np.percentile(dur, 90), dur.max()))
busy = T[:, 3] > 0
rate = T[busy, 5] / np.maximum(dur[busy], 1e-3)           # entries per us per wave
print("entries/us per busy wave: median %.1f  p10 %.1f  p90 %.1f   (ns per entry: median %.1f)" % (np.median(rate), np.percentile(rate, 10), np.percentile(rate, 90), 1e3 / np.median(rate)))
# occupancy over time: waves alive per 50 us bin
bins = np.arange(0, en.max() + 50, 50.0)
alive = [(int(((st < b1) & (en > b0)).sum())) for b0, b1 in zip(bins[:-1], bins[1:])]
work = np.zeros(len(bins) - 1)
for i in range(len(T)):
    if T[i, 3] == 0: continue
    b0 = int(st[i] // 50); b1 = int(min(en[i] // 50, len(work) - 1))
    for b in range(b0, b1 + 1):
        lo = max(st[i], bins[b]); hi = min(en[i], bins[b + 1])
        if hi > lo: work[b] += T[i, 5] * (hi - lo) / max(dur[i], 1e-3)
print("time bin (50 us): waves alive / M entries processed")
for b in range(len(work)):
    print("  %5.0f us  alive %5d  %.1f M entries (%.0f G/s)" % (bins[b], alive[b], work[b] / 1e6, work[b] / 50.0 * 1e-3))
# the longest items
idx = np.argsort(-T[:, 5])[:8]
for i in idx:
    print("  heavy wave: tile %d listed %d processed %d items %d start %.0f end %.0f us -> %.1f ns/processed entry" % (T[i, 0], T[i, 3], T[i, 5], T[i, 4], st[i], en[i], dur[i] * 1e3 / max(T[i, 5], 1)))
last = np.argsort(-en)[:8]
for i in last:
    print("  last wave: tile %d listed %d processed %d items %d start %.0f end %.0f us -> %.1f ns/processed entry" % (T[i, 0], T[i, 3], T[i, 5], T[i, 4], st[i], en[i], dur[i] * 1e3 / max(T[i, 5], 1)))
