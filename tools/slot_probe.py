"""Accumulation-only timing probe: B slices x N raw events through eorb_fe_run_batch_raw_dev, per-kernel HIP-event times.
usage: python tools/slot_probe.py [--dist shapes|uniform|hot] [--batch B] [--events N] [--form F] [--reps R]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eorb_slam_amd import frontend as fe, synth

ap = argparse.ArgumentParser()
ap.add_argument("--dist", default="shapes"); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--events", type=int, default=1000000)
ap.add_argument("--form", type=int, default=0); ap.add_argument("--reps", type=int, default=5); ap.add_argument("--uniq", type=int, default=8)
ap.add_argument("--rec", type=int, default=16, help="bytes per event record in HBM: 16 (eorb_raw_event) or 4 (eorb_raw_event4)")
a = ap.parse_args()
W, H, B, N = 240, 180, a.batch, a.events
mx, my = synth.undistort_lut(W, H)
if a.dist == "shapes":
    base = [synth.shapes_events(N, W, H, seed=2 + b, motion=0.5, undistort=True, return_raw=True)[1] for b in range(min(B, a.uniq))]
elif a.dist == "uniform":
    base = [synth.random_raw_events(N, W, H, seed=b) for b in range(min(B, a.uniq))]
    for r in base:
        r["x"] = np.clip(r["x"], 30, 210); r["y"] = np.clip(r["y"], 25, 155)
else:
    rng = np.random.default_rng(0)
    base = []
    for b in range(min(B, a.uniq)):
        r = synth.random_raw_events(N, W, H, seed=b)
        r["x"] = np.clip(rng.normal(120, 2.0, N), 0, W - 1); r["y"] = np.clip(rng.normal(90, 2.0, N), 0, H - 1)
        base.append(r)
blob = np.concatenate([base[b % len(base)] for b in range(B)])
if a.rec == 4: blob = fe.pack_raw_events4(blob)
RAW = 4 if a.rec == 4 else True
fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=B, max_events=N)
c, cap = fb.ctx, fb.cap
c.debug_option("gather_form", a.form)
fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=c)
d_ev = c.dev_alloc(blob.nbytes); c.upload(d_ev, blob)
d_img = c.dev_alloc(B * W * H); d_kp = c.dev_alloc(B * cap * 28); d_desc = c.dev_alloc(B * cap * 32)
d_n = c.dev_alloc(B * 4); d_m = c.dev_alloc(B * cap * 4); d_nm = c.dev_alloc(B * 4)
off = np.arange(B + 1, dtype=np.int64) * N
for it in range(2):
    fb.run_dev(d_ev, off, d_img, d_kp, d_desc, d_n, d_m, d_nm, raw=RAW)
c.sync()
c.prof_reset(); c.prof_enable(True)
for it in range(a.reps):
    fb.run_dev(d_ev, off, d_img, d_kp, d_desc, d_n, d_m, d_nm, raw=RAW)
c.sync()
c.prof_enable(False)
res = c.prof_results()
print("dist=%s B=%d N=%d form=%d rec=%d slot_calls=%d :: " % (a.dist, B, N, a.form, a.rec, c.debug_counter("slot_calls")) +
      " ".join("%s=%.3f" % (k, ms / a.reps) for k, (ms, n) in sorted(res.items()) if k.startswith("ev_")))
