"""Host-side cost of a batched call: time inside eorb_fe_run_batch_*_dev (enqueue + the waits the call itself makes) vs the step time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eorb_slam_amd import frontend as fe, synth
W, H, B, N = 240, 180, 128, 1000000
NU = int(sys.argv[1]) if len(sys.argv) > 1 else 16
USE_TORCH = len(sys.argv) > 2
mx, my = synth.undistort_lut(W, H)
pairs = [synth.shapes_events(N, W, H, seed=2 + b, motion=0.5, undistort=True, return_raw=True) for b in range(NU)]
for mode in ("raw", "float"):
    blob = np.concatenate([(pairs[b % NU][1] if mode == "raw" else fe.pack_events(pairs[b % NU][0])) for b in range(B)])
    if USE_TORCH:
        import torch
        st = torch.cuda.Stream()
        cx = fe.Context(device=0, stream=st.cuda_stream)
        fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=B, max_events=N, ctx=cx)
    else:
        fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=B, max_events=N)
    c, cap = fb.ctx, fb.cap
    fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=c)
    d_ev = c.dev_alloc(blob.nbytes); c.upload(d_ev, blob)
    d_img = c.dev_alloc(B * W * H); d_kp = c.dev_alloc(B * cap * 28); d_desc = c.dev_alloc(B * cap * 32)
    d_n = c.dev_alloc(B * 4); d_m = c.dev_alloc(B * cap * 4); d_nm = c.dev_alloc(B * 4)
    off = np.arange(B + 1, dtype=np.int64) * N
    for it in range(3):
        fb.run_dev(d_ev, off, d_img, d_kp, d_desc, d_n, d_m, d_nm, raw=(mode == "raw"))
    c.sync()
    calls = []
    t0 = time.perf_counter()
    for it in range(30):
        t1 = time.perf_counter()
        fb.run_dev(d_ev, off, d_img, d_kp, d_desc, d_n, d_m, d_nm, raw=(mode == "raw"))
        calls.append(time.perf_counter() - t1)
    c.sync()
    tot = (time.perf_counter() - t0) / 30
    print("%s: step %.2f ms, inside the call %.2f ms (min %.2f, max %.2f at call %d)" % (mode, tot * 1e3, np.mean(calls) * 1e3, np.min(calls) * 1e3, np.max(calls) * 1e3, int(np.argmax(calls)) + 4))
    c.close()
