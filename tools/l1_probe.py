"""Per-kernel durations of the one-call L1 seams (run under rocprofv3 --kernel-trace; tools/l1_probe.sh prints the table):
eorb_ev_slice_extract / _track with float and with raw events for a few chunk sizes, and eorb_ev_mc_contest."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eorb_slam_amd import frontend as fe, synth, _lib

W, H = 240, 180
mx, my = synth.undistort_lut(W, H)
reps = int(os.environ.get("REPS", "30"))
for use_raw in (False, True):
    g = fe.EvImBuilder(W, H, cam=synth.EVETHZ_PINHOLE, raw_events=use_raw, keep_images=False)
    g.set_undistort_maps(mx, my, True)
    for n in (2000, 5000):
        pairs = [synth.shapes_events(n, W, H, seed=30 + k, motion=0.4 + 0.1 * k, undistort=True, return_raw=True) for k in range(4)]
        evs = [p[1] if use_raw else p[0] for p in pairs]
        t_ext, t_trk = [], []
        for r in range(reps):
            t0 = time.perf_counter(); kps, _ = g._slice_extract(evs[0]); t1 = time.perf_counter()
            pts = np.stack([kps["x"], kps["y"]], axis=1).astype(np.float32)
            for k in (1, 2, 3):
                t2 = time.perf_counter(); pts, st, er, _ = g._slice_track(evs[k], pts); t_trk.append(time.perf_counter() - t2)
            t_ext.append(t1 - t0)
        print("%s events, n=%d: slice_extract p50 %.3f ms, slice_track p50 %.3f ms (%d keypoints)" % ("raw" if use_raw else "float", n, np.median(t_ext) * 1e3, np.median(t_trk) * 1e3, len(kps)))
    if not use_raw:
        for n in (6000, 12000):
            win = synth.l1_stream(n_chunks=1, chunk=n, seed=3, motion=6.0)
            p = synth.l1_mci_poses(win)
            ts = []
            for r in range(reps):
                t0 = time.perf_counter(); g.generateMCImage(win, p); ts.append(time.perf_counter() - t0)
            print("contest, window of %d events: p50 %.3f ms" % (n, np.median(ts) * 1e3))
    g.close()
