"""Randomised parity sweep of the raw-event accumulation (ev_gather_raw_kernel / ev_gather_sparse_kernel + binning) against the CPU oracle: image sizes that are
not multiples of the tile, sigmas up to the 17x17 stamp, polarity, maps that throw pixels out of the image with and without
checkInImage, event counts on the 64-entry batch boundaries, hot pixels.  Run on the GPU box: python tests/fuzz/fuzz_raw.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from eorb_slam_amd import frontend as fe, synth
from oracle import oracle_py as orc

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
c = fe.Context()
bad = 0
t0 = time.time()
for case in range(ncase):
    W, H = [(240, 180), (346, 260), (64, 48), (33, 17), (100, 9), (16, 16), (250, 131)][rng.integers(0, 7)]
    LW, LH = W + int(rng.integers(0, 5)), H + int(rng.integers(0, 5))              # sensor a little larger than the image
    sigma = float([0.1, 0.21, 0.4, 0.5, 0.8, 1.0, 1.0, 1.0, 1.3, 1.7, 2.0, 2.5][rng.integers(0, 12)])
    pol, check = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    if sigma < 0.2: pol = False                                                   # rejected with polarity (EORB_E_CONFIG)
    n = int([0, 1, 63, 64, 65, 127, 128, 129, 1000, 4096, 4097, 20000, 150000][rng.integers(0, 13)])
    yy, xx = np.mgrid[0:LH, 0:LW].astype(np.float64)
    amp = rng.uniform(0, 6)
    mx = (xx + amp * np.sin(yy / 17.0) + rng.uniform(-3, 3)).astype(np.float32)
    my = (yy + amp * np.cos(xx / 23.0) + rng.uniform(-3, 3)).astype(np.float32)
    raw = np.zeros(n, synth.RAW_DTYPE)
    mode = rng.integers(0, 3)
    if mode == 0:                                                                 # uniform
        raw["x"] = rng.integers(0, LW, n); raw["y"] = rng.integers(0, LH, n)
    elif mode == 1:                                                               # one hot spot (long chains on a few tiles)
        raw["x"] = np.clip(rng.normal(LW * rng.uniform(0, 1), 2.5, n), 0, LW - 1); raw["y"] = np.clip(rng.normal(LH * rng.uniform(0, 1), 2.5, n), 0, LH - 1)
    else:                                                                         # a single sensor pixel, then noise
        raw["x"] = rng.integers(0, LW); raw["y"] = rng.integers(0, LH)
        k = n // 3
        raw["x"][:k] = rng.integers(0, LW, k); raw["y"][:k] = rng.integers(0, LH, k)
        rng.shuffle(raw)
    raw["p"] = rng.integers(0, 2, n); raw["t"] = np.arange(n) * 1e-6
    form = int(rng.integers(0, 5))                                                # gather kernel: by shape / workgroup per tile / wave per tile / no binning / slot lists
    c.debug_option("gather_form", form)
    c.debug_option("dedupe_min_events", 1 if rng.integers(0, 3) == 0 else 1 << 20)     # a third of the cases: float events take the bulk (position table) form
    fe.EvImConverter.set_undistort_maps(mx, my, check, ctx=c)
    ev = orc.undistort_events(raw, mx, my, W, H, check, 1.0)
    of, ou, omm = orc.ev2im_gauss(ev, W, H, sigma, pol, True)
    gf, gu, gmm = fe.EvImConverter.ev2im_gauss_raw(raw, W, H, sigma, pol, True, ctx=c, return_all=True)
    ok = np.array_equal(of.view(np.uint32), gf.view(np.uint32)) and np.array_equal(ou, gu) and \
        np.array_equal(np.asarray(omm, np.float32).view(np.uint32), gmm.view(np.uint32))
    if n <= 20000:                                                                # the float-event path (list pipeline K2) on the same events
        hf, hu, hmm = fe.EvImConverter.ev2im_gauss(ev, W, H, sigma, pol, True, ctx=c, return_all=True)
        ok = ok and np.array_equal(of.view(np.uint32), hf.view(np.uint32)) and np.array_equal(ou, hu) and \
            np.array_equal(np.asarray(omm, np.float32).view(np.uint32), hmm.view(np.uint32))
    if not ok:
        bad += 1
        print("MISMATCH case", case, dict(W=W, H=H, LW=LW, LH=LH, sigma=sigma, pol=pol, check=check, n=n, mode=int(mode), form=form),
              "pixels", int((of.view(np.uint32) != gf.view(np.uint32)).sum()), "minmax", omm, gmm, flush=True)
    if case % 25 == 24:
        print("case", case + 1, "bad", bad, "%.0f s" % (time.time() - t0), flush=True)
print("fuzz_raw: %d cases, %d mismatches" % (ncase, bad))
sys.exit(1 if bad else 0)
