"""Randomised parity sweep of the ORB extractor (pyramid, FAST cells, octree, orientation, blur, rBRIEF, output order) against the
CPU oracle: random image sizes / contents / extractor parameters.  Configurations the library rejects (EORB_E_CONFIG /
EORB_E_CAPACITY) are counted, not failed.  Run on the GPU box: python tests/fuzz/fuzz_orb.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from eorb_slam_amd import frontend as fe, synth
from oracle import oracle_py as orc

ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4321)
bad = rejected = 0
t0 = time.time()
for case in range(ncase):
    W, H = [(240, 180), (346, 260), (200, 150), (320, 240), (173, 131), (256, 256)][rng.integers(0, 6)]
    kind = rng.integers(0, 5)
    if kind == 0:
        img = synth.texture_image(W, H, seed=int(rng.integers(0, 1 << 30)))
    elif kind == 1:
        img = rng.integers(0, 256, (H, W)).astype(np.uint8)                       # white noise: candidates everywhere, many ties
    elif kind == 2:
        img = (rng.integers(0, 2, (H // 8 + 1, W // 8 + 1)) * 255).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:H, :W].copy()   # blocks
    elif kind == 3:
        ev = synth.shapes_events(int(rng.integers(2000, 80000)), W, H, seed=int(rng.integers(0, 1 << 30)))
        img = orc.ev2im_gauss(ev, W, H, 1.0, False, True)[1]                      # event image
    else:
        img = np.clip(synth.texture_image(W, H, seed=int(rng.integers(0, 1 << 30))).astype(np.int32) // 16 * 16 + rng.integers(-2, 3, (H, W)), 0, 255).astype(np.uint8)
    p = dict(nfeatures=int([100, 400, 800, 1000, 2000, 3000][rng.integers(0, 6)]),
             scaleFactor=float([1.0, 1.1, 1.2, 1.2, 1.3, 1.5][rng.integers(0, 6)]),
             nlevels=int(rng.integers(1, 9)), iniThFAST=int([0, 5, 10, 20, 40][rng.integers(0, 5)]),
             minThFAST=0, edgeTh=int([9, 15, 19, 19, 21, 25][rng.integers(0, 6)]))
    p["minThFAST"] = int(rng.integers(0, p["iniThFAST"] + 1))
    if p["scaleFactor"] == 1.0: p["nlevels"] = 1
    lap = [(0, 1000), (0, 0), (W // 4, W // 2)][rng.integers(0, 3)]
    want_desc = bool(rng.integers(0, 4) > 0)
    try:
        ge = fe.ORBextractor(imSize=(W, H), **p)
        gmono, gkp, gdesc, goob = ge(img, lap, want_desc)
    except fe.EorbError as e:
        rejected += 1
        print("rejected", dict(W=W, H=H, **p), str(e)[:140], flush=True)
        continue
    finally:
        try: ge.ctx.close()
        except Exception: pass
    oe = orc.OrbExtractor(imWidth=W, **p)
    omono, okp, odesc, ooob = oe.extract(img, lap, want_desc)
    ok = omono == gmono and len(okp) == len(gkp)
    if ok:
        for f in ("x", "y", "size", "angle", "response"):
            ok = ok and np.array_equal(okp[f].view(np.uint32), gkp[f].view(np.uint32))
        ok = ok and np.array_equal(okp["octave"], gkp["octave"]) and np.array_equal(okp["class_id"], gkp["class_id"])
        if want_desc:
            ok = ok and np.array_equal(odesc, gdesc) and np.array_equal(ooob, goob)
    if not ok:
        bad += 1
        print("MISMATCH case", case, dict(W=W, H=H, kind=int(kind), lap=lap, desc=want_desc, **p), "n", len(okp), len(gkp), flush=True)
    if case % 20 == 19:
        print("case", case + 1, "bad", bad, "rejected", rejected, "%.0f s" % (time.time() - t0), flush=True)
print("fuzz_orb: %d cases, %d mismatches, %d rejected configurations" % (ncase, bad, rejected))
sys.exit(1 if bad else 0)
