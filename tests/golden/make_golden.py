#!/usr/bin/env python3
"""Generate the golden vectors in this directory from the CPU oracle.

The reference ships no tests or fixtures for this path and cannot be built/imported here (C++ needing
OpenCV 3.4.1 etc.), so these vectors are produced by the oracle (oracle/), whose pinning status is
described in oracle/README.md.  They freeze the oracle's behaviour: tests/test_golden.py checks the oracle
(CPU) and the HIP path (GPU) against them.  Inputs are regenerated from seeds by eorb_slam_amd.synth.

Run from the repository root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from eorb_slam_amd import synth  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    W, H = 240, 180
    g = {}
    # C1 stand-in: LUT-undistorted shapes events, L1 chunk and L2 window
    for name, n, seed in (("l1", 2000, 1), ("l2", 6000, 1), ("dense", 40000, 2)):
        ev = synth.shapes_events(n, W, H, seed=seed, undistort=True)
        f32, u8, mm = orc.ev2im_gauss(ev, W, H, 1.0, False, True)
        g["acc_%s_events_sha" % name] = sha(ev)
        g["acc_%s_f32" % name] = f32
        g["acc_%s_u8" % name] = u8
        g["acc_%s_minmax" % name] = mm
    ev = synth.random_events(3000, W, H, seed=9)
    f32, u8, mm = orc.ev2im_gauss(ev, W, H, 1.0, True, True)
    g["acc_pol_f32"], g["acc_pol_u8"], g["acc_pol_minmax"] = f32, u8, mm
    f32, u8, mm = orc.ev2im(ev, W, H, False, True)
    g["cnt_f32"], g["cnt_u8"] = f32, u8
    # extraction: texture frame (C3 stand-in) and the dense event frame (C2 shape)
    e = orc.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=19)
    for name, img in (("tex", synth.texture_image(W, H, seed=3)), ("evimg", g["acc_dense_u8"])):
        mono, kps, desc, oob = e.extract(img)
        g["orb_%s_kps" % name] = kps.view(np.uint8).reshape(len(kps), 28)
        g["orb_%s_desc" % name] = desc
        g["orb_%s_mono" % name] = np.int32(mono)
    e1 = orc.OrbExtractor(400, 1.0, 1, 0, 0, edgeTh=9)
    mono, kps, _, _ = e1.extract(g["acc_l2_u8"], want_desc=False)
    g["orb_fast_kps"] = kps.view(np.uint8).reshape(len(kps), 28)
    # matching: texture frame against its shifted copy
    img1 = synth.texture_image(W, H, seed=3)
    img2 = np.roll(img1, (3, -3), axis=(0, 1))
    _, k1, d1, _ = e.extract(img1)
    _, k2, d2, _ = e.extract(img2)
    pm = np.stack([k1["x"], k1["y"]], axis=1)
    n, m12, pm2 = orc.search_for_initialization(orc.Frame(k1, d1, W, H), orc.Frame(k2, d2, W, H), pm, 100, 0.9, True)
    g["match_init_n"], g["match_init_m12"], g["match_init_pm"] = np.int32(n), m12, pm2
    t = synth.random_descriptors(300, seed=4)
    q, _ = synth.planted_descriptors(t, seed=5)
    idx, dist = orc.bf_knn2(q, t)
    g["bf_idx"], g["bf_dist"] = idx, dist
    np.savez_compressed(os.path.join(OUT, "golden_v1.npz"), **g)
    print("wrote golden_v1.npz:", os.path.getsize(os.path.join(OUT, "golden_v1.npz")), "bytes,", len(g), "arrays")


if __name__ == "__main__":
    main()
