"""Golden vectors (tests/golden/golden_v1.npz, produced by tests/golden/make_golden.py from the oracle):
the oracle must keep reproducing them (CPU) and the HIP path must match them bit for bit (GPU)."""
import hashlib
import os

import numpy as np
import pytest

from eorb_slam_amd import synth

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
W, H = 240, 180


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


CASES = (("l1", 2000, 1), ("l2", 6000, 1), ("dense", 40000, 2))


def _check_accumulate(ev2im_gauss, ev2im):
    for name, n, seed in CASES:
        ev = synth.shapes_events(n, W, H, seed=seed, undistort=True)
        assert _sha(ev) == str(G["acc_%s_events_sha" % name]), "synthetic generator drifted"
        f32, u8, mm = ev2im_gauss(ev, W, H, 1.0, False, True)
        assert np.array_equal(_bits(f32), _bits(G["acc_%s_f32" % name]))
        assert np.array_equal(u8, G["acc_%s_u8" % name]) and np.array_equal(_bits(mm), _bits(G["acc_%s_minmax" % name]))
    ev = synth.random_events(3000, W, H, seed=9)
    f32, u8, mm = ev2im_gauss(ev, W, H, 1.0, True, True)
    assert np.array_equal(_bits(f32), _bits(G["acc_pol_f32"])) and np.array_equal(u8, G["acc_pol_u8"])
    f32, u8, mm = ev2im(ev, W, H, False, True)
    assert np.array_equal(_bits(f32), _bits(G["cnt_f32"])) and np.array_equal(u8, G["cnt_u8"])


def _kp_bytes(kps):
    return np.ascontiguousarray(kps).view(np.uint8).reshape(len(kps), 28)


def test_oracle_reproduces_golden(oracle):
    _check_accumulate(oracle.ev2im_gauss, oracle.ev2im)
    e = oracle.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=19)
    for name, img in (("tex", synth.texture_image(W, H, seed=3)), ("evimg", G["acc_dense_u8"])):
        mono, kps, desc, oob = e.extract(img)
        assert np.array_equal(_kp_bytes(kps), G["orb_%s_kps" % name]) and np.array_equal(desc, G["orb_%s_desc" % name])
        assert mono == int(G["orb_%s_mono" % name])
    e1 = oracle.OrbExtractor(400, 1.0, 1, 0, 0, edgeTh=9)
    _, kps, _, _ = e1.extract(G["acc_l2_u8"], want_desc=False)
    assert np.array_equal(_kp_bytes(kps), G["orb_fast_kps"])
    img1 = synth.texture_image(W, H, seed=3); img2 = np.roll(img1, (3, -3), axis=(0, 1))
    _, k1, d1, _ = e.extract(img1); _, k2, d2, _ = e.extract(img2)
    pm = np.stack([k1["x"], k1["y"]], axis=1)
    n, m12, pm2 = oracle.search_for_initialization(oracle.Frame(k1, d1, W, H), oracle.Frame(k2, d2, W, H), pm, 100, 0.9, True)
    assert n == int(G["match_init_n"]) and np.array_equal(m12, G["match_init_m12"]) and np.array_equal(_bits(pm2), _bits(G["match_init_pm"]))
    t = synth.random_descriptors(300, seed=4); q, _ = synth.planted_descriptors(t, seed=5)
    idx, dist = oracle.bf_knn2(q, t)
    assert np.array_equal(idx, G["bf_idx"]) and np.array_equal(dist, G["bf_dist"])


@pytest.mark.gpu
def test_hip_path_reproduces_golden():
    from eorb_slam_amd import frontend as fe
    ctx = fe.Context()
    _check_accumulate(lambda *a: fe.EvImConverter.ev2im_gauss(*a, ctx=ctx, return_all=True),
                      lambda *a: fe.EvImConverter.ev2im(*a, ctx=ctx, return_all=True))
    e = fe.ORBextractor(1000, 1.2, 4, 10, 0, 19, (W, H), ctx=ctx)
    frames = {}
    for name, img in (("tex", synth.texture_image(W, H, seed=3)), ("evimg", G["acc_dense_u8"])):
        mono, kps, desc, oob = e(img)
        assert np.array_equal(_kp_bytes(kps), G["orb_%s_kps" % name]) and np.array_equal(desc, G["orb_%s_desc" % name])
        assert mono == int(G["orb_%s_mono" % name])
        frames[name] = (kps, desc)
    e1 = fe.ORBextractor(400, 1.0, 1, 0, 0, 9, (W, H), ctx=fe.Context())
    _, kps, _, _ = e1(G["acc_l2_u8"], want_desc=False)
    assert np.array_equal(_kp_bytes(kps), G["orb_fast_kps"])
    img2 = np.roll(synth.texture_image(W, H, seed=3), (3, -3), axis=(0, 1))
    k1, d1 = frames["tex"]
    _, k2, d2, _ = e(img2)
    pm = np.stack([k1["x"], k1["y"]], axis=1)
    n, m12, pm2 = fe.ORBmatcher(0.9, True, ctx).SearchForInitialization(fe.FrameView(k1, d1, W, H), fe.FrameView(k2, d2, W, H), pm, 100)
    assert n == int(G["match_init_n"]) and np.array_equal(m12, G["match_init_m12"]) and np.array_equal(_bits(pm2), _bits(G["match_init_pm"]))
    t = synth.random_descriptors(300, seed=4); q, _ = synth.planted_descriptors(t, seed=5)
    idx, dist = fe.BFMatcher(ctx).knnMatch2(q, t)
    assert np.array_equal(idx, G["bf_idx"]) and np.array_equal(dist, G["bf_dist"])
