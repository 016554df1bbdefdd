"""Known-answer tests pinning the oracle to values hand-derived from the reference source
(the reference has no tests or fixtures of its own)."""
import ctypes as C
import math

import numpy as np
import pytest

from eorb_slam_amd import synth


def test_math_vs_host_libm(oracle):
    """orc_expf / orc_sinf / orc_cosf restate glibc's algorithms; compare with the host libm."""
    L = oracle.lib()
    libm = C.CDLL("libm.so.6")
    for f in ("expf", "sinf", "cosf"):
        getattr(libm, f).restype = C.c_float; getattr(libm, f).argtypes = [C.c_float]
    rng = np.random.default_rng(0)
    xs = np.concatenate([-rng.uniform(0, 32, 20000), -np.linspace(0, 18, 5000)]).astype(np.float32)
    bad = sum(1 for x in xs if np.float32(L.orc_expf(float(x))).tobytes() != np.float32(libm.expf(float(x))).tobytes())
    assert bad == 0
    ang = np.concatenate([rng.uniform(0, 6.3, 20000), np.linspace(0, 6.2832, 5000)]).astype(np.float32)
    for x in ang:
        assert L.orc_sinf(float(x)) == libm.sinf(float(x))
        assert L.orc_cosf(float(x)) == libm.cosf(float(x))
    # the single known disagreement with glibc's fma variant (oracle = correctly rounded there)
    assert L.orc_expf(float.fromhex("-0x1.f8cbb2p+5")) == float.fromhex("0x1.f45324p-92")


def test_exp2_table_is_exact(oracle):
    # tab[i] = bits(2^(i/32)) - (i << 47); re-derive each entry with exact rational bracketing
    from fractions import Fraction
    import struct
    L = oracle.lib()
    # indirect check: orc_expf(x) for x = ln2 * i/32 must be within 1 ulp of 2^(i/32)
    for i in range(32):
        x = np.float32(math.log(2.0) * i / 32)
        ref = math.exp(float(x))
        got = L.orc_expf(float(x))
        assert abs(got - ref) <= abs(ref) * 2 ** -23


def test_cvround_half_even(oracle):
    L = oracle.lib()
    assert [L.orc_cvround(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def test_fast_atan2(oracle):
    L = oracle.lib()
    assert L.orc_fast_atan2(0.0, 1.0) == 0.0
    assert abs(L.orc_fast_atan2(1.0, 1.0) - 45.0) < 0.02
    assert abs(L.orc_fast_atan2(1.0, 0.0) - 90.0) < 0.02
    assert abs(L.orc_fast_atan2(0.0, -1.0) - 180.0) < 0.02
    assert abs(L.orc_fast_atan2(-1.0, 0.0) - 270.0) < 0.02
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = rng.normal(size=2) * 1000
        a = L.orc_fast_atan2(float(y), float(x))
        t = math.degrees(math.atan2(y, x)) % 360
        d = abs(a - t); d = min(d, 360 - d)
        assert d < 0.3 and 0 <= a <= 360      # OpenCV documents ~0.3 deg accuracy


def test_extractor_ctor_tables(oracle):
    """src/ORBextractor.cc:420-489 — values derived by hand in SURVEY.md §4 / §8(c)."""
    e = oracle.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=19, imWidth=240)
    assert e.features_per_level == [322, 268, 224, 186]
    assert e.umax == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    sf = e.scale_factors
    assert sf[0] == 1.0 and sf[1] == np.float32(np.float64(np.float32(1.2)))
    assert sf[2] == np.float32(np.float64(sf[1]) * np.float64(np.float32(1.2)))
    # adaptive edge threshold rule :481-488 -> 19*(240/752)=6.06 -> 6 -> made odd: 5
    e2 = oracle.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=-1, imWidth=240)
    assert e2.edge == 5
    e3 = oracle.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=-1, imWidth=752)
    assert e3.edge == 19
    # single level ("FAST mode", EvBaseTracker.cpp:150-164): last level takes everything
    e4 = oracle.OrbExtractor(400, 1.0, 1, 0, 0, edgeTh=9)
    assert e4.features_per_level == [400]


def test_pyramid_sizes(oracle):
    img = synth.texture_image(240, 180, seed=3)
    e = oracle.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=19)
    mono, kps, desc, oob = e.extract(img)
    assert [e.level_size(l) for l in range(4)] == [(240, 180), (200, 150), (167, 125), (139, 104)]
    e8 = oracle.OrbExtractor(2000, 1.2, 8, 10, 0, edgeTh=15)
    img2 = synth.texture_image(346, 260, seed=4)
    e8.extract(img2)
    sizes = [e8.level_size(l) for l in range(8)]
    assert sizes[0] == (346, 260) and sizes[7] == (97, 73)


def test_gauss_kernel_q8(oracle):
    k = np.zeros(5, np.int32)
    oracle.lib().orc_gauss_kernel_q8(5, 2.0, k.ctypes.data_as(C.c_void_p))
    assert list(k) == [39, 57, 64, 57, 39] and k.sum() == 256
    flat = np.full((20, 30), 77, np.uint8)
    assert (oracle.gaussian_blur5(flat) == 77).all()


def test_resize_identity_and_constant(oracle):
    img = synth.texture_image(64, 48, seed=9)
    assert (oracle.resize_linear(img, 64, 48) == img).all()
    flat = np.full((48, 64), 200, np.uint8)
    assert (oracle.resize_linear(flat, 53, 40) == 200).all()


def test_descriptor_distance_identities(oracle):
    z = np.zeros(32, np.uint8); f = np.full(32, 255, np.uint8)
    assert oracle.descriptor_distance(z, z) == 0
    assert oracle.descriptor_distance(z, f) == 256
    rng = np.random.default_rng(2)
    for _ in range(200):
        a = rng.integers(0, 256, 32, dtype=np.uint8); b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert oracle.descriptor_distance(a, b) == int(np.unpackbits(a ^ b).sum())
    # AKAZE rows are 61 B wide, only the first 32 B count
    a = rng.integers(0, 256, 61, dtype=np.uint8); b = a.copy(); b[40] ^= 0xFF
    assert oracle.descriptor_distance(a, b) == 0


def test_three_maxima(oracle):
    s = [0] * 30
    s[3], s[7], s[20] = 50, 40, 4            # third < 10 % of first -> dropped
    assert oracle.three_maxima(s) == (3, 7, -1)
    s[20] = 5
    assert oracle.three_maxima(s) == (3, 7, 20)
    s[7] = 4; s[20] = 3                      # second < 10 % -> both dropped
    assert oracle.three_maxima(s) == (3, -1, -1)


def test_gauss_stamp_centre_value(oracle):
    """One event at an integer pixel, sigma=1: centre tap = 1/(2*pi) (EventConversion.cc:59-65)."""
    ev = oracle.make_events([100.0], [90.0])
    f32, u8, mm = oracle.ev2im_gauss(ev, 240, 180, 1.0, False, True)
    assert f32[90, 100] == np.float32(1.0) / (np.float32(2.0) * np.float32(math.pi) * np.float32(1.0))
    assert np.count_nonzero(f32) == 49 and u8[90, 100] == 255 and mm[0] == 0.0 and mm[1] == f32[90, 100]
    # symmetric stamp
    assert f32[90, 97] == f32[90, 103] == f32[87, 100] == f32[93, 100]


def test_ev2im_count_image(oracle):
    ev = oracle.make_events([10.4, 10.6, 10.5, 300.0], [5.0, 5.0, 5.0, 5.0], p=[1, 0, 1, 1])
    f32, u8, mm = oracle.ev2im(ev, 240, 180, pol=False, normalized=False)
    v = np.float32(0)
    assert f32[5, 10] == np.float32(0.001)
    assert f32[5, 11] == np.float32(np.float32(0.001) + np.float32(0.001))   # 10.6->11, 10.5->roundf away: 11
    assert u8 is None and np.count_nonzero(f32) == 2
    f32p, _, mmp = oracle.ev2im(ev, 240, 180, pol=True, normalized=False)
    assert f32p[5, 11] == np.float32(np.float32(-0.001) + np.float32(0.001))
    assert mmp[0] == np.float32(-0.001)


def test_empty_events_normalised_is_zero(oracle):
    """N=0, normalized: alpha = 255/(-1e6-0) on an all-zero image -> all-zero u8."""
    ev = np.zeros(0, oracle.EVENT_DTYPE)
    f32, u8, mm = oracle.ev2im_gauss(ev, 64, 48)
    assert (f32 == 0).all() and (u8 == 0).all()


def test_fast_on_synthetic_corner(oracle):
    img = np.full((32, 32), 20, np.uint8)
    img[16, 16] = 200                           # isolated bright pixel: all 16 ring pixels darker
    k = oracle.fast9_16(img, 10)
    assert k.tolist() == [[16, 16, 179]]        # score = max t that keeps it a corner = 180 - 1
    img[16, 17] = 200                           # plateau: equal scores suppress each other (strict >)
    assert len(oracle.fast9_16(img, 10)) == 0
    img[16, 17] = 190                           # weaker neighbour loses NMS
    k = oracle.fast9_16(img, 10)
    assert k.tolist() == [[16, 16, 179]]
    # every detection is a strict 3x3 local maximum of the score and >= 3 px from the border
    assert ((k[:, 0] >= 3) & (k[:, 0] < 29) & (k[:, 1] >= 3) & (k[:, 1] < 29)).all()
    assert (oracle.fast9_16(np.full((32, 32), 99, np.uint8), 0).shape[0]) == 0


def test_octree_simple(oracle):
    # 4 well separated points, N=4 -> all four kept, output order = final std::list order
    kp = np.zeros(4, oracle.KP_DTYPE)
    kp["x"] = [10, 150, 20, 160]; kp["y"] = [10, 20, 120, 130]; kp["response"] = [5, 6, 7, 8]
    out = oracle.distribute_octree(kp, 0, 200, 0, 150, 4)
    assert len(out) == 4
    # one root (round(200/150)=1); children pushed front in order n1,n2,n3,n4 -> list = n4,n3,n2,n1
    assert list(out["response"]) == [8, 7, 6, 5]
    # two points in the same leaf region with N=1: best response survives
    kp2 = np.zeros(2, oracle.KP_DTYPE); kp2["x"] = [10, 11]; kp2["y"] = [10, 10]; kp2["response"] = [3, 9]
    out2 = oracle.distribute_octree(kp2, 0, 200, 0, 150, 1)
    assert len(out2) >= 1 and out2["response"].max() == 9


def test_extract_output_order_is_reversed(oracle):
    """SURVEY App.B H12: mono callers pass lapping {0,1000}: every kp takes the back-fill branch."""
    img = synth.texture_image(240, 180, seed=3)
    e = oracle.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=19)
    mono, kps, desc, oob = e.extract(img, lap=(0, 1000))
    assert mono == 0 and len(kps) > 300
    l0 = e.level_keypoints(0)
    n = len(kps)
    assert kps[n - 1]["x"] == l0[0]["x"] and kps[n - 1]["y"] == l0[0]["y"]
    assert (np.diff(kps["octave"]) <= 0).all()          # levels descending after reversal
    mono2, kps2, desc2, _ = e.extract(img, lap=(0, 0))
    assert mono2 == len(kps2) - int((kps2["x"] == 0).sum())
    assert (kps2["octave"][:mono2][:-1] <= kps2["octave"][:mono2][1:]).all()
    assert not oob.any()                                # edge 19: no tap leaves the level buffer


def test_extract_empty_image_returns_minus_one(oracle):
    e = oracle.OrbExtractor()
    L = oracle.lib()
    n = C.c_int()
    assert L.orc_orb_extract(e.h, None, 0, 0, 0, 0, 1000, 1, None, None, None, 0, C.byref(n)) == -1


# ---- known answers for the "next" rows (loader, vocabulary, KeyFrame matchers, LK) derived by hand from the reference -------------

def test_loader_text_and_rectification_known_answers(oracle):
    """EventLoader.cpp:80-92 "stream >> ts >> x >> y >> p", isComment DataStore.cpp:111-114, ts/tsFactor :120-123,
    undistPointMaps MyCalibrator.cpp:164-180, isInImage :31-34."""
    raw = oracle.parse_events_text(b"# header\n0.003811000 96 133 0\n  1468941032.229165\t13 7 1\r\n\n12.5 3 4 1")
    assert raw["x"].tolist() == [96, 13, 3] and raw["y"].tolist() == [133, 7, 4] and raw["p"].tolist() == [0, 1, 1]
    assert raw["t"].tolist() == [0.003811, 1468941032.229165, 12.5]          # strtod of the same literals
    mx = np.arange(20, dtype=np.float32).reshape(1, 20).repeat(10, 0) + 0.25   # map: x + 0.25, y - 0.5
    my = np.arange(10, dtype=np.float32).reshape(10, 1).repeat(20, 1) - 0.5
    r = np.zeros(3, oracle.RAW_DTYPE); r["x"] = [3, 19, 5]; r["y"] = [4, 9, 0]; r["p"] = [1, 0, 1]; r["t"] = [2e6, 4e6, 6e6]
    ev = oracle.undistort_events(r, mx, my, 20, 10, True, 1e6)
    # (5, 0) maps to y = -0.5: outside the image, dropped; the others keep their order
    assert ev["x"].tolist() == [3.25, 19.25] and ev["y"].tolist() == [3.5, 8.5] and ev["ts"].tolist() == [2.0, 4.0] and ev["p"].tolist() == [1, 0]
    assert len(oracle.undistort_events(r, mx, my, 20, 10, False, 1e6)) == 3
    with pytest.raises(ValueError):
        oracle.parse_events_text(b"1.0 2 3 1\n1.0 2 3 7\n")


def test_bow_transform_known_answers(oracle):
    """TemplatedVocabulary::transform :1125-1250 on a 2-level binary tree: descent by Hamming distance, addWeight in feature
    order, L1 normalisation, FeatureVector at the level above the words."""
    z = np.zeros(32, np.uint8); f = np.full(32, 255, np.uint8)
    half = z.copy(); half[:16] = 255
    q1 = z.copy(); q1[:4] = 255; q3 = f.copy(); q3[:4] = 0
    # nodes: 0 root; 1 = 000.., 2 = 111..; words: 3 (child of 1) = 000.., 4 (child of 1) = 32 leading ones; 5 (of 2) = 111.., 6 (of 2) = half
    voc = dict(L=2, child_off=[0, 2, 4, 6, 6, 6, 6, 6], child_ids=[1, 2, 3, 4, 5, 6],
               node_desc=np.stack([z, z, f, z, q1, f, half]), word_id=[-1, -1, -1, 0, 1, 2, 3],
               weight=[0, 0, 0, 2.0, 1.0, 4.0, 0.0])
    h2 = half.copy(); h2[16] = 255                # 136 ones: nearer to node 2, then 8 bits from the stopped word's descriptor
    desc = np.stack([z, z, q1, f, h2])           # words 0, 0, 1, 2 and the stopped word 3
    bw, bv, fv, wo, no = oracle.bow_transform(voc, desc, levelsup=1, weighting=0, norm=1)
    assert wo.tolist() == [0, 0, 1, 2, -1] and no.tolist() == [1, 1, 1, 2, -1]
    assert bw.tolist() == [0, 1, 2] and bv.tolist() == [4.0 / 9.0, 1.0 / 9.0, 4.0 / 9.0]     # (2+2, 1, 4) / 9
    assert fv[0].tolist() == [1, 2] and fv[1].tolist() == [0, 3, 4] and fv[2].tolist() == [0, 1, 2, 3]
    # a tie between siblings goes to the first child (strict '<' :1229): `half` is 128 bits from both level-1 nodes -> node 1,
    # then 128 from word 0 and 96 from word 1
    assert oracle.bow_transform(voc, half[None], 1, 0, 1)[3].tolist() == [1]
    tie = z.copy(); tie[:2] = 255                 # 16 bits from word 0 and 16 from word 1 -> the first
    assert oracle.bow_transform(voc, tie[None], 1, 0, 1)[3].tolist() == [0]
    # BINARY weighting keeps the first weight, no normalisation -> the idf weights themselves
    bw, bv, _, _, _ = oracle.bow_transform(voc, desc, 1, 3, 0)
    assert bv.tolist() == [2.0, 1.0, 4.0]


def test_distinctive_descriptor_and_window_match_known_answers(oracle):
    """MapPoint.cc:349-423: the descriptor with the least median distance to the others; ORBmatcher.cc:754-774 best / second."""
    a = np.zeros(32, np.uint8)
    b = a.copy(); b[0] = 0xFF            # 8 bits from a
    c = a.copy(); c[:2] = 0xFF           # 16 from a, 8 from b
    # medians (index floor(0.5*(N-1)) = 1 of the sorted rows incl. the zero self-distance): a -> 8, b -> 8, c -> 8 ... first wins
    assert oracle.distinctive_descriptors(np.stack([a, b, c]), [0, 3]).tolist() == [0]
    d = a.copy(); d[:4] = 0xFF           # 32 from a, 24 from b, 16 from c
    # rows sorted: a [0,8,16,32], b [0,8,8,24], c [0,8,16,16], d [0,16,24,32]; median index 1 -> all 8 except d (16): first wins
    assert oracle.distinctive_descriptors(np.stack([a, b, c, d]), [0, 4]).tolist() == [0]
    # five descriptors: median index 2: a [0,8,16,32,40]->16, b [0,8,8,24,32]->8
    e = a.copy(); e[:5] = 0xFF
    assert oracle.distinctive_descriptors(np.stack([a, b, c, d, e]), [0, 5]).tolist() == [1]
    bi, bd, si, sd = oracle.hamming_window_match(np.stack([a]), np.stack([c, b, b, a]), [0, 3], [0, 1, 2])
    assert (bi[0], bd[0], si[0], sd[0]) == (1, 8, 2, 8)                      # first of the two equal candidates is "best"
    bi, bd, si, sd = oracle.hamming_window_match(np.stack([a]), np.stack([c, b]), [0, 0], [])
    assert (bi[0], bd[0], si[0], sd[0]) == (-1, 256, -1, 256)


def test_epipolar_and_triangulation_known_answers(oracle):
    """Pinhole::epipolarConstrain Pinhole.cpp:142-156 with F for a pure x-translation (lines y2 = y1): dsqr = (y2-y1)^2 < 3.84*unc;
    SearchForTriangulation :1066-1145 keeps the LAST of equally distant passing candidates."""
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    kp = np.zeros(1, oracle.KP_DTYPE); kp["x"] = 50; kp["y"] = 40; kp["angle"] = 10
    k2 = np.zeros(4, oracle.KP_DTYPE); k2["x"] = [80, 90, 100, 110]; k2["y"] = [40, 41.9, 42.1, 40]; k2["angle"] = 10
    d1 = np.zeros((1, 32), np.uint8); d2 = np.zeros((4, 32), np.uint8)
    fv1 = ([7], [0, 1], [0]); fv2 = ([7], [0, 4], [0, 1, 2, 3])
    one = np.ones(1, np.float32)
    n, m = oracle.search_for_triangulation(kp, d1, [1], fv1, k2, d2, [1, 1, 1, 1], fv2, (-1000.0, 0.0), F, one, one, False, True)
    assert n == 1 and m.tolist() == [3]                      # candidates 0, 1, 3 pass (1.9^2 = 3.61 < 3.84), all distance 0 -> the last
    n, m = oracle.search_for_triangulation(kp, d1, [1], fv1, k2[:3], d2[:3], [1, 1, 1], fv2[:1] + ([0, 3], [0, 1, 2]), (-1000.0, 0.0), F, one, one, False, True)
    assert m.tolist() == [1]                                 # 2.1^2 = 4.41 fails
    n, m = oracle.search_for_triangulation(kp, d1, [1], fv1, k2, d2, [1, 1, 1, 1], fv2, (109.0, 40.0), F, one, one, False, True)
    assert m.tolist() == [1]                                 # candidate 3 lies within sqrt(100*scale) = 10 px of the epipole
    n, m = oracle.search_for_triangulation(kp, d1, [0], fv1, k2, d2, [1, 1, 1, 1], fv2, (-1000.0, 0.0), F, one, one, False, True)
    assert n == 0 and m.tolist() == [-1]                     # pKF1 feature already has a map point


def test_pyr_lk_known_answers(oracle):
    """calcOpticalFlowPyrLK: a smooth pattern shifted by whole pixels is tracked to the shift; a flat patch fails the minimum
    eigenvalue test; a point whose window leaves the padded image loses its status."""
    yy, xx = np.mgrid[0:120, 0:160]
    img1 = (127 + 60 * np.sin(xx / 7.0) * np.cos(yy / 9.0) + 40 * np.sin((xx + 2 * yy) / 11.0)).astype(np.uint8)
    img2 = np.roll(img1, (1, 2), axis=(0, 1))                # content moves +2 in x, +1 in y
    pts = np.array([[60.0, 50.0], [80.5, 70.25], [100.0, 40.0]], np.float32)
    n, st, er = oracle.calc_optical_flow_pyr_lk(img1, img2, pts, None, 23, 1, 10, 0.03, 0)
    assert st.tolist() == [1, 1, 1] and np.abs(n - pts - np.float32([2, 1])).max() < 0.05 and er.max() < 1.0
    flat = np.full((120, 160), 77, np.uint8)
    n, st, er = oracle.calc_optical_flow_pyr_lk(flat, flat, pts[:1], None, 23, 1, 10, 0.03, 0)
    assert st.tolist() == [0]                                # A = 0: minEig < 1e-4
    n, st, er = oracle.calc_optical_flow_pyr_lk(img1, img2, np.float32([[-40.0, 50.0]]), None, 23, 1, 10, 0.03, 0)
    assert st.tolist() == [0] and er.tolist() == [0.0]
    # maxLevel is cut where the next level would not exceed the window (buildOpticalFlowPyramid): 160x120 -> 80x60 -> 40x30 -> 20x15 (<= 23)
    a = oracle.calc_optical_flow_pyr_lk(img1, img2, pts, None, 23, 2, 10, 0.03, 0)
    b = oracle.calc_optical_flow_pyr_lk(img1, img2, pts, None, 23, 6, 10, 0.03, 0)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_kannala_brandt8_known_answers(oracle):
    """tanf / atanf / atan2f restatements against exactly representable answers and the host libm on spot values (the exhaustive
    comparison is oracle/check_libm.c), and the warp's unproject -> project round trip with no motion."""
    import math
    L = oracle.lib()
    assert L.orc_atanf(1.0) == np.float32(math.pi / 4) and L.orc_atan2f(1.0, 1.0) == np.float32(math.pi / 4)
    assert L.orc_atan2f(0.0, -1.0) == np.float32(math.pi) and L.orc_atan2f(-1.0, 0.0) == np.float32(-math.pi / 2)
    assert L.orc_tanf(0.0) == 0.0 and L.orc_atanf(0.0) == 0.0
    rng = np.random.default_rng(3)
    import ctypes as C
    libm = C.CDLL("libm.so.6")                      # the host glibc (numpy's float32 loops are its own SIMD code, not libm)
    for f in (libm.tanf, libm.atanf):
        f.restype = C.c_float; f.argtypes = [C.c_float]
    libm.atan2f.restype = C.c_float; libm.atan2f.argtypes = [C.c_float, C.c_float]
    for x in rng.uniform(-2.3, 2.3, 2000).astype(np.float32):
        assert L.orc_tanf(float(x)) == libm.tanf(float(x)), x
        assert L.orc_atanf(float(x)) == libm.atanf(float(x)), x
    ys = rng.normal(0, 50, 2000).astype(np.float32); xs = rng.normal(0, 50, 2000).astype(np.float32)
    for y, x in zip(ys, xs):
        assert L.orc_atan2f(float(y), float(x)) == libm.atan2f(float(y), float(x)), (y, x)
    # identity motion (angle 0, t 0, depth 1): every event must come back to its own pixel up to the float round trip
    from tests.test_gpu_parity import MVSEC_KB8
    W, H = 346, 260
    ev = np.zeros(400, __import__("eorb_slam_amd.synth", fromlist=["EVENT_DTYPE"]).EVENT_DTYPE)
    ev["x"] = rng.uniform(5, W - 5, 400).astype(np.float32); ev["y"] = rng.uniform(5, H - 5, 400).astype(np.float32)
    ev["ts"] = np.arange(400) * 1e-5
    uv = np.zeros((400, 2), np.float32)
    cam = oracle._camera(MVSEC_KB8)
    L.orc_mci_warp_se3_cam.restype = None
    L.orc_mci_warp_se3_cam(ev.ctypes.data_as(C.c_void_p), C.c_size_t(400), C.byref(cam), C.c_double(0.0),
                           np.array([0.0, 0.0, 1.0]).ctypes.data_as(C.c_void_p), np.zeros(3).ctypes.data_as(C.c_void_p),
                           C.c_float(1.0), None, uv.ctypes.data_as(C.c_void_p))
    assert np.abs(uv[:, 0] - ev["x"]).max() < 2e-3 and np.abs(uv[:, 1] - ev["y"]).max() < 2e-3


def test_projection_stereo_gate_known_answers(oracle):
    """src/ORBmatcher.cc:2056-2062 / :96-104 on a hand-built case: one projected point, two candidates in its window; the nearer
    descriptor has a right coordinate off by more than the radius and drops out, the other (within the radius, or without a right
    match at all) wins."""
    e = oracle.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=19)
    _, k, d, _ = e.extract(synth.texture_image(240, 180, seed=5))
    k = k[:3].copy(); d = d[:3].copy()
    k["octave"] = 0; k["angle"] = 0.0
    k["x"] = [100.0, 101.0, 180.0]; k["y"] = [90.0, 90.0, 40.0]
    q = np.zeros(32, np.uint8)
    d[0] = q; d[0, 0] = 0x01                                           # distance 1: the mono winner
    d[1] = q; d[1, 0] = 0x07                                           # distance 3
    d[2] = 0xFF
    last = k[:1].copy(); last["x"] = 100.5; last["y"] = 90.0
    cur_mp = np.full(3, -1, np.int32)
    args = (oracle.Frame(k, d, 240, 180), oracle.Frame(last, q[None], 240, 180), np.ones(1, np.uint8), np.array([[100.5, 90.0]], np.float32),
            q[None], np.ones(1, np.uint8), cur_mp, 7.0, np.ones(1, np.float32), 0, False)
    n, cm = oracle.search_by_projection_last(*args)
    assert n == 1 and cm.tolist() == [0, -1, -1]
    # radius = th * scale = 7: candidate 0 at |80 - 90| = 10 > 7 is gated out, candidate 1 at |80 - 85| = 5 stays
    n, cm = oracle.search_by_projection_last(*args, uright=np.array([90.0, 85.0, -1.0], np.float32), proj_ur=np.array([80.0], np.float32))
    assert n == 1 and cm.tolist() == [-1, 0, -1]
    # exactly on the radius is kept (er > radius), no right match (<= 0) is never gated
    n, cm = oracle.search_by_projection_last(*args, uright=np.array([87.0, 85.0, -1.0], np.float32), proj_ur=np.array([80.0], np.float32))
    assert cm.tolist() == [0, -1, -1]
    n, cm = oracle.search_by_projection_last(*args, uright=np.array([0.0, 85.0, -1.0], np.float32), proj_ur=np.array([80.0], np.float32))
    assert cm.tolist() == [0, -1, -1]
    # the map form: r = th * RadiusByViewingCos(0.999) = 2.5, gate r * scale
    margs = (oracle.Frame(k, d, 240, 180), np.ones(1, np.uint8), np.array([[100.5, 90.0]], np.float32), np.zeros(1, np.int32),
             np.array([0.9995], np.float32), q[None], np.ones(1, np.uint8), cur_mp, 1.0, 0.8, np.ones(1, np.float32))
    n, fm = oracle.search_by_projection_map(*margs)
    assert n == 1 and fm.tolist() == [0, -1, -1]
    n, fm = oracle.search_by_projection_map(*margs, uright=np.array([83.0, 82.0, -1.0], np.float32), proj_xr=np.array([80.0], np.float32))
    assert n == 1 and fm.tolist() == [-1, 0, -1]
    n, fm = oracle.search_by_projection_map(*margs, uright=np.array([82.5, 82.0, -1.0], np.float32), proj_xr=np.array([80.0], np.float32))
    assert fm.tolist() == [0, -1, -1]


def test_compute_stereo_matches_known_answers(oracle):
    """Frame::ComputeStereoMatches (src/Frame.cc:869-1048) on a pair whose answer is known: the right image is the left one moved by
    exactly 6 pixels, so every accepted match has its best shift at the keypoint itself (a zero L1 norm, a symmetric parabola unless
    the neighbours differ) and a disparity within a pixel of 6; the median cut keeps the zero norms; no right keypoints -> no match;
    a baseline that forbids the disparity (maxD = mbf / mb < 6) -> no match."""
    left = synth.texture_image(346 + 16, 260, seed=4)
    right = np.ascontiguousarray(left[:, 6:6 + 346]); left = np.ascontiguousarray(left[:, :346])
    eL = oracle.OrbExtractor(1000, 1.2, 8, 20, 7, edgeTh=19, imWidth=346); eR = oracle.OrbExtractor(1000, 1.2, 8, 20, 7, edgeTh=19, imWidth=346)
    _, kL, dL, _ = eL.extract(left, (0, 0)); _, kR, dR, _ = eR.extract(right, (0, 0))
    ur, dp, n = eL.compute_stereo_matches(eR, kL, dL, kR, dR, 0.11, 40.0)
    m = ur > 0
    assert n > 300 and m.sum() > 200
    d = kL["x"][m] - ur[m]
    assert np.all(np.abs(d - 6.0) < 1.0 * eL.scale_factors[kL["octave"][m]] + 1e-3)
    lvl0 = m & (kL["octave"] == 0)
    assert lvl0.sum() > 20 and np.all(np.abs(kL["x"][lvl0] - ur[lvl0] - 6.0) <= 0.5)          # level 0: the pixels are identical, the parabola stays within half a pixel
    assert np.array_equal(dp[m].view(np.uint32), (np.float32(40.0) / (kL["x"][m] - ur[m]).astype(np.float32)).view(np.uint32))
    assert np.all(ur[~m] == -1) and np.all(dp[~m] == -1)
    ur0, dp0, n0 = eL.compute_stereo_matches(eR, kL, dL, kR[:0], dR[:0], 0.11, 40.0)
    assert n0 == 0 and np.all(ur0 == -1)
    ur1, dp1, n1 = eL.compute_stereo_matches(eR, kL, dL, kR, dR, 0.11, 0.5)                    # maxD = 4.5 pixels
    assert (ur1 > 0).sum() < 0.1 * m.sum()
