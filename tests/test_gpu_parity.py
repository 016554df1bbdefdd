"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle, bit-exact.
Run on the GPU box with `pytest -m gpu`."""
import os

import numpy as np
import pytest

from eorb_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fe():
    from eorb_slam_amd import frontend
    return frontend


@pytest.fixture(scope="module")
def ctx(fe):
    c = fe.Context()
    yield c
    c.close()


def _same_bits(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


# ---- event accumulation ---------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [
    dict(n=2000, sigma=1.0, pol=False, frac=True),
    dict(n=2000, sigma=1.0, pol=True, frac=True),
    dict(n=6000, sigma=1.0, pol=False, frac=False),
    dict(n=5000, sigma=0.5, pol=False, frac=True),      # h = 2
    dict(n=5000, sigma=1.5, pol=False, frac=True),      # h = 5: events span up to 3 tiles per axis
    dict(n=3000, sigma=2.5, pol=True, frac=True),       # h = 8
    dict(n=1, sigma=1.0, pol=False, frac=True),
    dict(n=0, sigma=1.0, pol=False, frac=True),
    dict(n=20000, sigma=1.0, pol=False, frac=True, W=346, H=260),   # MVSEC shape, partial last tile
])
def test_ev2im_gauss_bit_exact(oracle, fe, ctx, case):
    """form: the accumulation kernels -- chosen by the call's size (0: up to 16 384 events take the binning-free kernel), the
    binned list pipeline whatever the size (1), the binning-free kernel whatever the size (3)."""
    W, H = case.get("W", 240), case.get("H", 180)
    ev = synth.random_events(case["n"], W, H, seed=11 + case["n"], frac=case["frac"])
    try:
        for form in (0, 1, 3, -1, -4):
            # -1: the bulk form of the float path at test size (the distinct positions of the call tabulated, then the raw kernels);
            # -4: the same with the slot lists (rows computed from the call's positions)
            ctx.debug_option("gather_form", (2 if form == -1 else 4) if form < 0 else form)
            ctx.debug_option("dedupe_min_events", 1 if form < 0 else 1 << 20)
            for normalized in (True, False):
                of, ou, omm = oracle.ev2im_gauss(ev, W, H, case["sigma"], case["pol"], normalized)
                gf, gu, gmm = fe.EvImConverter.ev2im_gauss(ev, W, H, case["sigma"], case["pol"], normalized, ctx=ctx, return_all=True)
                assert _same_bits(of, gf), "form %d: f32 image differs: %d px" % (form, int((of.view(np.uint32) != gf.view(np.uint32)).sum()))
                assert _same_bits(omm, gmm), form
                if normalized:
                    assert np.array_equal(ou, gu), form
    finally:
        ctx.debug_option("gather_form", 0)
        ctx.debug_option("dedupe_min_events", 1 << 20)


def test_float_bulk_form_and_its_fallback(oracle, fe):
    """Float events in bulk: (a) positions that repeat (events a loader resolved through its maps) take the per-call position table and
    the raw kernels; (b) 600 000 distinct positions overflow the table's 2^19 rows and the call falls back to the list pipeline.  Both
    equal the oracle, also with polarity."""
    c = fe.Context()
    c.debug_option("dedupe_min_events", 1)
    W, H = 240, 180
    ev_rep = synth.shapes_events(300000, W, H, seed=41, undistort=True)
    ev_dis = synth.random_events(600000, W, H, seed=42, frac=True)
    for ev, pol in ((ev_rep, False), (ev_rep, True), (ev_dis, False)):
        of, ou, omm = oracle.ev2im_gauss(ev, W, H, 1.0, pol, True)
        gf, gu, gmm = fe.EvImConverter.ev2im_gauss(ev, W, H, 1.0, pol, True, ctx=c, return_all=True)
        assert _same_bits(of, gf) and np.array_equal(ou, gu) and _same_bits(omm, gmm), (len(ev), pol)
    c.close()


def test_float_bulk_position_dictionary(oracle, fe):
    """Float events in bulk across calls: the positions of a call are frozen into the context's dictionary and the next calls look
    theirs up in the count pass (one pass over the events).  Calls 2-3 are served by the dictionary; a call with a new position falls
    back to the per-call tabulation and refreezes; a different sigma rebuilds; positions that never repeat (600 000 distinct) are not
    frozen; with the test hook off nothing is.  Every image against the oracle; also as a batch through the front end."""
    W, H = 240, 180
    c = fe.Context()
    c.debug_option("dedupe_min_events", 1)

    def check(ev, sigma=1.0):
        of, ou, omm = oracle.ev2im_gauss(ev, W, H, sigma, False, True, fast=True)
        gf, gu, gmm = fe.EvImConverter.ev2im_gauss(ev, W, H, sigma, False, True, ctx=c, return_all=True)
        assert _same_bits(of, gf) and np.array_equal(ou, gu) and _same_bits(omm, gmm), (len(ev), sigma)
    evs = [synth.shapes_events(200000 + 1000 * k, W, H, seed=41 + k, undistort=True) for k in range(6)]
    # a sparse call does not see every sensor pixel: the positions accumulate over the calls that miss (each extends the table and
    # refreezes) until the dictionary holds all that fire
    check(evs[0]); assert c.debug_counter("dict_hits") == 0 and 1000 < c.debug_counter("dict_positions") <= 43200
    n1 = c.debug_counter("dict_positions")
    check(evs[5]); assert c.debug_counter("dict_misses") == 1 and n1 < c.debug_counter("dict_positions") <= 43200
    # a call that visits every pixel completes it
    mx, my = synth.undistort_lut(W, H)
    yy, xx = np.mgrid[0:H, 0:W]
    allpix = np.zeros(W * H, synth.RAW_DTYPE); allpix["x"] = xx.ravel(); allpix["y"] = yy.ravel(); allpix["p"] = 1; allpix["t"] = np.arange(W * H) * 1e-6
    cover = np.concatenate([oracle.undistort_events(allpix, mx, my, W, H, True, 1.0), evs[0][:180000]])
    check(cover)
    m0, h0 = c.debug_counter("dict_misses"), c.debug_counter("dict_hits")
    check(evs[1]); check(evs[2])
    assert c.debug_counter("dict_hits") == h0 + 2 and c.debug_counter("dict_misses") == m0
    odd = evs[3].copy(); odd["x"][777] += np.float32(0.125); odd["y"][12345] = np.float32(-2.5)      # two positions no map produces
    check(odd); assert c.debug_counter("dict_misses") == m0 + 1 and c.debug_counter("dict_hits") == h0 + 2
    check(odd); check(evs[4]); assert c.debug_counter("dict_hits") == h0 + 4
    check(evs[0], 0.7); check(evs[1], 0.7); assert c.debug_counter("dict_hits") == h0 + 5      # another sigma: frozen afresh (the positions are kept), then used
    nan = evs[2].copy(); nan["x"][5] = np.nan
    check(nan, 0.7); assert c.debug_counter("dict_hits") == h0 + 6                                 # NaN coordinates are dropped, not looked up
    hits = c.debug_counter("dict_hits")
    check(synth.random_events(600000, W, H, seed=42, frac=True))                                   # never repeats: list pipeline, nothing frozen, nothing looked up
    assert c.debug_counter("dict_hits") == hits
    check(evs[3], 0.7); assert c.debug_counter("dict_hits") == hits + 1                            # (the sigma 0.7 dictionary is still there)
    hits += 1
    c.debug_option("position_dict", 0)
    check(evs[0]); check(evs[1]); assert c.debug_counter("dict_hits") == hits and c.debug_counter("dict_positions") == 0
    c.close()
    # the batched front end: the second batch of a context goes through the dictionary and equals the raw path's result
    mx, my = _maps(W, H)
    B, n = 3, 400000
    pairs = [synth.shapes_events(n, W, H, seed=90 + b, motion=0.4, undistort=True, return_raw=True) for b in range(2 * B)]
    fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=B, max_events=n)
    cc, cap = fb.ctx, fb.cap
    d_ev = cc.dev_alloc(B * n * 16); d_img = cc.dev_alloc(B * W * H)
    for half in range(2):
        blob = np.concatenate([fe.pack_events(p[0]) for p in pairs[half * B:(half + 1) * B]])
        cc.upload(d_ev, blob)
        fb.run_dev(d_ev, np.arange(B + 1, dtype=np.int64) * n, d_img, None, None, None, None, None, raw=False)
        cc.sync()
        imgs = np.zeros((B, H, W), np.uint8); cc.download(imgs, d_img)
        for b in range(B):
            assert np.array_equal(imgs[b], oracle.ev2im_gauss(pairs[half * B + b][0], W, H, 1.0, False, True, fast=True)[1]), (half, b)
    assert cc.debug_counter("dict_hits") + cc.debug_counter("dict_misses") == 1            # (1.2 M events: the second batch may still meet a new pixel)
    cc.dev_free(d_ev); cc.dev_free(d_img); cc.close()


def test_ev2im_gauss_shapes_lut(oracle, fe, ctx):
    """C1 stand-in: LUT-undistorted shapes events, L1 chunk (2000) and L2 window (6000)."""
    for n in (2000, 6000, 50000):
        ev = synth.shapes_events(n, seed=1, undistort=True)
        of, ou, omm = oracle.ev2im_gauss(ev, 240, 180, 1.0, False, True)
        gf, gu, gmm = fe.EvImConverter.ev2im_gauss(ev, 240, 180, 1.0, False, True, ctx=ctx, return_all=True)
        assert _same_bits(of, gf) and np.array_equal(ou, gu) and _same_bits(omm, gmm)


def test_ev2im_gauss_hot_pixel_order(oracle, fe, ctx):
    """Many events on few pixels (long per-pixel chains crossing many chunks): order must be preserved."""
    rng = np.random.default_rng(5)
    n = 30000
    ev = synth.random_events(n, seed=5)
    ev["x"] = (100 + rng.uniform(0, 3, n)).astype(np.float32); ev["y"] = (60 + rng.uniform(0, 3, n)).astype(np.float32)
    try:
        for form in (0, 3):                          # 30 000 events: the binned pipeline, then the binning-free kernel (its list is flushed)
            ctx.debug_option("gather_form", form)
            for pol in (False, True):
                of, ou, omm = oracle.ev2im_gauss(ev, 240, 180, 1.0, pol, True)
                gf, gu, gmm = fe.EvImConverter.ev2im_gauss(ev, 240, 180, 1.0, pol, True, ctx=ctx, return_all=True)
                assert _same_bits(of, gf) and np.array_equal(ou, gu) and _same_bits(omm, gmm), form
    finally:
        ctx.debug_option("gather_form", 0)


@pytest.mark.parametrize("pol", [False, True])
def test_ev2im_count_bit_exact(oracle, fe, ctx, pol):
    ev = synth.random_events(20000, seed=3, frac=True)
    for normalized in (True, False):
        of, ou, omm = oracle.ev2im(ev, 240, 180, pol, normalized)
        gf, gu, gmm = fe.EvImConverter.ev2im(ev, 240, 180, pol, normalized, ctx=ctx, return_all=True)
        assert _same_bits(of, gf) and _same_bits(omm, gmm)
        assert (ou is None) == (gu is None)
        if ou is not None:
            assert np.array_equal(ou, gu)


# ---- ORB extractor -------------------------------------------------------------------------------------------
def _event_image(oracle, n=60000, seed=2):
    ev = synth.shapes_events(n, seed=seed)
    return oracle.ev2im_gauss(ev, 240, 180, 1.0, False, True)[1]


def _check_extract(oracle, fe, img, lap=(0, 1000), want_desc=True, **p):
    H, W = img.shape
    oe = oracle.OrbExtractor(imWidth=W, **p)
    ge = fe.ORBextractor(imSize=(W, H), **p)
    omono, okp, odesc, ooob = oe.extract(img, lap, want_desc)
    gmono, gkp, gdesc, goob = ge(img, lap, want_desc)
    assert list(ge.mnFeaturesPerLevel) == oe.features_per_level and ge.edge == oe.edge
    assert np.array_equal(ge.mvScaleFactor.view(np.uint32), oe.scale_factors.view(np.uint32))
    assert omono == gmono
    assert len(okp) == len(gkp), "keypoint count %d vs %d" % (len(okp), len(gkp))
    for f in ("x", "y", "size", "angle", "response"):
        assert np.array_equal(okp[f].view(np.uint32), gkp[f].view(np.uint32)), f
    assert np.array_equal(okp["octave"], gkp["octave"]) and np.array_equal(okp["class_id"], gkp["class_id"])
    if want_desc:
        assert np.array_equal(odesc, gdesc)
        assert np.array_equal(ooob, goob)
    if lap == (0, 1000):
        # (0, 1000) holds every keypoint and (0, 0) none: both go through ONE launch for orientation, descriptors and output order
        # (describe_kernel); the three kernels an area inside the image needs (debug option) must agree with it and the oracle
        assert omono == 0
        o0 = oe.extract(img, (0, 0), want_desc)
        g0 = ge(img, (0, 0), want_desc)
        ge.ctx.debug_option("orb_three_launches", 1)
        g3 = ge(img, (0, 0), want_desc)
        g3s = ge(img, lap, want_desc)
        ge.ctx.debug_option("orb_three_launches", 0)
        for a, b in ((o0, g0), (o0, g3), ((omono, okp, odesc, ooob), g3s)):
            assert a[0] == b[0] and np.array_equal(a[1].view(np.uint8), b[1].view(np.uint8))
            if want_desc:
                assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
        assert o0[0] == len(okp)                        # every keypoint is "mono" now
    ge.ctx.close()
    return okp, odesc


@pytest.mark.parametrize("edge", [19, 21, 9])
def test_orb_extract_texture(oracle, fe, edge):
    img = synth.texture_image(240, 180, seed=3)
    kp, _ = _check_extract(oracle, fe, img, nfeatures=1000, scaleFactor=1.2, nlevels=4, iniThFAST=10, minThFAST=0, edgeTh=edge)
    assert len(kp) > 300


def test_orb_extract_event_image(oracle, fe):
    img = _event_image(oracle)
    _check_extract(oracle, fe, img, nfeatures=1000, scaleFactor=1.2, nlevels=4, iniThFAST=10, minThFAST=0, edgeTh=19)


def test_orb_extract_detect_only_fast_mode(oracle, fe):
    """Event path: 'FAST = ORB with 1 level', thresholds 0/0, edge 9, 400 and 800 points."""
    img = _event_image(oracle, n=6000, seed=7)
    for nf in (400, 800):
        _check_extract(oracle, fe, img, want_desc=False, nfeatures=nf, scaleFactor=1.0, nlevels=1, iniThFAST=0, minThFAST=0, edgeTh=9)


def test_orb_extract_mvsec_shape(oracle, fe):
    img = synth.texture_image(346, 260, seed=4)
    _check_extract(oracle, fe, img, nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=10, minThFAST=1, edgeTh=15)


def test_orb_extract_euroc_frame_size(oracle, fe):
    """The reference's default extractor width (DEF_IMAGE_WIDTH 752, include/ORBextractor.h:31) and EuRoC frames (752x480,
    Examples/Event/EuRoC.yaml:86): 8 levels, FAST 20 / 7, 1 000 features, and the monocular initialiser's 5 x nFeatures
    (src/Tracking.cc:119-122); the default edge threshold (edgeTh < 0: 19 * (imWidth / 752), ORBextractor.cc:481-488)."""
    img = synth.texture_image(752, 480, seed=12)
    kp, _ = _check_extract(oracle, fe, img, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, edgeTh=-1)
    assert len(kp) > 800
    _check_extract(oracle, fe, img, nfeatures=5000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, edgeTh=19)
    ev = synth.shapes_events(400000, 752, 480, seed=5)                 # an event image of that size: low thresholds, many candidates
    img2 = oracle.ev2im_gauss(ev, 752, 480, 1.0, False, True, fast=True)[1]
    _check_extract(oracle, fe, img2, nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=10, minThFAST=0, edgeTh=19)


def test_orb_extract_stereo_lapping_and_flat(oracle, fe):
    img = synth.texture_image(240, 180, seed=8)
    _check_extract(oracle, fe, img, lap=(0, 0), nfeatures=500, scaleFactor=1.2, nlevels=3, iniThFAST=20, minThFAST=7, edgeTh=19)
    _check_extract(oracle, fe, img, lap=(60, 120), nfeatures=500, scaleFactor=1.2, nlevels=3, iniThFAST=20, minThFAST=7, edgeTh=19)
    flat = np.full((180, 240), 128, np.uint8)           # no corners at all: zero keypoints, descriptors released
    ge = fe.ORBextractor(imSize=(240, 180))
    mono, kp, desc, oob = ge(flat)
    assert mono == 0 and len(kp) == 0
    assert ge(np.zeros((0, 0), np.uint8))[0] == -1      # empty image -> -1 (ORBextractor.cc:1096)
    ge.ctx.close()


def test_orb_extract_bad_config_is_error_not_crash(fe):
    with pytest.raises(fe.EorbError) as ei:
        fe.ORBextractor(nfeatures=1000, scaleFactor=1.2, nlevels=8, edgeTh=19, imSize=(240, 180))   # level 7 < one 30-px cell
    assert ei.value.code == -2


# ---- matchers ----------------------------------------------------------------------------------------------------
def test_bf_knn2(oracle, fe, ctx):
    bf = fe.BFMatcher(ctx)
    t = synth.random_descriptors(2000, seed=4)
    for q in (synth.random_descriptors(2000, seed=14), synth.planted_descriptors(t, seed=5)[0], t[:37]):
        oi, od = oracle.bf_knn2(q, t)
        gi, gd = bf.knnMatch2(q, t)
        assert np.array_equal(oi, gi) and np.array_equal(od, gd)
    # heavy ties: many identical train rows -> lowest train index wins
    t2 = np.repeat(synth.random_descriptors(8, seed=1), 50, axis=0)
    oi, od = oracle.bf_knn2(t2[::7], t2)
    gi, gd = bf.knnMatch2(t2[::7], t2)
    assert np.array_equal(oi, gi) and np.array_equal(od, gd)
    # a single train row: second neighbour absent
    oi, od = oracle.bf_knn2(t[:5], t[:1])
    gi, gd = bf.knnMatch2(t[:5], t[:1])
    assert np.array_equal(oi, gi) and np.array_equal(od, gd)


def _two_frames(oracle, seed=3, shift=3):
    img1 = synth.texture_image(240, 180, seed=seed)
    img2 = np.roll(img1, (shift, -shift), axis=(0, 1))
    e = oracle.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=19)
    _, k1, d1, _ = e.extract(img1)
    _, k2, d2, _ = e.extract(img2)
    return k1, d1, k2, d2


@pytest.mark.parametrize("window,ratio,ori", [(100, 0.9, True), (30, 0.6, True), (100, 0.9, False)])
def test_search_for_initialization(oracle, fe, ctx, window, ratio, ori):
    k1, d1, k2, d2 = _two_frames(oracle)
    pm = np.stack([k1["x"], k1["y"]], axis=1)
    on, om, opm = oracle.search_for_initialization(oracle.Frame(k1, d1, 240, 180), oracle.Frame(k2, d2, 240, 180), pm, window, ratio, ori)
    gn, gm, gpm = fe.ORBmatcher(ratio, ori, ctx).SearchForInitialization(fe.FrameView(k1, d1, 240, 180), fe.FrameView(k2, d2, 240, 180), pm, window)
    assert on == gn and np.array_equal(om, gm) and np.array_equal(opm.view(np.uint32), gpm.view(np.uint32))
    assert on > 20


def test_search_for_initialization_mixed_gate(oracle, fe, ctx):
    """MixedMatcher: ORB rows + 61-byte AKAZE-like rows; only same-type pairs, first 32 bytes compared."""
    k1, d1, k2, d2 = _two_frames(oracle, seed=13)
    rng = np.random.default_rng(1)

    def mix(k, d, seed):
        n = len(k)
        d61 = np.zeros((n, 61), np.uint8); d61[:, :32] = d; d61[:, 32:] = rng.integers(0, 256, (n, 29))
        is_orb = (np.arange(n) < n * 2 // 3).astype(np.uint8)
        k = k.copy(); k["class_id"] = np.where(is_orb == 1, -1, 0); k["octave"] = np.where(is_orb == 1, k["octave"], 2)
        return k, d61, is_orb
    k1, d1, o1 = mix(k1, d1, 1); k2, d2, o2 = mix(k2, d2, 2)
    pm = np.stack([k1["x"], k1["y"]], axis=1)
    on, om, opm = oracle.search_for_initialization(oracle.Frame(k1, d1, 240, 180, o1), oracle.Frame(k2, d2, 240, 180, o2), pm, 100, 0.9, True)
    gn, gm, gpm = fe.ORBmatcher(0.9, True, ctx).SearchForInitialization(fe.FrameView(k1, d1, 240, 180, o1), fe.FrameView(k2, d2, 240, 180, o2), pm, 100)
    assert on == gn and np.array_equal(om, gm) and np.array_equal(opm.view(np.uint32), gpm.view(np.uint32))


def test_search_by_projection_last(oracle, fe, ctx):
    k1, d1, k2, d2 = _two_frames(oracle, seed=21, shift=2)
    rng = np.random.default_rng(2)
    n1 = len(k1)
    valid = (rng.uniform(size=n1) < 0.8).astype(np.uint8)
    uv = np.stack([k1["x"] - 2 + rng.normal(0, 1, n1), k1["y"] + 2 + rng.normal(0, 1, n1)], axis=1).astype(np.float32)
    mp_obs = (rng.uniform(size=n1) < 0.7).astype(np.uint8)
    sf = oracle.OrbExtractor(1000, 1.2, 4).scale_factors
    ls = sf[np.clip(k1["octave"], 0, 3)]
    cur_mp = np.full(len(k2), -1, np.int32); cur_mp[::17] = -2; cur_mp[5::23] = -3
    for mode in (0, 1, 2):
        for ori in (True, False):
            on, ocm = oracle.search_by_projection_last(oracle.Frame(k2, d2, 240, 180), oracle.Frame(k1, d1, 240, 180), valid, uv, d1, mp_obs, cur_mp, 15.0, ls, mode, ori)
            gn, gcm = fe.ORBmatcher(0.9, ori, ctx).SearchByProjectionLast(fe.FrameView(k2, d2, 240, 180), fe.FrameView(k1, d1, 240, 180), valid, uv, d1, mp_obs, cur_mp, 15.0, ls, mode)
            assert on == gn and np.array_equal(ocm, gcm)
    assert on > 10


def test_search_by_projection_map(oracle, fe, ctx):
    k1, d1, k2, d2 = _two_frames(oracle, seed=22, shift=1)
    rng = np.random.default_rng(3)
    M = len(k1)
    in_view = (rng.uniform(size=M) < 0.9).astype(np.uint8)
    proj = np.stack([k1["x"] - 1 + rng.normal(0, 0.7, M), k1["y"] + 1 + rng.normal(0, 0.7, M)], axis=1).astype(np.float32)
    level = k1["octave"].astype(np.int32)
    vc = rng.uniform(0.99, 1.0, M).astype(np.float32)
    mp_obs = (rng.uniform(size=M) < 0.6).astype(np.uint8)
    sf = oracle.OrbExtractor(1000, 1.2, 4).scale_factors
    ls = sf[np.clip(level, 0, 3)]
    fm = np.full(len(k2), -1, np.int32); fm[::19] = -2
    for th in (1.0, 3.0):
        on, ofm = oracle.search_by_projection_map(oracle.Frame(k2, d2, 240, 180), in_view, proj, level, vc, d1, mp_obs, fm, th, 0.8, ls)
        gn, gfm = fe.ORBmatcher(0.8, True, ctx).SearchByProjectionMap(fe.FrameView(k2, d2, 240, 180), in_view, proj, level, vc, d1, mp_obs, fm, th, ls)
        assert on == gn and np.array_equal(ofm, gfm)
    assert on > 10


def test_search_by_projection_stereo_gate(oracle, fe, ctx):
    """The rectified-stereo gate of the two projection matchers (src/ORBmatcher.cc:96-104, :2056-2062): a candidate with a right
    coordinate (mvuRight > 0) must also lie within the radius of the query's predicted right coordinate.  Keypoints without a right
    match (<= 0) are not gated; with none at all the result is the mono matcher's."""
    k1, d1, k2, d2 = _two_frames(oracle, seed=23, shift=2)
    rng = np.random.default_rng(4)
    n1, n2 = len(k1), len(k2)
    sf = oracle.OrbExtractor(1000, 1.2, 4).scale_factors
    ls = sf[np.clip(k1["octave"], 0, 3)]
    valid = (rng.uniform(size=n1) < 0.85).astype(np.uint8)
    uv = np.stack([k1["x"] - 2 + rng.normal(0, 1, n1), k1["y"] + 2 + rng.normal(0, 1, n1)], axis=1).astype(np.float32)
    mp_obs = (rng.uniform(size=n1) < 0.7).astype(np.uint8)
    cur_mp = np.full(n2, -1, np.int32); cur_mp[::17] = -2
    disp = rng.uniform(2, 30, n2).astype(np.float32)
    ur2 = (k2["x"] - disp).astype(np.float32)
    ur2[rng.uniform(size=n2) < 0.3] = -1.0                        # no right match
    # the queries' predicted right coordinate: near the true one for most, far off for some (those candidates drop out)
    q_ur = (uv[:, 0] - rng.uniform(2, 30, n1)).astype(np.float32)
    F2, F1 = oracle.Frame(k2, d2, 240, 180), oracle.Frame(k1, d1, 240, 180)
    G2, G1 = fe.FrameView(k2, d2, 240, 180), fe.FrameView(k1, d1, 240, 180)
    seen = []
    for mode in (0, 1, 2):
        for th in (7.0, 15.0):
            on, ocm = oracle.search_by_projection_last(F2, F1, valid, uv, d1, mp_obs, cur_mp, th, ls, mode, True, uright=ur2, proj_ur=q_ur)
            gn, gcm = fe.ORBmatcher(0.9, True, ctx).SearchByProjectionLast(G2, G1, valid, uv, d1, mp_obs, cur_mp, th, ls, mode, uright=ur2, proj_ur=q_ur)
            assert on == gn and np.array_equal(ocm, gcm)
            mn, mcm = fe.ORBmatcher(0.9, True, ctx).SearchByProjectionLast(G2, G1, valid, uv, d1, mp_obs, cur_mp, th, ls, mode)
            seen.append((on, mn, not np.array_equal(gcm, mcm)))
    assert any(d for _, _, d in seen) and all(a > 5 for a, _, _ in seen)          # the gate changed the result, and matches remain
    none = np.full(n2, -1.0, np.float32)
    gn, gcm = fe.ORBmatcher(0.9, True, ctx).SearchByProjectionLast(G2, G1, valid, uv, d1, mp_obs, cur_mp, 15.0, ls, 0, uright=none, proj_ur=q_ur)
    mn, mcm = fe.ORBmatcher(0.9, True, ctx).SearchByProjectionLast(G2, G1, valid, uv, d1, mp_obs, cur_mp, 15.0, ls, 0)
    assert gn == mn and np.array_equal(gcm, mcm)
    # the map form: mTrackProjXR per map point
    M = n1
    in_view = (rng.uniform(size=M) < 0.9).astype(np.uint8)
    proj = np.stack([k1["x"] - 2 + rng.normal(0, 0.7, M), k1["y"] + 2 + rng.normal(0, 0.7, M)], axis=1).astype(np.float32)
    level = k1["octave"].astype(np.int32)
    vc = rng.uniform(0.99, 1.0, M).astype(np.float32)
    lsm = sf[np.clip(level, 0, 3)]
    fm = np.full(n2, -1, np.int32); fm[::19] = -2
    xr = (proj[:, 0] - rng.uniform(2, 30, M)).astype(np.float32)
    changed = False
    for th in (1.0, 3.0):
        on, ofm = oracle.search_by_projection_map(F2, in_view, proj, level, vc, d1, mp_obs, fm, th, 0.8, lsm, uright=ur2, proj_xr=xr)
        gn, gfm = fe.ORBmatcher(0.8, True, ctx).SearchByProjectionMap(G2, in_view, proj, level, vc, d1, mp_obs, fm, th, lsm, uright=ur2, proj_xr=xr)
        assert on == gn and np.array_equal(ofm, gfm)
        mn, mfm = fe.ORBmatcher(0.8, True, ctx).SearchByProjectionMap(G2, in_view, proj, level, vc, d1, mp_obs, fm, th, lsm)
        changed |= not np.array_equal(gfm, mfm)
    assert changed and on > 5
    gn, gfm = fe.ORBmatcher(0.8, True, ctx).SearchByProjectionMap(G2, in_view, proj, level, vc, d1, mp_obs, fm, 3.0, lsm, uright=none, proj_xr=xr)
    mn, mfm = fe.ORBmatcher(0.8, True, ctx).SearchByProjectionMap(G2, in_view, proj, level, vc, d1, mp_obs, fm, 3.0, lsm)
    assert gn == mn and np.array_equal(gfm, mfm)


def _mix(k, d, rng, frac=(2, 3)):
    """A MixedFrame stand-in: the first 2/3 of the rows are ORB, the rest "AKAZE" (61-byte rows whose first 32 bytes are
    compared, octaves of their own scale ladder)."""
    n = len(k)
    d61 = np.zeros((n, 61), np.uint8); d61[:, :32] = d; d61[:, 32:] = rng.integers(0, 256, (n, 29))
    is_orb = (np.arange(n) < n * frac[0] // frac[1]).astype(np.uint8)
    k = k.copy(); k["class_id"] = np.where(is_orb == 1, -1, 0); k["octave"] = np.where(is_orb == 1, k["octave"], k["octave"] % 3)
    return k, d61, is_orb


AKAZE_SF = np.array([1.0, 1.2599211, 1.5874010, 2.0], np.float32)      # getAKAZEScaleFactor: 2^(level / nOctaveLayers)


def test_search_by_projection_last_mixed_gate(oracle, fe, ctx):
    """MixedMatcher::SearchByProjection(cur, last) (src/MixedMatcher.cpp:693-926): only same-type pairs are compared and the window
    of an AKAZE map point scales with the AKAZE ladder."""
    k1, d1, k2, d2 = _two_frames(oracle, seed=23, shift=2)
    rng = np.random.default_rng(4)
    k1, d1m, o1 = _mix(k1, d1, rng); k2, d2m, o2 = _mix(k2, d2, rng, frac=(3, 5))
    n1 = len(k1)
    valid = (rng.uniform(size=n1) < 0.85).astype(np.uint8)
    uv = np.stack([k1["x"] - 2 + rng.normal(0, 1, n1), k1["y"] + 2 + rng.normal(0, 1, n1)], axis=1).astype(np.float32)
    mp_obs = (rng.uniform(size=n1) < 0.7).astype(np.uint8)
    sf = oracle.OrbExtractor(1000, 1.2, 4).scale_factors
    ls = np.where(o1 == 1, sf[np.clip(k1["octave"], 0, 3)], AKAZE_SF[np.clip(k1["octave"], 0, 3)]).astype(np.float32)
    cur_mp = np.full(len(k2), -1, np.int32); cur_mp[::17] = -2; cur_mp[5::23] = -3
    mp_desc = np.ascontiguousarray(d1m[:, :32])
    totals = []
    for mode in (0, 1, 2):
        for ori in (True, False):
            on, ocm = oracle.search_by_projection_last(oracle.Frame(k2, d2m, 240, 180, o2), oracle.Frame(k1, d1m, 240, 180, o1), valid, uv,
                                                       mp_desc, mp_obs, cur_mp, 15.0, ls, mode, ori)
            gn, gcm = fe.ORBmatcher(0.9, ori, ctx).SearchByProjectionLast(fe.FrameView(k2, d2m, 240, 180, o2), fe.FrameView(k1, d1m, 240, 180, o1),
                                                                          valid, uv, mp_desc, mp_obs, cur_mp, 15.0, ls, mode)
            assert on == gn and np.array_equal(ocm, gcm)
            m = ocm >= 0
            assert np.array_equal(o2[m], o1[ocm[m]]), "a pair of different feature types was matched"
            totals.append(on)
    # the gate changes the result: without the type flags more (cross-type) pairs are found
    on_plain, _ = oracle.search_by_projection_last(oracle.Frame(k2, d2m, 240, 180), oracle.Frame(k1, d1m, 240, 180), valid, uv, mp_desc,
                                                   mp_obs, cur_mp, 15.0, ls, 0, True)
    assert totals[0] > 10 and on_plain != totals[0]


def test_search_by_projection_map_mixed_gate(oracle, fe, ctx):
    """MixedMatcher::SearchByProjection(F, vpMapPoints) (src/MixedMatcher.cpp:500-691): isORBMapPoint gate + AKAZE window scale."""
    k1, d1, k2, d2 = _two_frames(oracle, seed=24, shift=1)
    rng = np.random.default_rng(5)
    k2, d2m, o2 = _mix(k2, d2, rng)
    M = len(k1)
    mp_is_orb = (rng.uniform(size=M) < 0.6).astype(np.uint8)
    in_view = (rng.uniform(size=M) < 0.9).astype(np.uint8)
    proj = np.stack([k1["x"] - 1 + rng.normal(0, 0.7, M), k1["y"] + 1 + rng.normal(0, 0.7, M)], axis=1).astype(np.float32)
    level = np.where(mp_is_orb == 1, k1["octave"], k1["octave"] % 3).astype(np.int32)
    vc = rng.uniform(0.99, 1.0, M).astype(np.float32)
    mp_obs = (rng.uniform(size=M) < 0.6).astype(np.uint8)
    sf = oracle.OrbExtractor(1000, 1.2, 4).scale_factors
    ls = np.where(mp_is_orb == 1, sf[np.clip(level, 0, 3)], AKAZE_SF[np.clip(level, 0, 3)]).astype(np.float32)
    fm = np.full(len(k2), -1, np.int32); fm[::19] = -2
    for th in (1.0, 3.0):
        on, ofm = oracle.search_by_projection_map(oracle.Frame(k2, d2m, 240, 180, o2), in_view, proj, level, vc, d1, mp_obs, fm, th, 0.8, ls,
                                                  mp_is_orb=mp_is_orb)
        gn, gfm = fe.ORBmatcher(0.8, True, ctx).SearchByProjectionMap(fe.FrameView(k2, d2m, 240, 180, o2), in_view, proj, level, vc, d1,
                                                                      mp_obs, fm, th, ls, mp_is_orb=mp_is_orb)
        assert on == gn and np.array_equal(ofm, gfm)
        m = ofm >= 0
        assert np.array_equal(o2[m], mp_is_orb[ofm[m]]), "a map point was matched to a feature of the other type"
    on_plain, _ = oracle.search_by_projection_map(oracle.Frame(k2, d2m, 240, 180), in_view, proj, level, vc, d1, mp_obs, fm, 3.0, 0.8, ls)
    assert on > 10 and on_plain != on


@pytest.mark.parametrize("wcap,ecap,lds", [(2, 0, 0), (0, 40, 0), (1, 3, 0), (0, 0, 1)])
def test_window_matchers_full_scan_path(oracle, fe, wcap, ecap, lds):
    """The two-phase window matchers keep a short candidate list per query; a query whose list (or whose frame pair's pool) overflows
    is resolved by a full scan inside the sequential phase.  Tiny capacities force that path for all three matchers; the last case keeps
    the lists but leaves most of them in global memory (phase 2 stages as many entries in LDS as fit: all of them at these sizes)."""
    c = fe.Context()
    if lds:
        c.debug_option("win_lds_entries", lds)
    if wcap:
        c.debug_option("win_list_cap", wcap)
    if ecap:
        c.debug_option("win_pool_cap", ecap)
    k1, d1, k2, d2 = _two_frames(oracle, seed=27, shift=2)
    rng = np.random.default_rng(6)
    k1m, d1m, o1 = _mix(k1, d1, rng); k2m, d2m, o2 = _mix(k2, d2, rng, frac=(3, 5))
    pm = np.stack([k1m["x"], k1m["y"]], axis=1)
    on, om, opm = oracle.search_for_initialization(oracle.Frame(k1m, d1m, 240, 180, o1), oracle.Frame(k2m, d2m, 240, 180, o2), pm, 100, 0.9, True)
    gn, gm, gpm = fe.ORBmatcher(0.9, True, c).SearchForInitialization(fe.FrameView(k1m, d1m, 240, 180, o1), fe.FrameView(k2m, d2m, 240, 180, o2), pm, 100)
    assert on == gn and np.array_equal(om, gm) and np.array_equal(opm.view(np.uint32), gpm.view(np.uint32)) and on > 20
    n1 = len(k1)
    valid = (rng.uniform(size=n1) < 0.85).astype(np.uint8)
    uv = np.stack([k1["x"] - 2 + rng.normal(0, 1, n1), k1["y"] + 2 + rng.normal(0, 1, n1)], axis=1).astype(np.float32)
    mp_obs = (rng.uniform(size=n1) < 0.7).astype(np.uint8)
    sf = oracle.OrbExtractor(1000, 1.2, 4).scale_factors
    ls = sf[np.clip(k1["octave"], 0, 3)]
    cur_mp = np.full(len(k2), -1, np.int32); cur_mp[::17] = -2; cur_mp[5::23] = -3
    for mode in (0, 1, 2):
        on, ocm = oracle.search_by_projection_last(oracle.Frame(k2, d2, 240, 180), oracle.Frame(k1, d1, 240, 180), valid, uv, d1, mp_obs, cur_mp, 15.0, ls, mode, True)
        gn, gcm = fe.ORBmatcher(0.9, True, c).SearchByProjectionLast(fe.FrameView(k2, d2, 240, 180), fe.FrameView(k1, d1, 240, 180), valid, uv, d1, mp_obs, cur_mp, 15.0, ls, mode)
        assert on == gn and np.array_equal(ocm, gcm) and on > 10
    level = k1["octave"].astype(np.int32)
    vc = rng.uniform(0.99, 1.0, n1).astype(np.float32)
    fm = np.full(len(k2), -1, np.int32); fm[::19] = -2
    for th in (1.0, 3.0):
        on, ofm = oracle.search_by_projection_map(oracle.Frame(k2, d2, 240, 180), valid, uv, level, vc, d1, mp_obs, fm, th, 0.8, ls)
        gn, gfm = fe.ORBmatcher(0.8, True, c).SearchByProjectionMap(fe.FrameView(k2, d2, 240, 180), valid, uv, level, vc, d1, mp_obs, fm, th, ls)
        assert on == gn and np.array_equal(ofm, gfm)
    c.close()


def test_window_matchers_state_chains(oracle, fe, ctx):
    """Repeated structure: many near-identical descriptors inside one window, so that queries steal each other's matches
    (vnMatches21, :774-781) and skip candidates already matched at a smaller distance (:755) -- the state the sequential phase carries."""
    rng = np.random.default_rng(12)
    n = 600
    base = synth.random_descriptors(6, seed=3)
    kp = synth.random_keypoints(n, 240, 180, nlevels=1, seed=8); kp["octave"] = 0
    kp["x"] = (100 + rng.uniform(0, 60, n)).astype(np.float32); kp["y"] = (60 + rng.uniform(0, 50, n)).astype(np.float32)

    def noisy(seed):
        r = np.random.default_rng(seed)
        d = base[r.integers(0, len(base), n)].copy()
        for i in range(n):
            for b in r.choice(256, size=int(r.integers(0, 9)), replace=False):
                d[i, b >> 3] ^= np.uint8(1 << (b & 7))
        return d
    d1, d2 = noisy(1), noisy(2)
    k2 = kp.copy(); k2["x"] += rng.normal(0, 2, n).astype(np.float32); k2["y"] += rng.normal(0, 2, n).astype(np.float32)
    k2["angle"] = ((kp["angle"] + rng.normal(0, 4, n)) % 360).astype(np.float32)
    pm = np.stack([kp["x"], kp["y"]], axis=1)
    for ratio in (0.9, 1.0, 0.5):
        on, om, _ = oracle.search_for_initialization(oracle.Frame(kp, d1, 240, 180), oracle.Frame(k2, d2, 240, 180), pm, 100, ratio, True)
        gn, gm, _ = fe.ORBmatcher(ratio, True, ctx).SearchForInitialization(fe.FrameView(kp, d1, 240, 180), fe.FrameView(k2, d2, 240, 180), pm, 100)
        assert on == gn and np.array_equal(om, gm)
    valid = np.ones(n, np.uint8); uv = np.stack([kp["x"], kp["y"]], axis=1).astype(np.float32)
    mp_obs = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    ls = np.ones(n, np.float32)
    cur_mp = np.full(n, -1, np.int32)
    on, ocm = oracle.search_by_projection_last(oracle.Frame(k2, d2, 240, 180), oracle.Frame(kp, d1, 240, 180), valid, uv, d1, mp_obs, cur_mp, 15.0, ls, 0, True)
    gn, gcm = fe.ORBmatcher(0.9, True, ctx).SearchByProjectionLast(fe.FrameView(k2, d2, 240, 180), fe.FrameView(kp, d1, 240, 180), valid, uv, d1, mp_obs, cur_mp, 15.0, ls, 0)
    assert on == gn and np.array_equal(ocm, gcm) and on > 50
    lvl = np.zeros(n, np.int32); vc = np.full(n, 0.999, np.float32)
    for ratio in (0.8, 1.0):
        on, ofm = oracle.search_by_projection_map(oracle.Frame(k2, d2, 240, 180), valid, uv, lvl, vc, d1, mp_obs, cur_mp, 3.0, ratio, ls)
        gn, gfm = fe.ORBmatcher(ratio, True, ctx).SearchByProjectionMap(fe.FrameView(k2, d2, 240, 180), valid, uv, lvl, vc, d1, mp_obs, cur_mp, 3.0, ls)
        assert on == gn and np.array_equal(ofm, gfm)


# ---- the batched HBM-resident pipeline -------------------------------------------------------------------------------
def test_frontend_batch_matches_oracle_pipeline(oracle, fe):
    W, H, B, n = 240, 180, 3, 40000
    slices = [synth.shapes_events(n, seed=40 + b, motion=0.3) for b in range(B)]
    fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=B, max_events=n, windowSize=100, nnratio=0.9)
    c = fb.ctx
    ev16 = np.concatenate([fe.pack_events(s) for s in slices])
    cap = fb.cap
    d_ev = c.dev_alloc(ev16.nbytes); c.upload(d_ev, ev16)
    d_img = c.dev_alloc(B * W * H); d_kp = c.dev_alloc(B * cap * 28); d_desc = c.dev_alloc(B * cap * 32)
    d_n = c.dev_alloc(B * 4); d_m = c.dev_alloc(B * cap * 4); d_nm = c.dev_alloc(B * 4)
    offs = np.arange(B + 1, dtype=np.int64) * n
    fb.run_dev(d_ev, offs, d_img, d_kp, d_desc, d_n, d_m, d_nm)
    c.sync()
    imgs = np.zeros((B, H, W), np.uint8); c.download(imgs, d_img)
    kps = np.zeros((B, cap), synth.KP_DTYPE); c.download(kps, d_kp)
    desc = np.zeros((B, cap, 32), np.uint8); c.download(desc, d_desc)
    nk = np.zeros(B, np.int32); c.download(nk, d_n)
    m12 = np.zeros((B, cap), np.int32); c.download(m12, d_m)
    nm = np.zeros(B, np.int32); c.download(nm, d_nm)
    oe = oracle.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=19)
    prev = None
    for b in range(B):
        _, ou, _ = oracle.ev2im_gauss(slices[b], W, H, 1.0, False, True)
        assert np.array_equal(ou, imgs[b])
        _, okp, odesc, _ = oe.extract(ou)
        assert nk[b] == len(okp)
        assert np.array_equal(okp.view(np.uint8), kps[b, :nk[b]].view(np.uint8)) and np.array_equal(odesc, desc[b, :nk[b]])
        if prev is not None:
            pk, pd = prev
            pm = np.stack([pk["x"], pk["y"]], axis=1)
            on, om, _ = oracle.search_for_initialization(oracle.Frame(pk, pd, W, H), oracle.Frame(okp, odesc, W, H), pm, 100, 0.9, True)
            assert on == nm[b] and np.array_equal(om, m12[b, :len(pk)])
        prev = (okp, odesc)
    for p in (d_ev, d_img, d_kp, d_desc, d_n, d_m, d_nm):
        c.dev_free(p)
    c.close()


# ---- tracked keypoints (a14), SearchByBoW (a20), MixedFrame container ops (a23) ----------------------------------------
def test_tracked_descriptors_and_level_assignment(oracle, fe):
    img = synth.texture_image(240, 180, seed=31)
    oe = oracle.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=19)
    ge = fe.ORBextractor(1000, 1.2, 4, 10, 0, 19, (240, 180))
    _, kps, desc, _ = oe.extract(img)
    rng = np.random.default_rng(7)
    # tracked points: extractor keypoints moved by a KLT-like sub-pixel displacement, some with bogus octaves / near borders
    tk = kps.copy()
    tk["x"] += rng.normal(0, 1.5, len(tk)).astype(np.float32); tk["y"] += rng.normal(0, 1.5, len(tk)).astype(np.float32)
    tk["octave"][::37] = 9; tk["octave"][5::41] = -1
    tk["x"][::53] = 3.5; tk["y"][7::59] = 177.25
    od, oo = oe.tracked_descriptors(img, tk)
    gd, go = ge.ComputeTrackedKPtsDesc(img, tk)
    assert np.array_equal(od, gd) and np.array_equal(oo, go)
    ok = oe.assign_level_by_best_desc(img, desc, tk)
    gk = ge.AssignKPtLevelByBestDesc(desc, img, tk)
    assert np.array_equal(ok.view(np.uint8), gk.view(np.uint8))
    assert (gk["octave"] != tk["octave"]).any()
    assert ge.ctx.L.eorb_orb_tracked_descriptors(ge.ctx.h, None, 0, 0, 0, None, 0, None, None) == -1      # empty image
    ge.ctx.close()


def _feature_vector(nfeat, nnodes, rng):
    """A DBoW2::FeatureVector stand-in: every feature falls into exactly one node (node ids ascending)."""
    node_of = rng.integers(0, nnodes, nfeat)
    ids = np.unique(node_of)
    off = [0]; idx = []
    for nid in ids:
        members = np.nonzero(node_of == nid)[0]
        rng.shuffle(members)                       # vector order inside a node is the insertion order, not sorted
        idx.extend(members.tolist()); off.append(len(idx))
    return (ids.astype(np.uint32) * 7 + 3), np.array(off, np.int32), np.array(idx, np.int32)


@pytest.mark.parametrize("ori", [True, False])
def test_search_by_bow(oracle, fe, ctx, ori):
    k1, d1, k2, d2 = _two_frames(oracle, seed=41, shift=2)
    rng = np.random.default_rng(11)
    # correlated node assignment: matching features mostly share a node, as a vocabulary tree would give
    nn = 60
    kfv = _feature_vector(len(k1), nn, rng)
    # frame features inherit the node of their spatially nearest KeyFrame feature most of the time
    node_kf = np.zeros(len(k1), np.int64)
    for a in range(len(kfv[0])):
        node_kf[kfv[2][kfv[1][a]:kfv[1][a + 1]]] = kfv[0][a]
    dx = k2["x"][:, None] - (k1["x"][None, :] - 2); dy = k2["y"][:, None] - (k1["y"][None, :] + 2)
    near = np.argmin(dx * dx + dy * dy, axis=1)
    node_f = np.where(rng.uniform(size=len(k2)) < 0.85, node_kf[near], rng.integers(0, nn, len(k2)) * 7 + 3)
    ids = np.unique(node_f); off = [0]; idx = []
    for nid in ids:
        m = np.nonzero(node_f == nid)[0]; rng.shuffle(m); idx.extend(m.tolist()); off.append(len(idx))
    ffv = (ids.astype(np.uint32), np.array(off, np.int32), np.array(idx, np.int32))
    has_mp = (rng.uniform(size=len(k1)) < 0.8).astype(np.uint8)
    for ratio in (0.7, 0.95):
        on, om = oracle.search_by_bow(k1, d1, has_mp, kfv, k2, d2, ffv, ratio, ori)
        gn, gm = fe.SearchByBoW(k1, d1, has_mp, kfv, k2, d2, ffv, ratio, ori, ctx=ctx)
        assert on == gn and np.array_equal(om, gm)
    assert on > 20


@pytest.mark.parametrize("ori", [True, False])
def test_search_by_bow_keyframes(oracle, fe, ctx, ori):
    """f3: ORBmatcher::SearchByBoW(KF, KF) (:833-973): strict TH_LOW, vbMatched2, output per idx1."""
    k1, d1, k2, d2 = _two_frames(oracle, seed=43, shift=3)
    rng = np.random.default_rng(12)
    nn = 50
    fv1 = _feature_vector(len(k1), nn, rng)
    node1 = np.zeros(len(k1), np.int64)
    for a in range(len(fv1[0])):
        node1[fv1[2][fv1[1][a]:fv1[1][a + 1]]] = fv1[0][a]
    dx = k2["x"][:, None] - (k1["x"][None, :] - 3); dy = k2["y"][:, None] - (k1["y"][None, :] + 3)
    near = np.argmin(dx * dx + dy * dy, axis=1)
    node2 = np.where(rng.uniform(size=len(k2)) < 0.85, node1[near], rng.integers(0, nn, len(k2)) * 7 + 3)
    ids = np.unique(node2); off = [0]; idx = []
    for nid in ids:
        m = np.nonzero(node2 == nid)[0]; rng.shuffle(m); idx.extend(m.tolist()); off.append(len(idx))
    fv2 = (ids.astype(np.uint32), np.array(off, np.int32), np.array(idx, np.int32))
    h1 = (rng.uniform(size=len(k1)) < 0.8).astype(np.uint8)
    h2 = (rng.uniform(size=len(k2)) < 0.8).astype(np.uint8)
    for ratio in (0.7, 0.95):
        on, om = oracle.search_by_bow_kf(k1, d1, h1, fv1, k2, d2, h2, fv2, ratio, ori)
        gn, gm = fe.SearchByBoW_KF(k1, d1, h1, fv1, k2, d2, h2, fv2, ratio, ori, ctx=ctx)
        assert on == gn and np.array_equal(om, gm)
        assert np.all(h1[om >= 0] == 1) and np.all(h2[om[om >= 0]] == 1)
    assert on > 20
    # empty sides
    e = (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.int32))
    gn, gm = fe.SearchByBoW_KF(k1, d1, h1, fv1, k2[:0], d2[:0], h2[:0], e, 0.8, ori, ctx=ctx)
    assert gn == 0 and np.all(gm == -1)


def _correlated_fvs(k1, k2, shift, nn, rng):
    """Feature vectors in which spatially matching features mostly share a vocabulary node."""
    fv1 = _feature_vector(len(k1), nn, rng)
    node1 = np.zeros(len(k1), np.int64)
    for a in range(len(fv1[0])):
        node1[fv1[2][fv1[1][a]:fv1[1][a + 1]]] = fv1[0][a]
    dx = k2["x"][:, None] - (k1["x"][None, :] - shift); dy = k2["y"][:, None] - (k1["y"][None, :] + shift)
    near = np.argmin(dx * dx + dy * dy, axis=1)
    node2 = np.where(rng.uniform(size=len(k2)) < 0.85, node1[near], rng.integers(0, nn, len(k2)) * 7 + 3)
    ids = np.unique(node2); off = [0]; idx = []
    for nid in ids:
        m = np.nonzero(node2 == nid)[0]; rng.shuffle(m); idx.extend(m.tolist()); off.append(len(idx))
    return fv1, (ids.astype(np.uint32), np.array(off, np.int32), np.array(idx, np.int32))


@pytest.mark.parametrize("ori,coarse", [(True, False), (False, False), (True, True)])
def test_search_for_triangulation(oracle, fe, ctx, ori, coarse):
    """f3: mono ORBmatcher::SearchForTriangulation (:975-1214) with the MixedMatcher eligibility gate."""
    k1, d1, k2, d2 = _two_frames(oracle, seed=47, shift=3)
    rng = np.random.default_rng(21)
    fv1, fv2 = _correlated_fvs(k1, k2, 3, 40, rng)
    e1 = (rng.uniform(size=len(k1)) < 0.75).astype(np.uint8)
    e2 = (rng.uniform(size=len(k2)) < 0.75).astype(np.uint8)
    scale = (1.2 ** np.arange(4)).astype(np.float32); sigma2 = (scale * scale).astype(np.float32)
    a = 0.7
    Fd = np.array([[0, 0, a], [0, 0, a], [-a, -a, 0]], np.float64)           # translation along (-1, +1): x+y is preserved
    total = 0
    for trial, ep in enumerate([(120.0, 90.0), (-500.0, 40.0), (60.5, 130.25)]):
        F = (Fd + rng.normal(0, 1e-5 if trial < 2 else 2e-3, (3, 3))).astype(np.float32)
        on, om = oracle.search_for_triangulation(k1, d1, e1, fv1, k2, d2, e2, fv2, ep, F, scale, sigma2, coarse, ori)
        gn, pairs = fe.SearchForTriangulation(k1, d1, e1, fv1, k2, d2, e2, fv2, ep, F, scale, sigma2, coarse, ori, ctx=ctx)
        ok = np.nonzero(om >= 0)[0]
        assert on == gn and np.array_equal(pairs, np.stack([ok, om[ok]], axis=1))
        assert np.all(e1[pairs[:, 0]] == 1) and np.all(e2[pairs[:, 1]] == 1)
        total += on
    assert total > 40
    # rectified stereo (:1051, :1079, :1093): keypoints with a right coordinate carry bit 1; a pair with one skips the epipole-distance
    # test.  The epipole is put among the keypoints so that the test bites: with the bits the result differs from the mono one.
    s1 = (rng.uniform(size=len(k1)) < 0.5).astype(np.uint8); s2 = (rng.uniform(size=len(k2)) < 0.5).astype(np.uint8)
    e1s = (e1 | (s1 << 1)).astype(np.uint8); e2s = (e2 | (s2 << 1)).astype(np.uint8)
    ep = (float(np.median(k2["x"])), float(np.median(k2["y"])))
    F = (Fd + rng.normal(0, 1e-5, (3, 3))).astype(np.float32)
    scale_big = (12.0 * scale).astype(np.float32)                           # (a wide exclusion disc around the epipole)
    on, om = oracle.search_for_triangulation(k1, d1, e1s, fv1, k2, d2, e2s, fv2, ep, F, scale_big, sigma2, coarse, ori)
    gn, pairs = fe.SearchForTriangulation(k1, d1, e1s, fv1, k2, d2, e2s, fv2, ep, F, scale_big, sigma2, coarse, ori, ctx=ctx)
    ok = np.nonzero(om >= 0)[0]
    assert on == gn and np.array_equal(pairs, np.stack([ok, om[ok]], axis=1))
    mn, mpairs = fe.SearchForTriangulation(k1, d1, e1, fv1, k2, d2, e2, fv2, ep, F, scale_big, sigma2, coarse, ori, ctx=ctx)
    assert gn > mn, (gn, mn)                                                # pairs inside the disc survive when one side is a stereo keypoint
    # octave outside the level tables is an argument error, not a fault
    bad = k2.copy(); bad["octave"][np.nonzero(e2)[0][0]] = 9
    with pytest.raises(fe.EorbError):
        fe.SearchForTriangulation(k1, d1, e1, fv1, bad, d2, e2, fv2, (0, 0), F, scale, sigma2, ctx=ctx)


def _radius_queries(k1, k2, d2, rng, M=1500):
    pick = rng.integers(0, len(k2), M)
    uv = np.stack([k2["x"][pick] + 3 + rng.normal(0, 1.5, M), k2["y"][pick] - 3 + rng.normal(0, 1.5, M)], axis=1).astype(np.float32)
    uv[:20] = rng.uniform(-40, 300, (20, 2))                                  # outside / on the border of the grid
    level = (k2["octave"][pick] + rng.integers(0, 2, M)).astype(np.int32)
    qd = d2[pick].copy()
    flip = rng.uniform(size=(M, 256)) < 0.04
    qd ^= np.packbits(flip, axis=1)
    valid = (rng.uniform(size=M) < 0.9).astype(np.uint8)
    return valid, uv, level, qd


def test_kf_radius_match_fuse_sim3(oracle, fe, ctx):
    """f3: the search core of Fuse / SearchBySim3 / SearchByProjection(KF, Scw): independent and in-order variants."""
    k1, d1, k2, d2 = _two_frames(oracle, seed=53, shift=3)
    rng = np.random.default_rng(9)
    F1 = oracle.Frame(k1, d1, 240, 180); gb = fe.grid_bounds(240, 180)
    scale = (1.2 ** np.arange(6)).astype(np.float32)
    valid, uv, level, qd = _radius_queries(k1, k2, d2, rng)
    for th in (3.0, 7.5):
        radius = (np.float32(th) * scale[level]).astype(np.float32)
        obi, obd = oracle.kf_radius_match(F1, valid, uv, radius, level, qd)
        gbi, gbd = fe.KeyFrameRadiusMatch(k1, d1, gb, valid, uv, radius, level, qd, ctx=ctx)
        assert np.array_equal(obi, gbi) and np.array_equal(obd, gbd)
        assert (obi >= 0).sum() > 300 and np.all(obi[valid == 0] == -1)
    # Fuse: reprojection gate + TH_LOW
    inv_sigma2 = (1.0 / (scale * scale)).astype(np.float32)
    radius = (np.float32(3.0) * scale[level]).astype(np.float32)
    obi, obd = oracle.kf_radius_match(F1, valid, uv, radius, level, qd, inv_sigma2=inv_sigma2)
    gbi, gbd = fe.KeyFrameRadiusMatch(k1, d1, gb, valid, uv, radius, level, qd, inv_sigma2=inv_sigma2, ctx=ctx)
    assert np.array_equal(obi, gbi) and np.array_equal(obd, gbd)
    fused = fe.Fuse(k1, d1, gb, valid, uv, level, scale, inv_sigma2, qd, th=3.0, ctx=ctx)
    assert np.array_equal(fused, np.where(obd <= 50, obi, -1)) and (fused >= 0).sum() > 100
    # Fuse on a rectified-stereo KeyFrame: three-term error against 7.8 where the keypoint has a right coordinate (:1541-1553)
    uright = np.where(rng.uniform(size=len(k1)) < 0.6, k1["x"] - rng.uniform(2, 30, len(k1)), -1).astype(np.float32)
    q_ur = np.full(len(valid), -5, np.float32)
    hit = obi >= 0
    q_ur[hit] = np.where(uright[obi[hit]] >= 0, uright[obi[hit]], 0) + rng.normal(0, 1.5, hit.sum()).astype(np.float32)
    sbi, sbd = oracle.kf_radius_match(F1, valid, uv, radius, level, qd, inv_sigma2=inv_sigma2, uright=uright, q_ur=q_ur)
    gsi, gsd = fe.KeyFrameRadiusMatch(k1, d1, gb, valid, uv, radius, level, qd, inv_sigma2=inv_sigma2, uright=uright, q_ur=q_ur, ctx=ctx)
    assert np.array_equal(sbi, gsi) and np.array_equal(sbd, gsd)
    assert not np.array_equal(sbi, obi) and (sbi >= 0).sum() > 200             # the gate bites, both ways of it are taken
    assert (uright[sbi[sbi >= 0]] >= 0).sum() > 50 and (uright[sbi[sbi >= 0]] < 0).sum() > 50
    fs = fe.Fuse(k1, d1, gb, valid, uv, level, scale, inv_sigma2, qd, th=3.0, ctx=ctx, uright=uright, q_ur=q_ur)
    assert np.array_equal(fs, np.where(sbd <= 50, sbi, -1))
    with pytest.raises(ValueError):
        fe.KeyFrameRadiusMatch(k1, d1, gb, valid, uv, radius, level, qd, uright=uright, q_ur=q_ur, ctx=ctx)
    # SearchByProjection(KeyFrame, Scw, vpPoints, vpMatched, th, ratioHamming): in-order with vpMatched
    taken0 = (rng.uniform(size=len(k1)) < 0.2).astype(np.uint8)
    for ratio in (1.0, 0.8):
        o = oracle.kf_radius_match(F1, valid, uv, radius, level, qd, taken=taken0, accept_thr=50 * ratio)
        g = fe.KeyFrameRadiusMatch(k1, d1, gb, valid, uv, radius, level, qd, taken=taken0, accept_thr=50 * ratio, ctx=ctx)
        assert all(np.array_equal(a, b) for a, b in zip(o, g))
        assert o[2].sum() > taken0.sum() + 100 and not np.array_equal(o[0], obi)
    # SearchBySim3: both directions + agreement
    v2, uv2, lv2, qd2 = _radius_queries(k2, k1, d1, rng, M=len(k2))
    uv2[:, 0] -= 6; uv2[:, 1] += 6
    F2 = oracle.Frame(k2, d2, 240, 180)
    v1, uv1, lv1, qd1 = valid[:len(k1)], uv[:len(k1)], level[:len(k1)], qd[:len(k1)]
    # make the two directions consistent for part of the points: KF1 point i projects onto its true match in KF2 and back
    dx = k2["x"][None, :] - (k1["x"][:, None] - 3); dy = k2["y"][None, :] - (k1["y"][:, None] + 3)
    nn12 = np.argmin(dx * dx + dy * dy, axis=1)
    uv1 = np.stack([k2["x"][nn12], k2["y"][nn12]], axis=1).astype(np.float32); lv1 = k2["octave"][nn12].astype(np.int32); qd1 = d1.copy()
    uv2 = np.stack([k2["x"] + 3, k2["y"] - 3], axis=1).astype(np.float32); lv2 = k2["octave"].astype(np.int32); qd2 = d2.copy()
    v1 = np.ones(len(k1), np.uint8); v2 = np.ones(len(k2), np.uint8)
    r1 = (np.float32(7.5) * scale[lv1]).astype(np.float32); r2 = (np.float32(7.5) * scale[lv2]).astype(np.float32)
    a1, ad1 = oracle.kf_radius_match(F2, v1, uv1, r1, lv1, qd1)
    a2, ad2 = oracle.kf_radius_match(F1, v2, uv2, r2, lv2, qd2)
    vn1 = np.where(ad1 <= 100, a1, -1); vn2 = np.where(ad2 <= 100, a2, -1)
    exp = np.full(len(k1), -1, np.int32)
    for i1 in range(len(k1)):                                                 # :1951-1964
        if vn1[i1] >= 0 and vn2[vn1[i1]] == i1:
            exp[i1] = vn1[i1]
    nf, m12 = fe.SearchBySim3((k1, d1, gb, scale), (k2, d2, gb, scale), (v1, uv1, lv1, qd1), (v2, uv2, lv2, qd2), th=7.5, ctx=ctx)
    assert nf == (exp >= 0).sum() and np.array_equal(m12, exp) and nf > 100


@pytest.mark.parametrize("ori,orbdist", [(True, 100), (True, 64), (False, 100)])
def test_search_by_projection_keyframe(oracle, fe, ctx, ori, orbdist):
    """f3: relocalisation SearchByProjection(Frame, KeyFrame, sAlreadyFound, th, ORBdist) (:2189-2312) with the Mixed gate."""
    k1, d1, k2, d2 = _two_frames(oracle, seed=59, shift=3)
    rng = np.random.default_rng(31)
    is1 = (rng.uniform(size=len(k1)) < 0.85).astype(np.uint8); is2 = (rng.uniform(size=len(k2)) < 0.85).astype(np.uint8)
    k1 = k1.copy(); k2 = k2.copy()
    k1["class_id"] = k1["octave"]; k2["class_id"] = k2["octave"]
    uv = np.stack([k1["x"] - 3 + rng.normal(0, 1, len(k1)), k1["y"] + 3 + rng.normal(0, 1, len(k1))], axis=1).astype(np.float32)
    pred = (k1["octave"] + rng.integers(-1, 2, len(k1))).astype(np.int32)
    scale = (1.2 ** np.arange(8)).astype(np.float32)
    ls = scale[np.clip(pred, 0, 7)]
    valid = (rng.uniform(size=len(k1)) < 0.85).astype(np.uint8)
    cur_mp = np.where(rng.uniform(size=len(k2)) < 0.15, -2, -1).astype(np.int32)      # already-held slots are skipped
    mp_desc = d1.copy()
    Cur_o = oracle.Frame(k2, d2, 240, 180, is_orb=is2); Cur_g = fe.FrameView(k2, d2, 240, 180, is_orb=is2)
    for th in (10.0, 3.0):
        on, om = oracle.search_by_projection_kf(Cur_o, k1, is1, valid, uv, pred, ls, mp_desc, cur_mp, th, orbdist, ori)
        gn, gm = fe.ORBmatcher(0.9, ori, ctx).SearchByProjectionKF(Cur_g, k1, is1, valid, uv, pred, ls, mp_desc, cur_mp, th, orbdist)
        assert on == gn and np.array_equal(om, gm)
        assert np.all(gm[cur_mp == -2] == -2)
    assert on > 100
    # the threshold's edges: ORBdist 0 accepts exact copies only, a negative one nothing (phase 1 then lists no candidate at all:
    # its fixed point takes every LISTED entry as within the threshold)
    mp0 = d1.copy(); exact = rng.uniform(size=len(k1)) < 0.3
    pick = rng.integers(0, len(k2), len(k1))
    mp0[exact] = d2[pick[exact]]; uv0 = uv.copy(); uv0[exact] = np.stack([k2["x"][pick[exact]], k2["y"][pick[exact]]], axis=1)
    pred0 = pred.copy(); pred0[exact] = k2["octave"][pick[exact]]; is10 = is1.copy(); is10[exact] = is2[pick[exact]]
    ls0 = scale[np.clip(pred0, 0, 7)]
    for od in (0, -1):
        on, om = oracle.search_by_projection_kf(Cur_o, k1, is10, valid, uv0, pred0, ls0, mp0, cur_mp, 10.0, od, ori)
        gn, gm = fe.ORBmatcher(0.9, ori, ctx).SearchByProjectionKF(Cur_g, k1, is10, valid, uv0, pred0, ls0, mp0, cur_mp, 10.0, od)
        assert on == gn and np.array_equal(om, gm)
        assert (on > 20) if od == 0 else (on == 0)


@pytest.mark.parametrize("k,L,ragged", [(10, 3, False), (10, 4, False), (7, 5, True), (3, 1, False)])
def test_bow_transform(oracle, fe, ctx, k, L, ragged):
    """f4: DBoW2 TemplatedVocabulary::transform (BowVector + FeatureVector) on a synthetic vocabulary tree."""
    voc = synth.random_vocabulary(k, L, seed=5 + L, ragged=ragged)
    rng = np.random.default_rng(17)
    leaves = np.nonzero(voc["word_id"] >= 0)[0]
    n = 1500
    desc = voc["node_desc"][rng.choice(leaves, n)].copy()
    desc ^= np.packbits(rng.uniform(size=(n, 256)) < 0.08, axis=1)
    desc[:40] = rng.integers(0, 256, (40, 32), dtype=np.uint8)              # far from every node: ties and odd descents
    desc[40:80] = desc[40]                                                  # the same word many times: repeated addWeight
    for weighting, norm in ((0, 1), (1, 0), (2, 2), (3, 1), (0, 0)):
        V = fe.ORBVocabulary(voc, weighting, norm, ctx=ctx)
        for levelsup in (4, 1, 0, L + 2):
            ow, ov, ofv, owo, ono = oracle.bow_transform(voc, desc, levelsup, weighting, norm)
            gw, gv, gfv, gwo, gno = V.transform(desc, levelsup, return_assignments=True)
            assert np.array_equal(owo, gwo) and np.array_equal(ono, gno)
            assert np.array_equal(ow, gw) and np.array_equal(ov.view(np.uint64), gv.view(np.uint64))
            assert all(np.array_equal(a, b) for a, b in zip(ofv, gfv))
    assert len(ow) >= min(50, len(leaves) - 1) and (L < 3 or (owo < 0).sum() > 0)
    # the FeatureVector feeds SearchByBoW directly
    ow, ov, ofv, _, _ = oracle.bow_transform(voc, desc, 1, 0, 1)
    assert len(ofv[2]) == (owo >= 0).sum() and np.all(np.diff(ofv[0].astype(np.int64)) > 0)
    # empty input / malformed tree
    V = fe.ORBVocabulary(voc, 0, 1, ctx=ctx)
    gw, gv, gfv = V.transform(desc[:0])
    assert len(gw) == 0 and len(gfv[0]) == 0
    bad = dict(voc); bad["child_ids"] = voc["child_ids"].copy(); bad["child_ids"][0] = 0
    with pytest.raises(fe.EorbError):
        fe.ORBVocabulary(bad, ctx=ctx)


@pytest.mark.parametrize("win,maxLevel", [(23, 1), (23, 3), (15, 0), (9, 2)])
def test_klt_pyr_lk(oracle, fe, ctx, win, maxLevel):
    """f2: cv::calcOpticalFlowPyrLK as ELK_Tracker calls it: points, status and err bit-identical to the oracle restatement,
    with and without OPTFLOW_USE_INITIAL_FLOW, incl. points at the image border, flat patches and points that leave the image."""
    W, H = 240, 180
    img1 = synth.texture_image(W, H, seed=21)
    rng = np.random.default_rng(4)
    img2 = np.roll(img1, (2, -3), axis=(0, 1)).astype(np.int32) + rng.integers(-3, 4, (H, W))
    img2 = np.clip(img2, 0, 255).astype(np.uint8)
    img2[100:140, 150:200] = 90                                           # a flat block: minEig test and lost tracks
    e = oracle.OrbExtractor(600, 1.2, 4, 10, 0, edgeTh=19)
    _, k, _, _ = e.extract(img1)
    pts = np.stack([k["x"], k["y"]], axis=1).astype(np.float32)
    extra = np.array([[0.0, 0.0], [239.9, 179.9], [1.5, 100.25], [238.0, 3.0], [-20.0, 50.0], [300.0, 90.0], [120.0, -30.0],
                      [175.5, 120.5], [5.0, 5.0]], np.float32)
    pts = np.concatenate([pts + rng.uniform(-0.5, 0.5, pts.shape).astype(np.float32), extra])
    trk = fe.ELK_Tracker(win, maxLevel, 10, 0.03, ctx=ctx)
    for flags, guess in ((0, None), (4, pts + np.float32([-2.5, 1.5])), (8, None), (4, pts + np.float32([40.0, -40.0]))):
        on, os_, oe = oracle.calc_optical_flow_pyr_lk(img1, img2, pts, guess, win, maxLevel, 10, 0.03, flags)
        gn, gs, ge = trk.calcOpticalFlowPyrLK(img1, img2, pts, guess, flags)
        assert np.array_equal(os_, gs)
        assert np.array_equal(on.view(np.uint32), gn.view(np.uint32)) and np.array_equal(oe.view(np.uint32), ge.view(np.uint32))
    on, os_, oe = oracle.calc_optical_flow_pyr_lk(img1, img2, pts, None, win, maxLevel, 10, 0.03, 0)
    good = os_ == 1
    assert good.sum() > 200 and (~good).sum() >= 3
    d = on[good] - pts[good]
    assert np.median(np.abs(d - np.float32([-3, 2]))) < 0.2                # it is an optical-flow tracker
    # ELK_Tracker bookkeeping on top (refineTrackedPts): host logic
    trk.setRefImage(img1, np.concatenate([k, k[:len(extra)]]))
    nm, p1, m12, cnt, disp = trk.trackAndMatchCurrImage(img2)
    assert nm == (m12 >= 0).sum() and len(disp) == nm and nm > 200


def test_hamming_window_match(oracle, fe, ctx):
    """The shared inner loop of the windowed matchers: candidate lists from GetFeaturesInArea, best / second in visiting order."""
    k1, d1, k2, d2 = _two_frames(oracle, seed=61, shift=3)
    F2 = oracle.Frame(k2, d2, 240, 180)
    offs = [0]; cand = []
    rng = np.random.default_rng(3)
    for i in range(len(k1)):
        c = F2.features_in_area(float(k1["x"][i]) - 3, float(k1["y"][i]) + 3, float(rng.choice([0.5, 8.0, 30.0, 100.0])))
        cand.extend(c.tolist()); offs.append(len(cand))
    dd2 = d2.copy(); dd2[::7] = dd2[3]                                    # many equal distances: the visiting order decides
    o = oracle.hamming_window_match(d1, dd2, offs, cand)
    g = fe.HammingWindowMatch(d1, dd2, offs, cand, ctx=ctx)
    assert all(np.array_equal(a, b) for a, b in zip(o, g))
    assert (o[0] < 0).sum() > 0 and (o[2] >= 0).sum() > 100 and len(cand) > 5000
    with pytest.raises(fe.EorbError):
        fe.HammingWindowMatch(d1[:2], dd2, [0, 1, 2], [0, len(dd2)], ctx=ctx)


def test_distinctive_descriptors(oracle, fe, ctx):
    """f3: MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:349-423), batched; sizes 0, 1, 2, even/odd, > 64 rows."""
    rng = np.random.default_rng(5)
    sizes = [0, 1, 2, 3, 4, 7, 8, 33, 64, 65, 130, 301] + rng.integers(1, 40, 200).tolist()
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    desc = np.zeros((offs[-1], 32), np.uint8)
    for m, n in enumerate(sizes):
        if n == 0: continue
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        noise = rng.uniform(size=(n, 256)) < rng.uniform(0.02, 0.3, size=(n, 1))       # per-observation noise level
        desc[offs[m]:offs[m + 1]] = base[None, :] ^ np.packbits(noise, axis=1)
    desc[offs[5]:offs[6]] = desc[offs[5]]                                                # all equal: ties -> first row
    ob = oracle.distinctive_descriptors(desc, offs)
    gb = fe.ComputeDistinctiveDescriptors(desc, offs, ctx=ctx)
    assert np.array_equal(ob, gb)
    assert ob[0] == -1 and ob[5] == 0
    assert len(np.unique(ob[12:])) > 5


def test_mixed_frame_container_ops(oracle, fe, ctx):
    kps = synth.random_keypoints(1500, seed=12)
    kps["response"] = np.random.default_rng(3).integers(1, 40, 1500).astype(np.float32)      # many equal responses
    assert np.array_equal(oracle.sort_by_response(kps), fe.sortFeaturesResponse(kps, ctx))
    for args in ((1200, 600, 1500, 500), (1200, 300, 1500, 500), (800, 900, 1500, 500), (800, 300, 1500, 500), (0, 0, 1500, 500)):
        assert oracle.resolve_num_mixed(*args) == fe.resolveNumMixedPts(*args)


def test_frontend_batch_ragged_and_empty_slices(oracle, fe):
    """Slices of different lengths, including an empty one and one that is not a multiple of the 4096-event chunk."""
    W, H = 240, 180
    counts = [5000, 0, 12345, 1, 4096]
    slices = [synth.shapes_events(n, seed=60 + i, undistort=True) if n else np.zeros(0, synth.EVENT_DTYPE) for i, n in enumerate(counts)]
    B = len(counts)
    fb = fe.FrontEndBatch(W, H, 1.0, False, 500, 1.2, 3, 10, 0, 19, max_batch=B, max_events=max(counts), match=True)
    c, cap = fb.ctx, fb.cap
    ev16 = np.concatenate([fe.pack_events(s) for s in slices])
    d_ev = c.dev_alloc(max(ev16.nbytes, 16)); c.upload(d_ev, ev16)
    d_img = c.dev_alloc(B * W * H); d_kp = c.dev_alloc(B * cap * 28); d_desc = c.dev_alloc(B * cap * 32); d_n = c.dev_alloc(B * 4)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    fb.run_dev(d_ev, offs, d_img, d_kp, d_desc, d_n)
    c.sync()
    imgs = np.zeros((B, H, W), np.uint8); c.download(imgs, d_img)
    kps = np.zeros((B, cap), synth.KP_DTYPE); c.download(kps, d_kp)
    nk = np.zeros(B, np.int32); c.download(nk, d_n)
    oe = oracle.OrbExtractor(500, 1.2, 3, 10, 0, edgeTh=19)
    for b in range(B):
        _, ou, _ = oracle.ev2im_gauss(slices[b], W, H, 1.0, False, True)
        assert np.array_equal(ou, imgs[b]), b
        _, okp, _, _ = oe.extract(ou)
        assert nk[b] == len(okp) and np.array_equal(okp.view(np.uint8), kps[b, :nk[b]].view(np.uint8))
    assert (imgs[1] == 0).all()
    for p in (d_ev, d_img, d_kp, d_desc, d_n):
        c.dev_free(p)
    c.close()


def _maps(W=240, H=180):
    mx, my = synth.undistort_lut(W, H)
    return np.ascontiguousarray(mx, np.float32), np.ascontiguousarray(my, np.float32)


@pytest.mark.parametrize("check", [True, False])
def test_raw_undistort_events(oracle, fe, ctx, check):
    """f4: the rectification loop of EventDataStore::getEventChunkRectified (EventLoader.cpp:264-305) on device."""
    W, H = 240, 180
    mx, my = _maps(W, H)
    raw = synth.random_raw_events(70001, W, H, seed=4)                  # corners of the sensor map outside the image
    fe.EvImConverter.set_undistort_maps(mx, my, check, ctx=ctx)
    for tsf in (1.0, 1e6):
        o = oracle.undistort_events(raw, mx, my, W, H, check, tsf)
        g = fe.EvImConverter.undistort_events(raw, W, H, tsf, ctx=ctx)
        assert len(o) == len(g) and np.array_equal(o.view(np.uint8), g.view(np.uint8))
    assert (len(o) < len(raw)) == check
    assert len(fe.EvImConverter.undistort_events(raw[:0], W, H, ctx=ctx)) == 0
    bad = raw[:10].copy(); bad["x"][3] = W
    with pytest.raises(fe.EorbError):
        fe.EvImConverter.undistort_events(bad, W, H, ctx=ctx)


def test_parse_events_text(oracle, fe, ctx):
    """f4: the loader's text half on device: comments, blank lines, CRLF, no final newline, many timestamp widths."""
    rng = np.random.default_rng(23)
    n = 200000
    ts = np.cumsum(rng.integers(1, 2000, n)) * 1e-6
    lines = [b"# events: ts x y p", b"#second header", b""]
    for i in range(n):
        style = i % 5
        if style == 0: tss = "%.9f" % ts[i]
        elif style == 1: tss = "%.6f" % (1468941032.0 + ts[i])
        elif style == 2: tss = "%d" % int(ts[i] * 1e6)
        elif style == 3: tss = "%.3f" % ts[i]
        else: tss = "%.12f" % ts[i]
        sep = "\t" if i % 7 == 0 else " "
        lines.append(("%s%s%d%s%d %d%s" % (tss, sep, rng.integers(0, 240), sep, rng.integers(0, 180), rng.integers(0, 2),
                                             "\r" if i % 11 == 0 else "")).encode())
        if i % 50000 == 17: lines.append(b"   # a comment in the middle")
    lines.append(b"0000123.4500 007 8.000 1")
    for text in (b"\n".join(lines), b"\n".join(lines) + b"\n"):
        o = oracle.parse_events_text(text)
        g = fe.EvImConverter.parse_events_text(text, ctx=ctx)
        assert len(o) == n + 1 and len(g) == len(o) and np.array_equal(o.view(np.uint8), g.view(np.uint8))
    assert len(fe.EvImConverter.parse_events_text(b"", ctx=ctx)) == 0
    assert len(fe.EvImConverter.parse_events_text(b"# only a header\n\n", ctx=ctx)) == 0
    for bad in (b"1.0 2 3 1\n1e5 2 3 1\n", b"1.0 2 3\n", b"1.0 2.5 3 1\n", b"1.0 2 3 2\n", b"1.0 70000 3 1\n",
                b"12345678901234567890123 1 2 0\n", b"1.0 2 3 1 9\n", b"abc\n"):
        with pytest.raises(fe.EorbError):
            fe.EvImConverter.parse_events_text(b"0.5 1 1 1\n" + bad, ctx=ctx)
        with pytest.raises(ValueError):
            oracle.parse_events_text(b"0.5 1 1 1\n" + bad)


@pytest.mark.parametrize("sigma,pol", [(1.0, False), (1.0, True), (0.7, False), (1.5, True), (2.0, False)])
def test_raw_ev2im_gauss_equals_loader_then_ev2im_gauss(oracle, fe, ctx, sigma, pol):
    """f4 fused: raw sensor events + maps -> image must equal undistort -> ev2im_gauss of the reference, bit for bit."""
    W, H = 240, 180
    mx, my = _maps(W, H)
    _, raw = synth.shapes_events(60000, W, H, seed=71, return_raw=True, undistort=True)
    raw2 = synth.random_raw_events(30000, W, H, seed=9)                 # includes pixels that leave the image
    raw = np.concatenate([raw, raw2]); raw["t"] = np.arange(len(raw)) * 1e-6
    for check in (True, False):
        fe.EvImConverter.set_undistort_maps(mx, my, check, ctx=ctx)
        ev = oracle.undistort_events(raw, mx, my, W, H, check, 1.0)
        of, ou, omm = oracle.ev2im_gauss(ev, W, H, sigma, pol, True)
        gf, gu, gmm = fe.EvImConverter.ev2im_gauss_raw(raw, W, H, sigma, pol, True, ctx=ctx, return_all=True)
        assert np.array_equal(of.view(np.uint32), gf.view(np.uint32)) and np.array_equal(ou, gu)
        assert np.array_equal(np.asarray(omm, np.float32).view(np.uint32), gmm.view(np.uint32))
    # the table is rebuilt when sigma changes and reused otherwise: second call, same answer
    gf2 = fe.EvImConverter.ev2im_gauss_raw(raw, W, H, sigma, pol, False, ctx=ctx)
    assert np.array_equal(gf2.view(np.uint32), gf.view(np.uint32))
    # empty
    z = fe.EvImConverter.ev2im_gauss_raw(raw[:0], W, H, sigma, pol, False, ctx=ctx)
    assert (z == 0).all()


def test_raw_mvsec_size_and_wide_stamps(oracle, fe, ctx):
    """346x260 sensor (C4 geometry) with maps that move pixels by several px and stamps of 11x11 / 17x17 taps (column stride 12 / 20)."""
    W, H = 346, 260
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    r2 = ((xx - 170.0) / 220.0) ** 2 + ((yy - 128.0) / 220.0) ** 2
    mx = (xx + (xx - 170.0) * (0.08 * r2 - 0.03 * r2 * r2) + 0.37).astype(np.float32)
    my = (yy + (yy - 128.0) * (0.08 * r2 - 0.03 * r2 * r2) - 0.21).astype(np.float32)
    raw = synth.random_raw_events(120000, W, H, seed=33)
    raw["x"][:30000] = np.clip(np.random.default_rng(1).normal(200, 6, 30000), 0, W - 1).astype(np.uint16)      # a hot spot
    raw["y"][:30000] = np.clip(np.random.default_rng(2).normal(90, 6, 30000), 0, H - 1).astype(np.uint16)
    np.random.default_rng(3).shuffle(raw)
    fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=ctx)
    ev = oracle.undistort_events(raw, mx, my, W, H, True, 1.0)
    assert 0 < len(ev) < len(raw)
    for sigma, pol in ((1.5, False), (2.5, True), (1.0, False)):
        of, ou, omm = oracle.ev2im_gauss(ev, W, H, sigma, pol, True)
        gf, gu, gmm = fe.EvImConverter.ev2im_gauss_raw(raw, W, H, sigma, pol, True, ctx=ctx, return_all=True)
        assert np.array_equal(of.view(np.uint32), gf.view(np.uint32)) and np.array_equal(ou, gu)


def test_raw_ev2im_count(oracle, fe, ctx):
    W, H = 240, 180
    mx, my = _maps(W, H)
    raw = synth.random_raw_events(50000, W, H, seed=19)
    fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=ctx)
    ev = oracle.undistort_events(raw, mx, my, W, H, True, 1.0)
    for pol in (False, True):
        of, ou, omm = oracle.ev2im(ev, W, H, pol, True)
        gf, gu, gmm = fe.EvImConverter.ev2im_raw(raw, W, H, pol, True, ctx=ctx, return_all=True)
        assert np.array_equal(of.view(np.uint32), gf.view(np.uint32))
        assert (ou is None) == (gu is None) and (ou is None or np.array_equal(ou, gu))


def test_polarity_extremes_of_one_sided_images(oracle, fe, ctx):
    """resolveMinMaxVals (:32-39) is a RUNNING maximum / minimum over every add: with only negative events the maximum is the
    least negative single tap, not 0 (and the normalised image follows from it); both gathers, raw and float events."""
    W, H = 240, 180
    mx, my = _maps(W, H)
    fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=ctx)
    for n, polbit in ((1, 0), (1, 1), (700, 0), (700, 1)):
        raw = synth.random_raw_events(n, W, H, seed=40 + n + polbit)
        raw["x"] = np.clip(raw["x"], 20, 200); raw["y"] = np.clip(raw["y"], 20, 150); raw["p"] = polbit
        ev = oracle.undistort_events(raw, mx, my, W, H, True, 1.0)
        for sigma in (0.5, 1.0, 2.0):
            of, ou, omm = oracle.ev2im_gauss(ev, W, H, sigma, True, True)
            gf, gu, gmm = fe.EvImConverter.ev2im_gauss_raw(raw, W, H, sigma, True, True, ctx=ctx, return_all=True)
            hf, hu, hmm = fe.EvImConverter.ev2im_gauss(ev, W, H, sigma, True, True, ctx=ctx, return_all=True)
            for f, u, mm in ((gf, gu, gmm), (hf, hu, hmm)):
                assert np.array_equal(np.asarray(omm, np.float32).view(np.uint32), mm.view(np.uint32)), (n, polbit, sigma, omm, mm)
                assert np.array_equal(of.view(np.uint32), f.view(np.uint32)) and np.array_equal(ou, u)
    # taps that underflow to 0 make the running maximum depend on visits that add nothing: rejected with polarity, exact without
    raw = synth.random_raw_events(500, W, H, seed=3)
    ev = oracle.undistort_events(raw, mx, my, W, H, True, 1.0)
    with pytest.raises(fe.EorbError):
        fe.EvImConverter.ev2im_gauss_raw(raw, W, H, 0.1, True, True, ctx=ctx)
    with pytest.raises(fe.EorbError):
        fe.EvImConverter.ev2im_gauss(ev, W, H, 0.1, True, True, ctx=ctx)
    of, ou, omm = oracle.ev2im_gauss(ev, W, H, 0.1, False, True)
    for f, u, mm in (fe.EvImConverter.ev2im_gauss_raw(raw, W, H, 0.1, False, True, ctx=ctx, return_all=True),
                     fe.EvImConverter.ev2im_gauss(ev, W, H, 0.1, False, True, ctx=ctx, return_all=True)):
        assert np.array_equal(of.view(np.uint32), f.view(np.uint32)) and np.array_equal(ou, u)
        assert np.array_equal(np.asarray(omm, np.float32).view(np.uint32), mm.view(np.uint32))


@pytest.mark.parametrize("form", [0, 1, 2, 3, 4])
def test_raw_accumulation_random_sweep(oracle, fe, form):
    """A seeded sweep over image sizes that are not multiples of the tile, stamps of 3x3 ... 17x17 taps, polarity, maps that throw
    pixels out of the image with and without checkInImage, event counts on the 64-entry batch boundaries and hot pixels
    (tests/fuzz/fuzz_raw.py runs the long version).  form: the gather kernel -- chosen by the batch's shape (0: up to 16 384 events
    per call take the binning-free kernel), the pipelined workgroup per tile whatever the shape (1), the wave per tile whatever the
    shape (2), no binning whatever the size (3), the two-byte slot lists wherever they apply (4: no polarity, sigma <= 4/3, at most 254
    slots per tile; the pipelined workgroup per tile elsewhere)."""
    ctx = fe.Context()
    ctx.debug_option("gather_form", form)
    rng = np.random.default_rng(2024)
    for case in range(80 if form == 0 else 50):
        W, H = [(240, 180), (346, 260), (64, 48), (33, 17), (100, 9), (16, 16), (250, 131)][rng.integers(0, 7)]
        LW, LH = W + int(rng.integers(0, 5)), H + int(rng.integers(0, 5))
        sigma = float([0.21, 0.4, 0.5, 0.8, 1.0, 1.0, 1.3, 1.7, 2.0, 2.5][rng.integers(0, 10)])
        pol, check = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        n = int([0, 1, 63, 64, 65, 127, 128, 129, 1000, 4096, 4097, 20000][rng.integers(0, 12)])
        yy, xx = np.mgrid[0:LH, 0:LW].astype(np.float64)
        amp = rng.uniform(0, 6)
        mx = (xx + amp * np.sin(yy / 17.0) + rng.uniform(-3, 3)).astype(np.float32)
        my = (yy + amp * np.cos(xx / 23.0) + rng.uniform(-3, 3)).astype(np.float32)
        raw = np.zeros(n, synth.RAW_DTYPE)
        if rng.integers(0, 2):
            raw["x"] = rng.integers(0, LW, n); raw["y"] = rng.integers(0, LH, n)
        else:
            raw["x"] = np.clip(rng.normal(LW * rng.uniform(0, 1), 2.5, n), 0, LW - 1)
            raw["y"] = np.clip(rng.normal(LH * rng.uniform(0, 1), 2.5, n), 0, LH - 1)
        raw["p"] = rng.integers(0, 2, n); raw["t"] = np.arange(n) * 1e-6
        fe.EvImConverter.set_undistort_maps(mx, my, check, ctx=ctx)
        ev = oracle.undistort_events(raw, mx, my, W, H, check, 1.0)
        of, ou, omm = oracle.ev2im_gauss(ev, W, H, sigma, pol, True)
        gf, gu, gmm = fe.EvImConverter.ev2im_gauss_raw(raw, W, H, sigma, pol, True, ctx=ctx, return_all=True)
        what = dict(case=case, W=W, H=H, sigma=sigma, pol=pol, check=check, n=n, form=form)
        assert np.array_equal(of.view(np.uint32), gf.view(np.uint32)) and np.array_equal(ou, gu), what
        assert np.array_equal(np.asarray(omm, np.float32).view(np.uint32), gmm.view(np.uint32)), what
    ctx.close()


def test_slot_lists_form(oracle, fe):
    """The slot form of the raw accumulation (ev_slots.hip: two-byte entries, a tile position's rows in LDS, one ds_read_addtid per
    entry): f32 image, running extremes and u8 image against the oracle for one slice of 1 ... 300 000 events (list lengths around
    the 512-entry block and the 16-entry loop step, hot pixels, empty tiles), the DAVIS and the MVSEC sensor, three stamp sizes (and a fourth that falls back),
    maps with and without checkInImage; then slices of a batch against the same slices alone.  The test hook counters show that the
    slot form ran and raised no flag."""
    ctx = fe.Context()
    ctx.debug_option("gather_form", 4)
    rng = np.random.default_rng(7)
    calls = 0
    for (W, H), check in (((240, 180), True), ((346, 260), False), ((64, 48), True)):
        mx, my = _maps(W, H) if (W, H) == (240, 180) else (None, None)
        if mx is None:
            yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
            mx = (xx + 2.5 * np.sin(yy / 19.0) - 1.0).astype(np.float32); my = (yy + 2.0 * np.cos(xx / 29.0) + 0.5).astype(np.float32)
        fe.EvImConverter.set_undistort_maps(mx, my, check, ctx=ctx)
        for sigma in (1.0, 0.45, 0.7, 1.3):         # 1.3: (8 + 2 * 4)^2 = 256 slots per tile, one too many: served by the batch pipeline
            for n in (1, 15, 16, 17, 511, 512, 513, 5000, 70000, 300000):
                if n > 5000 and (sigma != 1.0 and (W, H) != (240, 180)):
                    continue
                if n >= 5000 and rng.integers(0, 2):
                    _, raw = synth.shapes_events(n, W, H, seed=int(rng.integers(1, 1000)), undistort=False, return_raw=True)
                else:
                    raw = synth.random_raw_events(n, W, H, seed=n + int(sigma * 10))
                    if n > 100 and rng.integers(0, 2):           # hot pixels: lists far longer than a block
                        raw["x"] = np.clip(rng.normal(W * 0.6, 1.5, n), 0, W - 1); raw["y"] = np.clip(rng.normal(H * 0.3, 1.5, n), 0, H - 1)
                ev = oracle.undistort_events(raw, mx, my, W, H, check, 1.0)
                of, ou, omm = oracle.ev2im_gauss(ev, W, H, sigma, False, True, fast=True)
                gf, gu, gmm = fe.EvImConverter.ev2im_gauss_raw(raw, W, H, sigma, False, True, ctx=ctx, return_all=True)
                calls += sigma <= 1.0
                what = (W, H, check, sigma, n)
                assert np.array_equal(of.view(np.uint32), gf.view(np.uint32)), (what, int((of.view(np.uint32) != gf.view(np.uint32)).sum()))
                assert np.array_equal(np.asarray(omm, np.float32).view(np.uint32), gmm.view(np.uint32)), (what, omm, gmm)
                assert np.array_equal(ou, gu), what
        # neighbours in the event stream whose stamps share a tile but start in different ones (one reaches it as its second tile, the
        # next as its first): the order inside the shared tile's list must still be the event order
        n = 60000
        raw = np.zeros(n, synth.RAW_DTYPE)
        bx, by = (W // 16) * 8, (H // 16) * 8                            # a tile corner near the middle of the image
        pat = rng.integers(0, 4, n)
        raw["x"] = np.clip(bx + np.where(pat & 1, 5, -3) + rng.integers(-1, 2, n), 0, W - 1)
        raw["y"] = np.clip(by + np.where(pat & 2, 5, -3) + rng.integers(-1, 2, n), 0, H - 1)
        raw["p"] = 1; raw["t"] = np.arange(n) * 1e-6
        ev = oracle.undistort_events(raw, mx, my, W, H, check, 1.0)
        of, ou, omm = oracle.ev2im_gauss(ev, W, H, 1.0, False, True, fast=True)
        gf, gu, gmm = fe.EvImConverter.ev2im_gauss_raw(raw, W, H, 1.0, False, True, ctx=ctx, return_all=True)
        calls += 1
        assert np.array_equal(of.view(np.uint32), gf.view(np.uint32)), ("tile corner", W, H, int((of.view(np.uint32) != gf.view(np.uint32)).sum()))
        assert ctx.debug_counter("slot_hot_items") >= 1              # (60 000 events on four tiles: lists long enough for the register-row kernel)
        assert np.array_equal(np.asarray(omm, np.float32).view(np.uint32), gmm.view(np.uint32)) and np.array_equal(ou, gu)
    assert ctx.debug_counter("slot_calls") == calls and ctx.debug_counter("slot_flags") == 0 and ctx.debug_counter("slot_rank_ok") == 1
    ctx.close()
    # a batch: every slice as it comes out alone
    W, H, B = 240, 180, 5
    mx, my = _maps(W, H)
    sizes = [40000, 0, 120000, 7, 65536]
    raws = [synth.shapes_events(max(sz, 1), W, H, seed=500 + b, motion=0.4, undistort=True, return_raw=True)[1][:sz] for b, sz in enumerate(sizes)]
    fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=B, max_events=max(sizes))
    c, cap = fb.ctx, fb.cap
    fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=c)
    blob = np.concatenate(raws)
    d_ev = c.dev_alloc(blob.nbytes); c.upload(d_ev, blob)
    d_img = c.dev_alloc(B * W * H); d_kp = c.dev_alloc(B * cap * 28); d_desc = c.dev_alloc(B * cap * 32)
    d_n = c.dev_alloc(B * 4); d_m = c.dev_alloc(B * cap * 4); d_nm = c.dev_alloc(B * 4)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    fb.run_dev(d_ev, off, d_img, d_kp, d_desc, d_n, d_m, d_nm, raw=True)
    c.sync()
    assert c.debug_counter("slot_calls") == 1 and c.debug_counter("slot_flags") == 0
    imgs = np.zeros((B, H, W), np.uint8); c.download(imgs, d_img)
    for b in range(B):
        ev = oracle.undistort_events(raws[b], mx, my, W, H, True, 1.0)
        _, ou, _ = oracle.ev2im_gauss(ev, W, H, 1.0, False, True, fast=True)
        assert np.array_equal(ou, imgs[b]), b
    for p_ in (d_ev, d_img, d_kp, d_desc, d_n, d_m, d_nm):
        c.dev_free(p_)
    c.close()


def test_profiling_scopes_can_be_selected(fe):
    """eorb_prof_enable / eorb_prof_only (what bench.py's roofline times come from): all scopes of a batch call, then only the
    accumulation's, each once per call, with a positive time."""
    W, H, B, n = 240, 180, 2, 50000
    raws = [synth.shapes_events(n, W, H, seed=70 + b, motion=0.3, undistort=True, return_raw=True)[1] for b in range(B)]
    fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=B, max_events=n)
    c, cap = fb.ctx, fb.cap
    mx, my = _maps(W, H)
    fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=c)
    blob = np.concatenate(raws)
    d_ev = c.dev_alloc(blob.nbytes); c.upload(d_ev, blob)
    d_img = c.dev_alloc(B * W * H); d_kp = c.dev_alloc(B * cap * 28); d_desc = c.dev_alloc(B * cap * 32)
    d_n = c.dev_alloc(B * 4); d_m = c.dev_alloc(B * cap * 4); d_nm = c.dev_alloc(B * 4)
    off = np.arange(B + 1, dtype=np.int64) * n
    run = lambda: fb.run_dev(d_ev, off, d_img, d_kp, d_desc, d_n, d_m, d_nm, raw=True)
    run(); c.sync()
    c.prof_reset(); c.prof_only(()); c.prof_enable(True)
    run(); run(); c.sync(); c.prof_enable(False)
    every = c.prof_results()
    assert {"ev_count", "ev_scan", "ev_scatter", "ev_gather", "orb_fast_cells", "orb_octree", "search_init"} <= set(every)
    assert all(cnt == 2 and ms > 0 for ms, cnt in every.values()), every
    c.prof_reset(); c.prof_only(("ev_scatter", "ev_gather")); c.prof_enable(True)
    run(); c.sync(); c.prof_enable(False)
    some = {k: v for k, v in c.prof_results().items() if v[1]}
    assert set(some) == {"ev_scatter", "ev_gather"} and all(cnt == 1 and ms > 0 for ms, cnt in some.values()), some
    c.prof_only(())
    for p_ in (d_ev, d_img, d_kp, d_desc, d_n, d_m, d_nm):
        c.dev_free(p_)
    c.close()


def test_slot_lists_many_slices(oracle, fe):
    """The slot form on batches with more slices than the chunk kernel takes as a kernel argument (> 256: descriptors made on the host),
    and with 256 and 255 (device side), ragged and empty slices among them, 256-event chunks: every slice's image as it comes out alone
    from the oracle."""
    W, H = 240, 180
    mx, my = _maps(W, H)
    rng = np.random.default_rng(21)
    for B in (300, 256, 255):
        sizes = rng.integers(0, 3000, B); sizes[rng.integers(0, B, 12)] = 0; sizes[0] = 2999
        raws = [synth.random_raw_events(int(max(sz, 1)), W, H, seed=900 + b)[:int(sz)] for b, sz in enumerate(sizes)]
        fb = fe.FrontEndBatch(W, H, 1.0, False, 400, 1.2, 1, 10, 0, 9, max_batch=B, max_events=3000, want_desc=False, match=False)
        c, cap = fb.ctx, fb.cap
        c.debug_option("gather_form", 4)
        fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=c)
        blob = np.concatenate(raws)
        d_ev = c.dev_alloc(max(blob.nbytes, 16)); c.upload(d_ev, blob)
        d_img = c.dev_alloc(B * W * H); d_kp = c.dev_alloc(B * cap * 28); d_desc = c.dev_alloc(B * cap * 32)
        d_n = c.dev_alloc(B * 4); d_m = c.dev_alloc(B * cap * 4); d_nm = c.dev_alloc(B * 4)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        fb.run_dev(d_ev, off, d_img, d_kp, d_desc, d_n, d_m, d_nm, raw=True)
        c.sync()
        assert c.debug_counter("slot_calls") == 1 and c.debug_counter("slot_flags") == 0
        imgs = np.zeros((B, H, W), np.uint8); c.download(imgs, d_img)
        for b in list(range(0, B, 17)) + [B - 1]:
            ev = oracle.undistort_events(raws[b], mx, my, W, H, True, 1.0)
            _, ou, _ = oracle.ev2im_gauss(ev, W, H, 1.0, False, True, fast=True)
            assert np.array_equal(ou, imgs[b]), (B, b, int(sizes[b]))
        for p_ in (d_ev, d_img, d_kp, d_desc, d_n, d_m, d_nm):
            c.dev_free(p_)
        c.close()


def test_register_row_kernel_list_tails(oracle, fe):
    """sl_hot_kernel walks a list 64 entries per scalar load and enters its unrolled sequence in the middle for the last (length mod 64)
    entries, with the bytes of the last dword past the end replaced by the null row: every residue of the length mod 64 (hence mod 4),
    on lists just long enough for that kernel (4 096 entries), against the oracle.  All events sit on a 3x3 patch of sensor pixels in the
    middle of a tile, so the tile's list holds every event."""
    W, H = 240, 180
    mx, my = _maps(W, H)
    ctx = fe.Context()
    ctx.debug_option("gather_form", 4)
    fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=ctx)
    rng = np.random.default_rng(11)
    hot_seen = 0
    for n in list(range(4096, 4096 + 66)) + [4096 + 127, 4096 + 128, 4096 + 129, 8191, 8192, 8193, 12345]:
        raw = np.zeros(n, synth.RAW_DTYPE)
        raw["x"] = 123 + rng.integers(0, 3, n); raw["y"] = 91 + rng.integers(0, 3, n); raw["p"] = 1; raw["t"] = np.arange(n) * 1e-6
        ev = oracle.undistort_events(raw, mx, my, W, H, True, 1.0)
        of, ou, omm = oracle.ev2im_gauss(ev, W, H, 1.0, False, True, fast=True)
        gf, gu, gmm = fe.EvImConverter.ev2im_gauss_raw(raw, W, H, 1.0, False, True, ctx=ctx, return_all=True)
        assert np.array_equal(of.view(np.uint32), gf.view(np.uint32)), (n, int((of.view(np.uint32) != gf.view(np.uint32)).sum()))
        assert np.array_equal(np.asarray(omm, np.float32).view(np.uint32), gmm.view(np.uint32)) and np.array_equal(ou, gu), n
        hot_seen += ctx.debug_counter("slot_hot_items") >= 1
    assert hot_seen >= 60
    ctx.close()


def _raw_batch(fe, raws, W, H, mx, my, opts=(), check=True, sigma=1.0):
    """One eorb_fe_run_batch_raw_dev call over the slices `raws` (detect-only extraction, no matching): u8 images, the float images
    behind them (eorb_fe_last_f32_dev), the running extremes and the slot form's test-hook counters."""
    B = len(raws)
    sizes = [len(r) for r in raws]
    fb = fe.FrontEndBatch(W, H, sigma, False, 400, 1.2, 1, 10, 0, 9, max_batch=B, max_events=max(max(sizes), 1), want_desc=False, match=False)
    c, cap = fb.ctx, fb.cap
    for k, v in opts:
        c.debug_option(k, v)
    fe.EvImConverter.set_undistort_maps(mx, my, check, ctx=c)
    blob = np.concatenate(raws)
    d_ev = c.dev_alloc(max(blob.nbytes, 16)); c.upload(d_ev, blob)
    d_img = c.dev_alloc(B * W * H); d_kp = c.dev_alloc(B * cap * 28); d_n = c.dev_alloc(B * 4)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    fb.run_dev(d_ev, off, d_img, d_kp, None, d_n, None, None, raw=True)
    c.sync()
    imgs = np.zeros((B, H, W), np.uint8); c.download(imgs, d_img)
    p32, mm = fb.last_f32(B)
    f32 = np.zeros((B, H, W), np.float32); c.download(f32, p32)
    cnt = {k: c.debug_counter(k) for k in ("slot_calls", "slot_flags", "slot_hot_items", "slot_hot_overflow", "slot_scatter_form", "slot_chunk", "slot_rank_ok", "slot_parts")}
    for p_ in (d_ev, d_img, d_kp, d_n):
        c.dev_free(p_)
    c.close()
    return imgs, f32, mm, cnt


def _check_raw_batch(oracle, raws, W, H, mx, my, got, what, check=True, sigma=1.0):
    imgs, f32, mm, _ = got
    for b, raw in enumerate(raws):
        ev = oracle.undistort_events(raw, mx, my, W, H, check, 1.0)
        of, ou, omm = oracle.ev2im_gauss(ev, W, H, sigma, False, True, fast=True)
        assert np.array_equal(of.view(np.uint32), f32[b].view(np.uint32)), (what, b, len(raw), int((of.view(np.uint32) != f32[b].view(np.uint32)).sum()))
        assert np.array_equal(np.asarray(omm, np.float32).view(np.uint32), mm[b].view(np.uint32)), (what, b, omm, mm[b])
        assert np.array_equal(ou, imgs[b]), (what, b)


def _hot_slices(B, n, W, H, seed):
    """Slices whose events pile up on a few tiles (60 % on 3x3-pixel patches around one of three centres), the rest anywhere."""
    rng = np.random.default_rng(seed)
    cents = [(123, 91), (60, 40), (200, 150)]
    raws = []
    for b in range(B):
        m = int(n * (0.5 + 0.5 * rng.uniform())) if b % 7 else n
        raw = synth.random_raw_events(m, W, H, seed=seed * 1000 + b)
        hot = rng.uniform(size=m) < 0.6
        cx, cy = cents[b % 3]
        raw["x"][hot] = np.clip(cx + rng.integers(-1, 2, int(hot.sum())) + 8 * rng.integers(0, 2, int(hot.sum())) * (b % 2), 0, W - 1)
        raw["y"][hot] = np.clip(cy + rng.integers(-1, 2, int(hot.sum())), 0, H - 1)
        raws.append(raw)
    return raws


def test_slot_hot_bucket_overflow_many_slices(oracle, fe):
    """The slot form at the benchmark's batch shape in small: 160 slices whose long lists compete for the register-row kernel's length
    buckets.  (a) the buckets as shipped; (b) buckets of 8 lists (test hook "slot_hot_cap"): the lists that find their bucket full are
    handed back to the LDS gather (ev_slots.hip, sl_plan_kernel: `the bucket is full`), counted by "slot_hot_overflow"; (c) the
    register-row kernel off: 160 lists per tile position through the LDS gather's longest-first tickets; (d) the batch as two halves on
    their own workspaces and streams (the second half binned under the first half's gather).  Every slice of every run:
    float image, running extremes and u8 image against the oracle, bit for bit (the `for k` loop of src/Event/EventConversion.cc:231-263)."""
    W, H, B = 240, 180, 160
    mx, my = _maps(W, H)
    raws = _hot_slices(B, 6000, W, H, seed=31)
    ref = None
    for name, opts, expect in (
            ("buckets as shipped", (("gather_form", 4), ("slot_hot_min", 256)), dict(hot=True, over=False)),
            ("buckets of 8", (("gather_form", 4), ("slot_hot_min", 256), ("slot_hot_cap", 8)), dict(hot=True, over=True)),
            ("register-row kernel off", (("gather_form", 4), ("slot_hot_min", 0)), dict(hot=False, over=False)),
            ("two halves in flight", (("gather_form", 4), ("slot_hot_min", 256), ("slot_halves", 1)), dict(hot=True, over=False))):
        got = _raw_batch(fe, raws, W, H, mx, my, opts)
        cnt = got[3]
        assert cnt["slot_calls"] == 1 and cnt["slot_flags"] == 0 and cnt["slot_parts"] == (2 if "halves" in name else 1), (name, cnt)
        assert (cnt["slot_hot_items"] > 0) == expect["hot"] and (cnt["slot_hot_overflow"] > 0) == expect["over"], (name, cnt)
        if expect["over"]:
            assert cnt["slot_hot_items"] <= 16 * 8, cnt
        if ref is None:
            _check_raw_batch(oracle, raws, W, H, mx, my, got, name)
            ref = got
        else:       # (the oracle's answer is the first run's, already compared)
            assert np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32)) and np.array_equal(ref[0], got[0]), name
            assert np.array_equal(ref[2].view(np.uint32), got[2].view(np.uint32)), name


def test_slot_scatter_ballot_form_equals_rank_form(oracle, fe):
    """The product's safety net for the rank scatter (whose order rests on the lane order of LDS atomics, checked on the device once per
    context) is the ballot scatter sl_scatter_kernel: both forms on the same batches (test hook "slot_rank"; the environment's
    EORB_SLOT_RANK=0 selects the same branch), against the oracle -- two 300 000-event slices (4 096-event chunks for the rank form,
    2 048 for the ballot form), a ragged batch with empty slices (256- / 1 024-event chunks) and the hot-tile slices."""
    W, H = 240, 180
    mx, my = _maps(W, H)
    rng = np.random.default_rng(77)
    batches = {
        "2 x 300k": [synth.shapes_events(300000, W, H, seed=810 + b, motion=0.4, undistort=True, return_raw=True)[1] for b in range(2)],
        "ragged": [synth.random_raw_events(max(int(sz), 1), W, H, seed=820 + b)[:int(sz)] for b, sz in enumerate([40000, 0, 25000, 7, 65536, 0, 1])],
        "hot tiles": _hot_slices(12, 20000, W, H, seed=32),
    }
    # (form 1 twice: with the ranks the count pass keeps -- sl_scatter_pre_kernel, the default -- and with the scatter ranking by itself,
    #  sl_scatter_rank_kernel, which sensors of more than 65 535 pixels still take)
    for name, raws in batches.items():
        for form, pre in ((1, 1), (1, 0), (0, 0)):
            got = _raw_batch(fe, raws, W, H, mx, my, (("gather_form", 4), ("slot_rank", form), ("slot_prerank", pre)))
            cnt = got[3]
            assert cnt["slot_calls"] == 1 and cnt["slot_flags"] == 0 and cnt["slot_rank_ok"] == 1, (name, form, cnt)
            assert cnt["slot_scatter_form"] == form, (name, form, cnt)
            if name == "2 x 300k":
                assert cnt["slot_chunk"] == (4096 if form else 2048), cnt
            _check_raw_batch(oracle, raws, W, H, mx, my, got, (name, form, pre))


def test_raw_dense_batch_on_vga_class_sensors(oracle, fe):
    """Dense raw batches on the 640x480 and 752x480 sensors of the reference's Examples/Event/*.yaml (4 800 / 5 640 tiles): the slot
    form runs with the rank scatter's larger LDS footprint; where that scatter is not available (ballot form: its per-wave count rows
    alone exceed 64 KB) the call falls back to the batch pipeline instead of failing.  Both against the oracle."""
    rng = np.random.default_rng(5)
    for (W, H) in ((640, 480), (752, 480)):
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
        mx = (xx + 3.0 * np.sin(yy / 41.0) - 1.5).astype(np.float32); my = (yy + 2.5 * np.cos(xx / 57.0) + 0.75).astype(np.float32)
        raws = []
        for b in range(2):
            n = 250000 - 70000 * b
            raw = synth.random_raw_events(n, W, H, seed=840 + b)
            hot = rng.uniform(size=n) < 0.3                       # a blob: long lists on a few tiles
            raw["x"][hot] = np.clip(rng.normal(W * 0.7, 6.0, int(hot.sum())), 0, W - 1); raw["y"][hot] = np.clip(rng.normal(H * 0.4, 6.0, int(hot.sum())), 0, H - 1)
            raws.append(raw)
        for form, slot_rank, slot_calls in ((0, 1, 1), (4, 1, 1), (0, 0, 0), (4, 0, 0)):
            got = _raw_batch(fe, raws, W, H, mx, my, (("gather_form", form), ("slot_rank", slot_rank)))
            assert got[3]["slot_calls"] == slot_calls and got[3]["slot_flags"] == 0, (W, H, form, slot_rank, got[3])
            _check_raw_batch(oracle, raws, W, H, mx, my, got, (W, H, form, slot_rank))


def test_packed_wire_records_equal_raw_records(oracle, fe):
    """eorb_raw_event4 (x | p << 15 | y << 16, a quarter of the bytes on the host -> HBM link) and eorb_raw_event2 (y * LW + x, an
    eighth) through the batch entry points: the images, keypoints and descriptors of the 16-byte records, for a dense batch (slot
    lists read the short records directly), a sparse one and a polarity image (both widen the records first; 4-byte only)."""
    W, H = 240, 180
    mx, my = _maps(W, H)
    for n, B, pol in ((60000, 3, False), (1500, 3, False), (30000, 2, True)):
        raws = [synth.shapes_events(n, W, H, seed=600 + b, motion=0.4, undistort=True, return_raw=True)[1] for b in range(B)]
        outs = []
        for fmt in ((True, 4) if pol else (True, 4, 2)):           # (the 2-byte record carries no polarity)
            fb = fe.FrontEndBatch(W, H, 1.0, pol, 1000, 1.2, 4, 10, 0, 19, max_batch=B, max_events=n)
            c, cap = fb.ctx, fb.cap
            fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=c)
            blob = np.concatenate(raws) if fmt is True else np.concatenate([fe.pack_raw_events4(r) if fmt == 4 else fe.pack_raw_events2(r, W) for r in raws])
            d_ev = c.dev_alloc(blob.nbytes); c.upload(d_ev, blob)
            d_img = c.dev_alloc(B * W * H); d_kp = c.dev_alloc(B * cap * 28); d_desc = c.dev_alloc(B * cap * 32); d_n = c.dev_alloc(B * 4)
            fb.run_dev(d_ev, np.arange(B + 1, dtype=np.int64) * n, d_img, d_kp, d_desc, d_n, raw=fmt)
            c.sync()
            imgs = np.zeros((B, H, W), np.uint8); c.download(imgs, d_img)
            nk = np.zeros(B, np.int32); c.download(nk, d_n)
            kps = np.zeros((B, cap), synth.KP_DTYPE); c.download(kps, d_kp)
            desc = np.zeros((B, cap, 32), np.uint8); c.download(desc, d_desc)
            outs.append((imgs, nk, [kps[b, :nk[b]].copy() for b in range(B)], [desc[b, :nk[b]].copy() for b in range(B)]))
            for p_ in (d_ev, d_img, d_kp, d_desc, d_n):
                c.dev_free(p_)
            c.close()
        a = outs[0]
        for b in outs[1:]:
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (n, B, pol)
            for x, y in zip(a[2] + a[3], b[2] + b[3]):
                assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
        ev0 = oracle.undistort_events(raws[0], mx, my, W, H, True, 1.0)
        assert np.array_equal(oracle.ev2im_gauss(ev0, W, H, 1.0, pol, True, fast=True)[1], outs[-1][0][0])


def test_host_entries_through_the_copy_engine():
    """The host-buffer entry points move small inputs and results through kernels that read / write pinned memory (and read a live
    slice's events and an image in place); beyond 1 MB, or when the pinned staging is taken, they use the copy engine.  The switches are
    read once per process: a child process runs a cross-section of the parity tests with every transfer forced through the copy engine,
    the events and images uploaded, and orientation / descriptors / output order as three launches."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EORB_UPLOAD_KERNEL_MAX="0", EORB_DOWNLOAD_KERNEL_MAX="0", EORB_SLICE_ZERO_COPY="0", EORB_IMAGE_ZERO_COPY="0",
               EORB_ORB_DESCRIBE="0")
    sel = ("test_orb_extract_texture or test_window_matchers_state_chains or test_search_by_bow_keyframes or test_search_for_triangulation or "
           "test_kf_radius_match_fuse_sim3 or test_bow_transform or test_klt_pyr_lk or test_distinctive_descriptors or "
           "test_ev2mci_se2_and_focus_contest or test_slice_calls_equal_the_separate_seams")
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_chain.py", "-x", "-q", "-m", "gpu", "-k", sel,
                        "-p", "no:cacheprovider"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    import re
    m = re.search(r"(\d+) passed", p.stdout)
    assert p.returncode == 0 and m and int(m.group(1)) >= 20, p.stdout[-3000:] + p.stderr[-2000:]


def test_raw_gather_four_column_variant(oracle):
    """ev_gather_raw_kernel has two instantiations: two tile columns per value wave (small launches) and four (launches of more
    than 32768 tiles, e.g. the benchmark's 64 slices).  The second one is forced here (EORB_GATHER_NC=4 is read once per process,
    hence the child process) and must give the same bit-exact images, also with polarity and wide stamps."""
    import os, subprocess, sys
    code = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from eorb_slam_amd import frontend as fe, synth
from oracle import oracle_py as orc
c = fe.Context()
for (W, H, n, sigma, pol, seed) in ((240, 180, 300000, 1.0, False, 5), (240, 180, 40000, 1.0, True, 6), (346, 260, 50000, 2.0, False, 7), (64, 48, 129, 0.5, True, 8)):
    mx, my = synth.undistort_lut(240, 180) if (W, H) == (240, 180) else tuple(np.ascontiguousarray(a, np.float32) for a in np.meshgrid(np.arange(W) + 0.3, np.arange(H) - 0.2))
    fe.EvImConverter.set_undistort_maps(np.ascontiguousarray(mx), np.ascontiguousarray(my), True, ctx=c)
    raw = synth.shapes_events(n, W, H, seed=seed, return_raw=True)[1] if W == 240 else synth.random_raw_events(n, W, H, seed=seed)
    ev = orc.undistort_events(raw, mx, my, W, H, True, 1.0)
    of, ou, omm = orc.ev2im_gauss(ev, W, H, sigma, pol, True)
    gf, gu, gmm = fe.EvImConverter.ev2im_gauss_raw(raw, W, H, sigma, pol, True, ctx=c, return_all=True)
    d = of.view(np.uint32) != gf.view(np.uint32)
    what = (W, H, n, sigma, pol, "f32 px", int(d.sum()), "u8 px", int((ou != gu).sum()), "minmax", list(map(float, omm)), gmm.tolist(),
            "tiles", sorted(set(zip((np.nonzero(d)[0] // 8).tolist(), (np.nonzero(d)[1] // 8).tolist())))[:8])
    assert not d.any() and np.array_equal(ou, gu), what
    assert np.array_equal(np.asarray(omm, np.float32).view(np.uint32), gmm.view(np.uint32)), what
print("nc4 ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EORB_GATHER_NC="4")
    p = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "nc4 ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_raw_needs_maps(fe):
    c = fe.Context()
    with pytest.raises(fe.EorbError):
        fe.EvImConverter.ev2im_gauss_raw(synth.random_raw_events(10), 240, 180, ctx=c)
    c.close()


def test_frontend_batch_raw_equals_float_path(oracle, fe):
    """The batched pipeline fed with raw sensor events gives the images / keypoints / matches of the float-event path."""
    W, H, B, n = 240, 180, 4, 50000
    mx, my = _maps(W, H)
    pairs = [synth.shapes_events(n, W, H, seed=80 + b, motion=0.3, undistort=True, return_raw=True) for b in range(B)]
    outs = []
    for use_raw in (False, True):
        fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=B, max_events=n)
        c, cap = fb.ctx, fb.cap
        if use_raw:
            fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=c)
            blob = np.concatenate([p[1] for p in pairs])
        else:
            blob = np.concatenate([fe.pack_events(p[0]) for p in pairs])
        d_ev = c.dev_alloc(blob.nbytes); c.upload(d_ev, blob)
        d_img = c.dev_alloc(B * W * H); d_kp = c.dev_alloc(B * cap * 28); d_desc = c.dev_alloc(B * cap * 32)
        d_n = c.dev_alloc(B * 4); d_m = c.dev_alloc(B * cap * 4); d_nm = c.dev_alloc(B * 4)
        fb.run_dev(d_ev, np.arange(B + 1, dtype=np.int64) * n, d_img, d_kp, d_desc, d_n, d_m, d_nm, raw=use_raw)
        c.sync()
        imgs = np.zeros((B, H, W), np.uint8); c.download(imgs, d_img)
        nk = np.zeros(B, np.int32); c.download(nk, d_n)
        kps = np.zeros((B, cap), synth.KP_DTYPE); c.download(kps, d_kp)
        desc = np.zeros((B, cap, 32), np.uint8); c.download(desc, d_desc)
        m12 = np.zeros((B, cap), np.int32); c.download(m12, d_m)
        nm = np.zeros(B, np.int32); c.download(nm, d_nm)
        outs.append((imgs, nk, [kps[b, :nk[b]].copy() for b in range(B)], [desc[b, :nk[b]].copy() for b in range(B)],
                     [m12[b, :nk[b - 1]].copy() for b in range(1, B)], nm[1:].copy()))
        for p in (d_ev, d_img, d_kp, d_desc, d_n, d_m, d_nm):
            c.dev_free(p)
        c.close()
    a, b = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[5], b[5])
    for x, y in zip(a[2] + a[3] + a[4], b[2] + b[3] + b[4]):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
    _, ou, _ = oracle.ev2im_gauss(pairs[2][0], W, H, 1.0, False, True)
    assert np.array_equal(ou, b[0][2]) and a[1].min() > 20


def test_large_batch_takes_the_many_workgroup_forms(oracle, fe):
    """96 slices x 4 levels = 384 octree workgroups (> 256 CUs: the second LDS placement, two workgroups per CU) and 96 x 690 = 66 240
    nearly empty tiles (the wave-per-tile gather with 8 tiles per wave): every slice must come out as it does in a batch of three
    (first placement, one tile per wave), and slice 0 is checked against the oracle."""
    W, H, n, B = 240, 180, 3000, 96
    mx, my = _maps(W, H)
    base = [synth.shapes_events(n, W, H, seed=300 + b, motion=0.4, undistort=True, return_raw=True) for b in range(6)]

    def run(idx):
        nb = len(idx)
        fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=nb, max_events=n)
        c, cap = fb.ctx, fb.cap
        fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=c)
        blob = np.concatenate([base[i][1] for i in idx])
        d_ev = c.dev_alloc(blob.nbytes); c.upload(d_ev, blob)
        d_img = c.dev_alloc(nb * W * H); d_kp = c.dev_alloc(nb * cap * 28); d_desc = c.dev_alloc(nb * cap * 32)
        d_n = c.dev_alloc(nb * 4); d_m = c.dev_alloc(nb * cap * 4); d_nm = c.dev_alloc(nb * 4)
        fb.run_dev(d_ev, np.arange(nb + 1, dtype=np.int64) * n, d_img, d_kp, d_desc, d_n, d_m, d_nm, raw=True)
        c.sync()
        imgs = np.zeros((nb, H, W), np.uint8); c.download(imgs, d_img)
        nk = np.zeros(nb, np.int32); c.download(nk, d_n)
        kps = np.zeros((nb, cap), synth.KP_DTYPE); c.download(kps, d_kp)
        desc = np.zeros((nb, cap, 32), np.uint8); c.download(desc, d_desc)
        for p in (d_ev, d_img, d_kp, d_desc, d_n, d_m, d_nm):
            c.dev_free(p)
        c.close()
        return imgs, nk, kps, desc

    big = run([b % 6 for b in range(B)])
    small = [run([i, (i + 1) % 6, (i + 2) % 6]) for i in (0, 3)]
    ref = {}
    for s, i0 in zip(small, (0, 3)):
        for k in range(3):
            ref[(i0 + k) % 6] = (s[0][k], int(s[1][k]), s[2][k], s[3][k])
    for b in range(B):
        img, nk, kps, desc = ref[b % 6]
        assert np.array_equal(big[0][b], img) and int(big[1][b]) == nk, b
        assert np.array_equal(big[2][b][:nk].view(np.uint8), kps[:nk].view(np.uint8)) and np.array_equal(big[3][b][:nk], desc[:nk]), b
    _, ou, _ = oracle.ev2im_gauss(base[0][0], W, H, 1.0, False, True)
    assert np.array_equal(ou, big[0][0]) and big[1].min() > 5


def test_frontend_batch_of_camera_frames(oracle, fe):
    """eorb_fe_run_batch_images_dev: B texture frames in HBM -> keypoints / descriptors of every frame and SearchForInitialization
    of frame b against frame b-1, equal to the oracle's per-frame extraction and matching (two calls: the last frame of a call is
    carried into the next one)."""
    W, H, B = 240, 180, 5
    base = synth.texture_image(W, H, seed=77)
    frames = [np.ascontiguousarray(np.roll(base, (i % 4, -(2 * i % 7)), axis=(0, 1))) for i in range(B)]
    oe = oracle.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=19)
    ref = [oe.extract(f) for f in frames]
    fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=3, max_events=1)
    c, cap = fb.ctx, fb.cap
    got = []
    for lo, hi in ((0, 3), (3, 5)):
        nb = hi - lo
        blob = np.concatenate([f.ravel() for f in frames[lo:hi]])
        d_img = c.dev_alloc(blob.nbytes); c.upload(d_img, blob)
        d_kp = c.dev_alloc(nb * cap * 28); d_desc = c.dev_alloc(nb * cap * 32); d_n = c.dev_alloc(nb * 4)
        d_m = c.dev_alloc(nb * cap * 4); d_nm = c.dev_alloc(nb * 4)
        fb.run_images_dev(d_img, nb, d_kp, d_desc, d_n, d_m, d_nm)
        c.sync()
        nk = np.zeros(nb, np.int32); c.download(nk, d_n)
        kps = np.zeros((nb, cap), synth.KP_DTYPE); c.download(kps, d_kp)
        desc = np.zeros((nb, cap, 32), np.uint8); c.download(desc, d_desc)
        m12 = np.zeros((nb, cap), np.int32); c.download(m12, d_m)
        nm = np.zeros(nb, np.int32); c.download(nm, d_nm)
        for j in range(nb):
            got.append((kps[j, :nk[j]].copy(), desc[j, :nk[j]].copy(), m12[j].copy(), int(nm[j])))
        for p in (d_img, d_kp, d_desc, d_n, d_m, d_nm):
            c.dev_free(p)
    c.close()
    for b in range(B):
        _, ok, od, _ = ref[b]
        assert len(ok) > 200
        assert np.array_equal(ok.view(np.uint8), got[b][0].view(np.uint8)) and np.array_equal(od, got[b][1]), b
        if b:
            _, pk, pd, _ = ref[b - 1]
            pm = np.stack([pk["x"], pk["y"]], axis=1)
            on, om, _ = oracle.search_for_initialization(oracle.Frame(pk, pd, W, H), oracle.Frame(ok, od, W, H), pm, 100, 0.9, True)
            assert on == got[b][3] and np.array_equal(om, got[b][2][:len(pk)]), b
            assert on > 20


def test_full_size_batch_properties(oracle, fe):
    """BASELINE.json configs[1] sizes (1 000 000 events per slice): the raw and the float inputs give the same images / keypoints,
    a slice's result does not depend on its position or company in the batch, and slices 0 and 1 are checked against the oracle:
    image, keypoints, descriptors and the match of slice 1 against slice 0."""
    W, H, N = 240, 180, 1000000
    mx, my = _maps(W, H)
    pairs = [synth.shapes_events(N, W, H, seed=2 + b, motion=0.5, undistort=True, return_raw=True) for b in range(3)]

    def run(order, use_raw):
        B = len(order)
        fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=B, max_events=N)
        c, cap = fb.ctx, fb.cap
        if use_raw:
            fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=c)
            blob = np.concatenate([pairs[i][1] for i in order])
        else:
            blob = np.concatenate([fe.pack_events(pairs[i][0]) for i in order])
        d_ev = c.dev_alloc(blob.nbytes); c.upload(d_ev, blob)
        d_img = c.dev_alloc(B * W * H); d_kp = c.dev_alloc(B * cap * 28); d_desc = c.dev_alloc(B * cap * 32); d_n = c.dev_alloc(B * 4)
        d_m = c.dev_alloc(B * cap * 4); d_nm = c.dev_alloc(B * 4)
        fb.run_dev(d_ev, np.arange(B + 1, dtype=np.int64) * N, d_img, d_kp, d_desc, d_n, d_m, d_nm, raw=use_raw)
        c.sync()
        imgs = np.zeros((B, H, W), np.uint8); c.download(imgs, d_img)
        nk = np.zeros(B, np.int32); c.download(nk, d_n)
        kps = np.zeros((B, cap), synth.KP_DTYPE); c.download(kps, d_kp)
        desc = np.zeros((B, cap, 32), np.uint8); c.download(desc, d_desc)
        m12 = np.zeros((B, cap), np.int32); c.download(m12, d_m)
        nm = np.zeros(B, np.int32); c.download(nm, d_nm)
        for p in (d_ev, d_img, d_kp, d_desc, d_n, d_m, d_nm):
            c.dev_free(p)
        c.close()
        return {i: (imgs[j].copy(), kps[j, :nk[j]].copy(), desc[j, :nk[j]].copy(), m12[j].copy(), int(nm[j])) for j, i in enumerate(order)}

    a = run([0, 1, 2], True)
    b = run([2, 0], False)
    c1 = run([1], True)
    for i, other in ((0, b), (2, b), (1, c1)):
        assert np.array_equal(a[i][0], other[i][0])
        assert np.array_equal(a[i][1].view(np.uint8), other[i][1].view(np.uint8)) and np.array_equal(a[i][2], other[i][2])
    oe = oracle.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=19, fast=True)
    ref = []
    for i in (0, 1):
        _, ou, _ = oracle.ev2im_gauss(pairs[i][0], W, H, 1.0, False, True, fast=True)
        assert np.array_equal(ou, a[i][0])
        _, okp, odesc, _ = oe.extract(ou)
        assert len(okp) == len(a[i][1]) and np.array_equal(okp.view(np.uint8), a[i][1].view(np.uint8)), "keypoints of slice %d" % i
        assert np.array_equal(odesc, a[i][2]), "descriptors of slice %d" % i
        ref.append((okp, odesc))
    pm = np.stack([ref[0][0]["x"], ref[0][0]["y"]], axis=1)
    on, om, _ = oracle.search_for_initialization(oracle.Frame(ref[0][0], ref[0][1], W, H), oracle.Frame(ref[1][0], ref[1][1], W, H), pm, 100, 0.9, True)
    assert on == a[1][4] and np.array_equal(om, a[1][3][:len(ref[0][0])]), "match of slice 1 against slice 0"
    assert len(a[1][1]) > 50


def test_large_slice_many_chunks(oracle, fe, ctx):
    """3 M events in one slice: 733 chunks per tile list; order must survive every chunk boundary."""
    ev = synth.shapes_events(3000000, seed=77, undistort=True)
    of, ou, omm = oracle.ev2im_gauss(ev, 240, 180, 1.0, False, True, fast=True)
    gf, gu, gmm = fe.EvImConverter.ev2im_gauss(ev, 240, 180, 1.0, False, True, ctx=ctx, return_all=True)
    assert _same_bits(of, gf) and np.array_equal(ou, gu) and _same_bits(omm, gmm)


def test_unsynced_ragged_batches_keep_their_own_tables(oracle, fe):
    """Back-to-back eorb_fe_run_batch_dev calls without a sync in between, every call with different ragged offsets: each batch must
    run with ITS chunk table (the host staging buffer is a ring guarded by events), results equal to one-call-one-sync runs."""
    W, H = 240, 180
    rng = np.random.default_rng(9)
    calls = []
    for k in range(6):
        B = int(rng.integers(1, 5))
        ns = [int(v) for v in rng.integers(0, 30000, B)]
        if k == 2:
            ns[0] = 0
        slices = [synth.shapes_events(n, W, H, seed=300 + 10 * k + b, motion=0.4) if n else np.zeros(0, synth.EVENT_DTYPE) for b, n in enumerate(ns)]
        calls.append((ns, slices))
    Bmax, nmax = 4, 4 * 30000

    def run(sync_each):
        fb = fe.FrontEndBatch(W, H, 1.0, False, 500, 1.2, 3, 10, 0, 19, max_batch=Bmax, max_events=nmax)
        c, cap = fb.ctx, fb.cap
        bufs = []
        for ns, slices in calls:
            B = len(ns)
            blob = np.concatenate([fe.pack_events(s) for s in slices]) if sum(ns) else np.zeros(1, fe.EV16_DTYPE)
            d_ev = c.dev_alloc(max(blob.nbytes, 16)); c.upload(d_ev, blob)
            d = dict(ev=d_ev, img=c.dev_alloc(B * W * H), kp=c.dev_alloc(B * cap * 28), desc=c.dev_alloc(B * cap * 32), n=c.dev_alloc(B * 4),
                     m=c.dev_alloc(B * cap * 4), nm=c.dev_alloc(B * 4), B=B, offs=np.concatenate([[0], np.cumsum(ns)]).astype(np.int64))
            bufs.append(d)
        for d in bufs:
            fb.run_dev(d["ev"], d["offs"], d["img"], d["kp"], d["desc"], d["n"], d["m"], d["nm"])
            if sync_each:
                c.sync()
        c.sync()
        out = []
        for d in bufs:
            B = d["B"]
            imgs = np.zeros((B, H, W), np.uint8); c.download(imgs, d["img"])
            nk = np.zeros(B, np.int32); c.download(nk, d["n"])
            kps = np.zeros((B, cap), synth.KP_DTYPE); c.download(kps, d["kp"])
            desc = np.zeros((B, cap, 32), np.uint8); c.download(desc, d["desc"])
            m12 = np.zeros((B, cap), np.int32); c.download(m12, d["m"])
            nm = np.zeros(B, np.int32); c.download(nm, d["nm"])
            out.append((imgs, nk, [kps[b, :nk[b]].tobytes() for b in range(B)], [desc[b, :nk[b]].tobytes() for b in range(B)], m12, nm))
            for k in ("ev", "img", "kp", "desc", "n", "m", "nm"):
                c.dev_free(d[k])
        c.close()
        return out

    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2] and x[3] == y[3]
        assert np.array_equal(x[5], y[5])
    # and the synced run is the oracle's (images of the first and the last call)
    for k in (0, len(calls) - 1):
        for bidx, s in enumerate(calls[k][1]):
            _, ou, _ = oracle.ev2im_gauss(s, W, H, 1.0, False, True)
            assert np.array_equal(ou, a[k][0][bidx])


def test_batched_overflow_is_reported_by_sync(oracle, fe):
    """The octree kernel ORs its overflow bits into a sticky device word; the batched path returns EORB_OK (it never synchronises)
    and eorb_sync reports EORB_E_CAPACITY once.  The overflow is forced through the test hook that shrinks the node pool."""
    W, H, B, n = 240, 180, 2, 60000
    slices = [synth.shapes_events(n, W, H, seed=50 + b, motion=0.3) for b in range(B)]
    c = fe.Context()
    c.debug_option("octree_pool_shrink", 400)
    fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=B, max_events=n, ctx=c)
    blob = np.concatenate([fe.pack_events(s) for s in slices])
    d_ev = c.dev_alloc(blob.nbytes); c.upload(d_ev, blob)
    fb.run_dev(d_ev, np.arange(B + 1, dtype=np.int64) * n)               # EORB_OK: nothing is read back here
    with pytest.raises(fe.EorbError) as e:
        c.sync()
    assert e.value.code == -3 and "capacity" in str(e.value)
    c.sync()                                                             # reported once
    # the host entry point reports it from the call itself
    img = _event_image(oracle)
    ge = fe.ORBextractor(1000, 1.2, 4, 10, 0, 19, (W, H), ctx=c)
    with pytest.raises(fe.EorbError) as e:
        ge(img)
    assert e.value.code == -3
    c.sync()
    # an overflow of a batch is not lost when a host call on the same context overflows before the sync: the host call reports its own
    # flag and leaves the sticky word of the batch alone
    fb.run_dev(d_ev, np.arange(B + 1, dtype=np.int64) * n)
    with pytest.raises(fe.EorbError):
        ge(img)
    with pytest.raises(fe.EorbError) as e:
        c.sync()
    assert e.value.code == -3
    c.sync()
    c.dev_free(d_ev); c.close()


@pytest.mark.parametrize("cfg", [
    dict(W=346, H=260, nfeatures=3000, scaleFactor=1.2, nlevels=1, edgeTh=15),          # one level of 3 000 features: > 160 KB
    dict(W=346, H=260, nfeatures=5000, scaleFactor=1.26, nlevels=6, edgeTh=15),         # the monocular initialiser on EvMVSEC_ETHZ.yaml
    dict(W=346, H=260, nfeatures=3000, scaleFactor=1.2, nlevels=8, edgeTh=15, th=(20, 7)),
    dict(W=240, H=180, nfeatures=1000, scaleFactor=1.2, nlevels=4, edgeTh=19, force_global=True),
])
def test_octree_working_set_in_global_memory(oracle, fe, cfg):
    """Octree items that do not fit the 160 KB of LDS live in global scratch (Tracking.cc:119-122 runs 5 x nFeatures for the monocular
    initialiser); results stay bit-exact.  The last case forces EVERY item into global memory on the default configuration."""
    W, H = cfg["W"], cfg["H"]
    ini, mn = cfg.get("th", (10, 0))
    img = synth.texture_image(W, H, seed=61)
    c = fe.Context()
    if cfg.get("force_global"):
        c.debug_option("octree_force_global", 1)
    oe = oracle.OrbExtractor(cfg["nfeatures"], cfg["scaleFactor"], cfg["nlevels"], ini, mn, edgeTh=cfg["edgeTh"], imWidth=W)
    ge = fe.ORBextractor(cfg["nfeatures"], cfg["scaleFactor"], cfg["nlevels"], ini, mn, cfg["edgeTh"], (W, H), ctx=c)
    for im in (img, np.roll(img, 7, axis=1)):
        omono, okp, odesc, ooob = oe.extract(im)
        gmono, gkp, gdesc, goob = ge(im)
        assert omono == gmono and len(okp) == len(gkp)
        assert np.array_equal(okp.view(np.uint8), gkp.view(np.uint8)) and np.array_equal(odesc, gdesc) and np.array_equal(ooob, goob)
    assert len(okp) > 0.6 * cfg["nfeatures"] or cfg["nfeatures"] >= 3000
    c.close()


def test_octree_dynamic_placement_and_its_fallback(oracle, fe):
    """Levels whose CAPACITY-sized working set does not fit the LDS get the key buffers and the points sized by the frame's candidates
    (the LDS kernel), and only levels whose candidates do not fit either are redone by the mixed-placement kernel: a sparse frame (every
    level fits), a frame crowded with corners at thresholds 1 / 1 (the fine levels do not), both against the oracle."""
    W, H = 752, 480
    rng = np.random.default_rng(71)
    sparse = synth.texture_image(W, H, seed=71)
    crowded = rng.integers(0, 256, (H, W)).astype(np.uint8)                # white noise: a corner on every other pixel
    seen = []
    for img, th in ((sparse, (20, 7)), (crowded, (1, 1))):
        c = fe.Context()
        oe = oracle.OrbExtractor(1200, 1.2, 8, th[0], th[1], edgeTh=19, imWidth=W)
        ge = fe.ORBextractor(1200, 1.2, 8, th[0], th[1], 19, (W, H), ctx=c)
        assert c.debug_counter("oct_dynamic") == 1
        omono, okp, odesc, _ = oe.extract(img)
        gmono, gkp, gdesc, _ = ge(img)
        seen.append(c.debug_counter("oct_redo_levels"))
        assert omono == gmono and len(okp) == len(gkp)
        assert np.array_equal(okp.view(np.uint8), gkp.view(np.uint8)) and np.array_equal(odesc, gdesc)
        c.close()
    assert seen[0] == 0 and seen[1] >= 1, seen


@pytest.mark.parametrize("list_algorithm", [0, 1])
def test_octree_direct_passes_equal_list_algorithm(oracle, fe, list_algorithm):
    """The octree's full passes are computed (two sorts of the candidates' cell paths) where a level's candidates fit the sort buffer,
    and executed pass by pass (the reference's list algorithm, restated) elsewhere: both against the oracle, over frames that leave the
    pass loop by every door -- the cut-off stage (:688-757), `size >= N` straight after a pass, nothing left to divide (fewer
    candidates than N), a single candidate, a wide frame with several root nodes, a square one with a single root."""
    cases = [
        (synth.texture_image(240, 180, seed=3), dict(nfeatures=1000, scaleFactor=1.2, nlevels=4, iniThFAST=10, minThFAST=0, edgeTh=19)),
        (synth.texture_image(240, 180, seed=5), dict(nfeatures=60, scaleFactor=1.2, nlevels=2, iniThFAST=10, minThFAST=0, edgeTh=19)),        # N far below the candidates
        (synth.texture_image(240, 180, seed=6), dict(nfeatures=17, scaleFactor=1.0, nlevels=1, iniThFAST=10, minThFAST=0, edgeTh=9)),
        (_event_image(oracle, n=6000, seed=7), dict(nfeatures=400, scaleFactor=1.0, nlevels=1, iniThFAST=0, minThFAST=0, edgeTh=9)),         # fewer candidates than N
        (_event_image(oracle, n=900, seed=8), dict(nfeatures=400, scaleFactor=1.0, nlevels=1, iniThFAST=0, minThFAST=0, edgeTh=9)),
        (synth.texture_image(640, 120, seed=9), dict(nfeatures=500, scaleFactor=1.2, nlevels=3, iniThFAST=10, minThFAST=0, edgeTh=19)),       # five roots
        (synth.texture_image(200, 200, seed=10), dict(nfeatures=300, scaleFactor=1.2, nlevels=3, iniThFAST=10, minThFAST=0, edgeTh=19)),
    ]
    one = np.zeros((180, 240), np.uint8); one[60:64, 100:104] = 255                        # a lone blob: a handful of corners
    cases.append((one, dict(nfeatures=400, scaleFactor=1.0, nlevels=1, iniThFAST=0, minThFAST=0, edgeTh=9)))
    seen_direct = 0
    for img, p in cases:
        H, W = img.shape
        c = fe.Context()
        c.debug_option("octree_list_algorithm", list_algorithm)
        oe = oracle.OrbExtractor(imWidth=W, **p)
        ge = fe.ORBextractor(imSize=(W, H), ctx=c, **p)
        seen_direct += 1 if (c.debug_counter("oct_direct_cap") & 0xffff) else 0
        for lap in ((0, 1000), (60, 120)):
            omono, okp, odesc, ooob = oe.extract(img, lap, True)
            gmono, gkp, gdesc, goob = ge(img, lap, True)
            assert omono == gmono and len(okp) == len(gkp), (p, len(okp), len(gkp))
            assert np.array_equal(okp.view(np.uint8), gkp.view(np.uint8)) and np.array_equal(odesc, gdesc), p
        c.close()
    assert (seen_direct == 0) if list_algorithm else (seen_direct == len(cases))


def _stereo_pair(seed, W=346, H=260, dmax=14):
    return synth.stereo_pair(seed, W, H, dmax)


@pytest.mark.parametrize("cfg", [dict(W=346, H=260, nf=1000, nl=8, th=(20, 7), mb=0.11, mbf=40.0, seed=31),
                                 dict(W=346, H=260, nf=2000, nl=8, th=(10, 1), mb=0.11, mbf=25.0, seed=32),     # many candidates per row band
                                 dict(W=240, H=180, nf=1000, nl=4, th=(10, 0), mb=0.11, mbf=3.0, seed=33),      # maxD = 27: most matches fall outside
                                 dict(W=752, H=480, nf=1200, nl=8, th=(20, 7), mb=0.11, mbf=47.9, seed=34)])    # EuRoC: Examples/Stereo/EuRoC.yaml
def test_frame_stereo_matches(oracle, fe, cfg):
    """Frame::Frame(imLeft, imRight) + ComputeStereoMatches (src/Frame.cc:97-152, :869-1048): keypoints / descriptors of both images,
    mvuRight and mvDepth as bit patterns, the number of correlated matches, against the oracle's restatement."""
    W, H = cfg["W"], cfg["H"]
    left, right = _stereo_pair(cfg["seed"], W, H)
    oL = oracle.OrbExtractor(cfg["nf"], 1.2, cfg["nl"], cfg["th"][0], cfg["th"][1], edgeTh=19, imWidth=W)
    oR = oracle.OrbExtractor(cfg["nf"], 1.2, cfg["nl"], cfg["th"][0], cfg["th"][1], edgeTh=19, imWidth=W)
    _, kL, dL, _ = oL.extract(left, (0, 0)); _, kR, dR, _ = oR.extract(right, (0, 0))
    our, odp, on = oL.compute_stereo_matches(oR, kL, dL, kR, dR, cfg["mb"], cfg["mbf"])
    ge = fe.ORBextractor(cfg["nf"], 1.2, cfg["nl"], cfg["th"][0], cfg["th"][1], 19, (W, H))
    for _ in range(2):                                                    # (the second call reuses the arena and the pyramids' buffers)
        g = ge.stereo(left, right, cfg["mb"], cfg["mbf"])
        assert len(g["kpsL"]) == len(kL) and len(g["kpsR"]) == len(kR)
        assert np.array_equal(g["kpsL"].view(np.uint8), kL.view(np.uint8)) and np.array_equal(g["kpsR"].view(np.uint8), kR.view(np.uint8))
        assert np.array_equal(g["descL"], dL) and np.array_equal(g["descR"], dR)
        assert g["nmatches"] == on
        assert np.array_equal(g["uRight"].view(np.uint32), our.view(np.uint32)) and np.array_equal(g["depth"].view(np.uint32), odp.view(np.uint32))
    matched = our > 0
    assert matched.sum() > (20 if cfg["mbf"] < 5 else 150), matched.sum()
    # the recovered disparities are the planted ones (sub-pixel): a sanity check of the test itself, not of parity
    d = (kL["x"] - our)[matched]
    assert np.median(np.abs(d - (2.0 + 12.0 * (0.5 + 0.5 * np.sin(kL["y"][matched] / 37.0 + kL["x"][matched] / 91.0))))) < 0.6
    ge.ctx.close()


def test_error_paths_return_codes(fe, ctx):
    ev = synth.random_events(10, seed=1)
    L = ctx.L
    import ctypes as C
    f32 = np.zeros((180, 240), np.float32)
    assert L.eorb_ev2im_gauss(ctx.h, ev.ctypes.data_as(C.c_void_p), 10, 240, 180, C.c_float(0.0), 0, 1, f32.ctypes.data_as(C.c_void_p), None, None) == -4
    assert L.eorb_ev2im_gauss(ctx.h, ev.ctypes.data_as(C.c_void_p), 10, 240, 180, C.c_float(4.0), 0, 1, f32.ctypes.data_as(C.c_void_p), None, None) == -2   # h = 12 > 8
    assert b"sigma" in L.eorb_last_error(ctx.h)
    assert L.eorb_ev2im_gauss(ctx.h, None, 10, 240, 180, C.c_float(1.0), 0, 1, None, None, None) == -4
    c2 = fe.Context()
    n = C.c_int(); m = C.c_int()
    img = np.zeros((180, 240), np.uint8)
    assert c2.L.eorb_orb_extract(c2.h, img.ctypes.data_as(C.c_void_p), 240, 180, 240, 0, 1000, 1, None, None, None, 0, C.byref(n), C.byref(m)) == -6   # not configured
    c2.close()


# ---- motion-compensated accumulation + focus -----------------------------------------------------------
EVETHZ_CAM = (199.092366542, 198.82882047, 132.192071378, 110.712660011)


@pytest.mark.parametrize("n,pol", [(6000, False), (6000, True), (1, False), (0, False)])
def test_ev2mci_se3(oracle, fe, ctx, n, pol):
    ev = synth.shapes_events(n, seed=81, undistort=True) if n else np.zeros(0, synth.EVENT_DTYPE)
    axis = np.array([0.12, -0.3, 0.946484], np.float64); axis /= np.linalg.norm(axis)
    t = np.array([0.013, -0.007, 0.002])
    depth = np.random.default_rng(1).uniform(0.8, 2.5, max(n, 1)).astype(np.float32)[:n]
    for kw in (dict(medDepth=1.7, depth=None), dict(medDepth=1.0, depth=depth)):
        for normalized in (False, True):
            of, ou, omm = oracle.ev2mci_se3(ev, EVETHZ_CAM, 0.031, axis, t, kw["medDepth"], 240, 180, 1.0, pol, normalized, depth=kw["depth"])
            gf, gu, gmm = fe.EvImConverter.ev2mci_gg_f_se3(ev, EVETHZ_CAM, 0.031, axis, t, kw["medDepth"], 240, 180, 1.0, pol, normalized,
                                                          depth_per_event=kw["depth"], ctx=ctx)
            assert _same_bits(of, gf)
            if n:
                assert _same_bits(omm, gmm)
            if normalized and n:
                assert np.array_equal(ou, gu)


# the MVSEC configuration's camera (Examples/Event/EvMVSEC_ETHZ.yaml:54-67): KannalaBrandt8 on 346x260
MVSEC_KB8 = (226.38018519795807, 226.15002947047415, 173.6470807871759, 133.73271487507847,
             -0.048031442223833355, 0.011330957517194437, -0.055378166304281135, 0.021500973881459395)


@pytest.mark.parametrize("n,pol", [(20000, False), (6000, True), (1, False)])
def test_ev2mci_kannala_brandt8(oracle, fe, ctx, n, pol):
    """ev2mci_gg_f with the fisheye camera: KannalaBrandt8::unproject (Newton iterations + tanf) and project (atan2f, sqrtf; double
    polynomial + double cos / sin for the SE3 form, float throughout for the SE2 form), src/CameraModels/KannalaBrandt8.cpp:87-190."""
    W, H = 346, 260
    ev = synth.random_events(n, W, H, seed=90 + n, frac=True)
    ev["ts"] = np.sort(ev["ts"])
    axis = np.array([0.2, -0.1, 0.97], np.float64); axis /= np.linalg.norm(axis)
    t = np.array([0.02, -0.01, 0.004])
    depth = np.random.default_rng(2).uniform(0.8, 2.5, n).astype(np.float32)
    for kw in (dict(medDepth=1.4, depth=None), dict(medDepth=1.0, depth=depth)):
        for normalized in (False, True):
            of, ou, omm = oracle.ev2mci_se3(ev, MVSEC_KB8, 0.045, axis, t, kw["medDepth"], W, H, 1.0, pol, normalized, depth=kw["depth"])
            gf, gu, gmm = fe.EvImConverter.ev2mci_gg_f_se3(ev, MVSEC_KB8, 0.045, axis, t, kw["medDepth"], W, H, 1.0, pol, normalized,
                                                          depth_per_event=kw["depth"], ctx=ctx)
            assert _same_bits(of, gf) and _same_bits(omm, gmm)
            if normalized:
                assert np.array_equal(ou, gu)
    for params in ([0.03, 0.004, -0.003], [-0.02, 0.002, 0.001, 0.96]):
        of, _, omm = oracle.ev2mci_se2(ev, MVSEC_KB8, params, W, H, 1.0, pol, False)
        gf, _, gmm = fe.EvImConverter.ev2mci_gg_f_se2(ev, MVSEC_KB8, params, W, H, 1.0, pol, False, ctx=ctx)
        assert _same_bits(of, gf) and _same_bits(omm, gmm)
    if n > 1:
        # the fisheye model matters: the same events through a pinhole with the same focal lengths give another image
        pf, _, _ = oracle.ev2mci_se3(ev, MVSEC_KB8[:4], 0.045, axis, t, 1.4, W, H, 1.0, pol, False)
        of, _, _ = oracle.ev2mci_se3(ev, MVSEC_KB8, 0.045, axis, t, 1.4, W, H, 1.0, pol, False)
        assert not _same_bits(pf, of)


def test_ev2mci_se2_and_focus_contest(oracle, fe, ctx):
    """The reconstruction contest of EvImBuilder::generateMCImage (EvImBuilder.cpp:1146-1247): event histogram vs SE3 vs SE2
    reconstructions, each scored by measureImageFocus and normalised with cv::normalize; best focus wins."""
    ev = synth.shapes_events(6000, seed=82, undistort=True, motion=1.5)
    axis = np.array([0.0, 0.0, 1.0]); t = np.array([0.01, 0.0, 0.0])
    cands = {}
    of, _, _ = oracle.ev2im_gauss(ev, 240, 180, 1.0, False, False)
    gf, _, _ = fe.EvImConverter.ev2im_gauss(ev, 240, 180, 1.0, False, False, ctx=ctx, return_all=True)
    cands["EH"] = (of, gf)
    of, _, _ = oracle.ev2mci_se3(ev, EVETHZ_CAM, 0.02, axis, t, 1.3, 240, 180)
    gf, _, _ = fe.EvImConverter.ev2mci_gg_f_se3(ev, EVETHZ_CAM, 0.02, axis, t, 1.3, 240, 180, ctx=ctx)
    cands["DP"] = (of, gf)
    for params in ([0.02, 0.004, -0.003], [-0.015, 0.002, 0.001, 0.97]):
        of, _, _ = oracle.ev2mci_se2(ev, EVETHZ_CAM, params, 240, 180)
        gf, _, _ = fe.EvImConverter.ev2mci_gg_f_se2(ev, EVETHZ_CAM, params, 240, 180, ctx=ctx)
        cands["A%d" % len(params)] = (of, gf)
    scores = {}
    for k, (of, gf) in cands.items():
        assert _same_bits(of, gf), k
        fo, fg = oracle.measure_image_focus(of), fe.EvImConverter.measureImageFocus(gf, ctx=ctx)
        assert np.float32(fo).tobytes() == np.float32(fg).tobytes(), k
        assert np.array_equal(oracle.cv_normalize_minmax_u8(of), fe.cv_normalize_minmax_u8(gf, ctx=ctx)), k
        scores[k] = fo
    assert len(set(scores.values())) == len(scores)
    # the contest's reconstructions scored in one call
    keys = list(cands)
    fn = fe.EvImConverter.measureImageFocusN(np.stack([cands[k][1] for k in keys]), ctx=ctx)
    assert [np.float32(scores[k]).tobytes() for k in keys] == [np.float32(v).tobytes() for v in fn]
    flat = np.zeros((180, 240), np.float32)
    assert np.array_equal(oracle.cv_normalize_minmax_u8(flat), fe.cv_normalize_minmax_u8(flat, ctx=ctx))
    assert oracle.measure_image_focus(flat) == fe.EvImConverter.measureImageFocus(flat, ctx=ctx) == 0.0


def test_two_ranks_on_one_gpu_gather_their_records(tmp_path):
    """The multi-GPU path of bench.py --gpus N with everything but the RCCL transport: shard.spawn_ranks starts two ranks that share
    this GPU, each runs its own sequence's batch through the HIP front end, the packed keypoint records are gathered to rank 0
    (gloo) and must equal, byte for byte, what each rank's sequence gives in a single process."""
    from eorb_slam_amd import shard
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rank_gpu_worker.py")
    out = tmp_path / "ranks.txt"
    rc = shard.spawn_ranks(script, [str(out)], 2, timeout_s=400)
    assert rc == 0 and out.read_text() == "ok 2"
