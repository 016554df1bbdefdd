"""Config 1 as the reference executes it: EvImBuilder::Track's chunk loop (src/Event/EvImBuilder.cpp:1300-1515) through the one-call
seams of the C ABI (eorb_ev_slice_extract / eorb_ev_slice_track / eorb_ev_mc_contest, host mirror frontend.EvImBuilder) against the
oracle's chain (oracle/orc_chain.py), chunk by chunk, bit for bit."""
import numpy as np
import pytest

from eorb_slam_amd import synth
import chain_cases as cc

pytestmark = pytest.mark.gpu


def _bits(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.nbytes == b.nbytes and np.array_equal(a.view(np.uint8).ravel(), b.view(np.uint8).ravel())


def _compare(i, g, o):
    assert g["state"] == o["state"] and g.get("skipped") == o.get("skipped") and g["dispatched"] == o["dispatched"], (i, g["state"], o["state"])
    if "image" in o:
        assert np.array_equal(g["image"], o["image"]), "chunk %d: event image" % i
    if "kps" in o:
        assert len(g["kps"]) == len(o["kps"]) and _bits(g["kps"], o["kps"]), "chunk %d: keypoints of the INIT frame" % i
    if "pts" in o:
        assert _bits(g["pts"], o["pts"]), "chunk %d: tracked points (%d differ)" % (i, int((g["pts"].view(np.uint32) != o["pts"].view(np.uint32)).sum()))
        assert np.array_equal(g["status"], o["status"]) and _bits(g["err"], o["err"]), "chunk %d: LK status / error" % i
        assert np.array_equal(g["matches12"], o["matches12"]) and g["nMatches"] == o["nMatches"], i
        assert np.float32(g["medPxDisp"]).view(np.uint32) == np.float32(o["medPxDisp"]).view(np.uint32), (i, g["medPxDisp"], o["medPxDisp"])
    assert g.get("chunk_size") == o.get("chunk_size"), (i, g.get("chunk_size"), o.get("chunk_size"))
    if o["dispatched"]:
        gm, om = g["mci"], o["mci"]
        assert gm["winner"] == om["winner"] and _bits(gm["focus"], om["focus"]), ("chunk %d: contest" % i, gm["focus"], om["focus"], gm["winner"], om["winner"])
        assert np.array_equal(gm["image"], om["image"]), "chunk %d: the winner's image" % i
        assert len(gm["l2_kps"]) == len(om["l2_kps"]) and _bits(gm["l2_kps"], om["l2_kps"]), "chunk %d: L2 keypoints on the winner" % i
        assert g["mci_good"] == o["mci_good"] and g["window"] == o["window"] and ("overlap" in g) == ("overlap" in o), i
        if "overlap" in o:
            assert _bits(g["overlap"], o["overlap"]), i


@pytest.mark.parametrize("cfg", [dict(), dict(l1FixedWinSz=True), dict(continTracking=False, maxPixelDisp=2.0)])
def test_l1_chain_equals_oracle_chain(oracle, cfg):
    from eorb_slam_amd import frontend
    from oracle import orc_chain
    W, H = 240, 180
    evs = cc.stream(n_chunks=64, chunk=2000, seed=5)
    g = frontend.EvImBuilder(W, H, cam=cc.CAM, **cfg)
    o = orc_chain.L1Chain(W, H, cam=cc.CAM, fast=True, **cfg)
    gres = orc_chain.run_sequence(lambda ch, mp: g.Track(ch, mp), evs, 2000, cc.mci_poses)
    ores = orc_chain.run_sequence(o.track, evs, 2000, cc.mci_poses)
    g.close()
    for i, (a, b) in enumerate(zip(gres, ores)):
        _compare(i, a, b)
    assert len(gres) == len(ores) >= (50 if cfg.get("continTracking", True) else 25), (len(gres), len(ores))
    ndisp = sum(r["dispatched"] for r in ores)
    assert ndisp >= 5, ndisp                                     # the window rule fired: contests and re-initialisations happened
    assert sum(1 for r in ores if "pts" in r) >= 15              # ... and LK tracked in between
    winners = {r["mci"]["winner"] for r in ores if r["dispatched"]}
    assert len(winners) >= 1


def test_contest_rules(oracle):
    """generateMCImage's decision rules on their own: absent methods, the first of equal keys wins (identical DP and BA poses), a
    winning event histogram is rebuilt from the later half of the window, an empty window returns no winner; windows above the
    binning-free kernel's size take the general accumulation."""
    from eorb_slam_amd import frontend
    from oracle import orc_chain
    W, H = 240, 180
    g = frontend.EvImBuilder(W, H, cam=cc.CAM)
    l2 = oracle.OrbExtractor(800, 1.0, 1, 0, 0, edgeTh=9, imWidth=W, fast=True)
    for n in (6000, 5999, 1, 2, 40000, 60000):           # (40 000: the binning-free kernel beyond a single slice's 16 384; 60 000: the general accumulation)
        evs = cc.stream(n_chunks=1, chunk=n, seed=9 + n % 7, motion=6.0)
        p = cc.mci_poses(evs)
        for poses in (None, dict(dp=p["dp"]), dict(dp=p["dp"], ba=p["dp"], se2=p["se2"]), dict(ba=p["ba"], se2=p["se2"]), p):
            gm = g.generateMCImage(evs, poses)
            om = orc_chain.generate_mc_image(evs, W, H, 1.0, cc.CAM, poses, l2, fast=True)
            what = (n, None if poses is None else sorted(poses))
            assert gm["winner"] == om["winner"] and _bits(gm["focus"], om["focus"]), (what, gm["focus"], om["focus"])
            assert np.array_equal(gm["image"], om["image"]) and _bits(gm["l2_kps"], om["l2_kps"]), what
            if poses is None:
                assert gm["winner"] == 2
            if poses is not None and "ba" in poses and poses.get("dp") is poses.get("ba"):
                assert gm["focus"][0] == gm["focus"][1] and gm["winner"] != 1
    gm = g.generateMCImage(evs[:0], None)
    assert gm["winner"] == -1 and len(gm["l2_kps"]) == 0
    g.close()


def test_slice_calls_equal_the_separate_seams(oracle):
    """eorb_ev_slice_extract / eorb_ev_slice_track against the separate host-buffer seams they fuse (eorb_ev2im_gauss[_raw] +
    eorb_orb_extract, eorb_calc_optical_flow_pyr_lk): float and raw events, with descriptors, the image fetched lazily; errors."""
    import ctypes as C
    from eorb_slam_amd import frontend as fe, _lib
    W, H = 240, 180
    mx, my = synth.undistort_lut(W, H)
    c = fe.Context(); c2 = fe.Context()
    ge = fe.ORBextractor(400, 1.0, 1, 0, 0, 9, imSize=(W, H), ctx=c)
    ge2 = fe.ORBextractor(400, 1.0, 1, 0, 0, 9, imSize=(W, H), ctx=c2)
    fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=c); fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=c2)
    klt = _lib.KltParams(23, 1, 10, 0.03, 1e-4)
    lk = fe.ELK_Tracker(23, 1, 10, 0.03, ctx=c2)
    cap = ge.cap
    for use_raw in (False, True):
        pairs = [synth.shapes_events(2000, W, H, seed=30 + k, motion=0.4 + 0.3 * k, undistort=True, return_raw=True) for k in range(3)]
        evs = [p[1] if use_raw else p[0] for p in pairs]
        kps = np.zeros(cap, synth.KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); oob = np.zeros(cap, np.uint8)
        n = C.c_int(0); mono = C.c_int(0)
        a = (None, fe._p(np.ascontiguousarray(evs[0]))) if use_raw else (fe._p(np.ascontiguousarray(evs[0])), None)
        c.check(c.L.eorb_ev_slice_extract(c.h, a[0], a[1], len(evs[0]), 1.0, 0, 1000, 1, fe._p(kps), fe._p(desc), fe._p(oob), cap, C.byref(n), C.byref(mono), None))
        img_lazy = np.zeros((H, W), np.uint8)
        c.check(c.L.eorb_ev_slice_image(c.h, fe._p(img_lazy)))
        u8 = (fe.EvImConverter.ev2im_gauss_raw if use_raw else fe.EvImConverter.ev2im_gauss)(evs[0], W, H, 1.0, False, True, ctx=c2)
        m2, k2, d2, o2 = ge2(u8, (0, 1000), True)
        assert np.array_equal(img_lazy, u8) and n.value == len(k2) and mono.value == m2
        assert _bits(kps[:n.value], k2) and np.array_equal(desc[:n.value], d2) and np.array_equal(oob[:n.value], o2)
        lk.setRefImage(u8, k2)
        pts = lk.mRefPoints.copy()
        for k in (1, 2):
            st = np.zeros(n.value, np.uint8); er = np.zeros(n.value, np.float32); img = np.zeros((H, W), np.uint8)
            a = (None, fe._p(np.ascontiguousarray(evs[k]))) if use_raw else (fe._p(np.ascontiguousarray(evs[k])), None)
            c.check(c.L.eorb_ev_slice_track(c.h, a[0], a[1], len(evs[k]), 1.0, C.byref(klt), fe._p(pts), fe._p(st), fe._p(er), n.value, fe._p(img)))
            u8k = (fe.EvImConverter.ev2im_gauss_raw if use_raw else fe.EvImConverter.ev2im_gauss)(evs[k], W, H, 1.0, False, True, ctx=c2)
            p2, s2, e2 = lk.trackCurrImage(u8k, lk.mLastTrackedPts)
            lk.mLastTrackedPts = p2
            assert np.array_equal(img, u8k) and _bits(pts, p2) and np.array_equal(st, s2) and _bits(er, e2), (use_raw, k)
        with pytest.raises(fe.EorbError):                       # a point count that is not the reference frame's
            c.check(c.L.eorb_ev_slice_track(c.h, a[0], a[1], len(evs[2]), 1.0, C.byref(klt), fe._p(pts), fe._p(st), fe._p(er), n.value - 1, None))
    fresh = fe.Context()
    fe.ORBextractor(400, 1.0, 1, 0, 0, 9, imSize=(W, H), ctx=fresh)
    with pytest.raises(fe.EorbError):                           # tracking before any reference frame
        fresh.check(fresh.L.eorb_ev_slice_track(fresh.h, fe._p(np.ascontiguousarray(pairs[0][0])), None, 2000, 1.0, C.byref(klt), fe._p(pts), fe._p(st), fe._p(er), n.value, None))
    for x in (c, c2, fresh):
        x.close()
