"""The seeded cases behind tests/golden/golden_v2.npz.  `compute(api)` evaluates every case through `api`, an object with the
oracle's function names: the oracle itself (oracle/oracle_py.py) or the GPU adapter in tests/test_golden_v2.py."""
import numpy as np

from eorb_slam_amd import synth

W, H = 240, 180


def events_text(n=3000, seed=3):
    rng = np.random.default_rng(seed)
    ts = np.cumsum(rng.integers(1, 900, n))
    lines = [b"# timestamp x y polarity"]
    for i in range(n):
        lines.append(("%d.%06d %d %d %d" % (ts[i] // 1000000, ts[i] % 1000000, rng.integers(0, W), rng.integers(0, H), rng.integers(0, 2))).encode())
    return b"\n".join(lines) + b"\n"


def two_frames(orc, seed, shift):
    img1 = synth.texture_image(W, H, seed=seed)
    img2 = np.roll(img1, (shift, -shift), axis=(0, 1))
    e = orc.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=19)
    _, k1, d1, _ = e.extract(img1)
    _, k2, d2, _ = e.extract(img2)
    return img1, img2, k1, d1, k2, d2


def oracle_api(oracle):
    """The oracle behind the case API (two cases are compositions of oracle functions)."""
    class Api:
        parse_events_text = staticmethod(oracle.parse_events_text)
        undistort_events = staticmethod(oracle.undistort_events)
        bow_transform = staticmethod(oracle.bow_transform)
        search_by_bow_kf = staticmethod(oracle.search_by_bow_kf)
        search_for_triangulation = staticmethod(oracle.search_for_triangulation)
        distinctive_descriptors = staticmethod(oracle.distinctive_descriptors)
        calc_optical_flow_pyr_lk = staticmethod(oracle.calc_optical_flow_pyr_lk)

        @staticmethod
        def ev2im_gauss_raw(raw, mx, my, W, H, sigma, pol, normalized):
            return oracle.ev2im_gauss(oracle.undistort_events(raw, mx, my, W, H, True, 1.0), W, H, sigma, pol, normalized)

        @staticmethod
        def kf_radius_match(kps, desc, valid, uv, radius, level, qd, inv_sigma2):
            return oracle.kf_radius_match(oracle.Frame(kps, desc, W, H), valid, uv, radius, level, qd, inv_sigma2=inv_sigma2)
    return Api


def compute(api, orc=None):
    """api: the implementation under test; orc: the oracle used to prepare INPUTS (keypoints etc.), defaults to api."""
    orc = orc or api
    g = {}
    mx, my = synth.undistort_lut(W, H)
    mx = np.ascontiguousarray(mx, np.float32); my = np.ascontiguousarray(my, np.float32)
    # loader: text -> raw events -> rectified events -> image (f4)
    raw = api.parse_events_text(events_text())
    g["txt_raw"] = raw.view(np.uint8).reshape(len(raw), 16)
    ev = api.undistort_events(raw, mx, my, W, H, True, 1e6)
    g["txt_rect"] = ev.view(np.uint8).reshape(len(ev), 24)
    f32, u8, mm = api.ev2im_gauss_raw(raw, mx, my, W, H, 1.0, True, True)
    g["raw_f32"], g["raw_u8"] = f32, u8
    # DBoW2 transform (f4)
    voc = synth.random_vocabulary(10, 3, seed=11)
    rng = np.random.default_rng(2)
    leaves = np.nonzero(voc["word_id"] >= 0)[0]
    desc = voc["node_desc"][rng.choice(leaves, 600)].copy()
    desc ^= np.packbits(rng.uniform(size=(600, 256)) < 0.06, axis=1)
    bw, bv, fv, wo, no = api.bow_transform(voc, desc, 1, 0, 1)
    g["bow_word"], g["bow_val"], g["bow_fv_node"], g["bow_fv_off"], g["bow_fv_idx"] = bw, bv, fv[0], fv[1], fv[2]
    # KeyFrame-side matchers (f3) on a frame pair
    img1, img2, k1, d1, k2, d2 = two_frames(orc, 47, 3)
    _, _, fv1, _, _ = orc.bow_transform(voc, d1, 1, 0, 1)
    _, _, fv2, _, _ = orc.bow_transform(voc, d2, 1, 0, 1)
    h1 = (rng.uniform(size=len(k1)) < 0.8).astype(np.uint8); h2 = (rng.uniform(size=len(k2)) < 0.8).astype(np.uint8)
    n, m = api.search_by_bow_kf(k1, d1, h1, fv1, k2, d2, h2, fv2, 0.8, True)
    g["bowkf_n"], g["bowkf_m12"] = np.int32(n), m
    scale = (1.2 ** np.arange(4)).astype(np.float32); sig2 = (scale * scale).astype(np.float32)
    F = (np.array([[0, 0, .7], [0, 0, .7], [-.7, -.7, 0]]) + rng.normal(0, 1e-5, (3, 3))).astype(np.float32)
    n, m = api.search_for_triangulation(k1, d1, 1 - h1, fv1, k2, d2, 1 - h2, fv2, (120.0, 90.0), F, scale, sig2, False, True)
    g["tri_n"], g["tri_m12"] = np.int32(n), m
    M = 800
    pick = rng.integers(0, len(k2), M)
    uv = np.stack([k2["x"][pick] + 3 + rng.normal(0, 1.5, M), k2["y"][pick] - 3 + rng.normal(0, 1.5, M)], axis=1).astype(np.float32)
    level = (k2["octave"][pick] + rng.integers(0, 2, M)).astype(np.int32)
    qd = d2[pick].copy()
    valid = (rng.uniform(size=M) < 0.9).astype(np.uint8)
    sc6 = (1.2 ** np.arange(6)).astype(np.float32)
    radius = (np.float32(3.0) * sc6[level]).astype(np.float32)
    bi, bd = api.kf_radius_match(k1, d1, valid, uv, radius, level, qd, (1.0 / (sc6 * sc6)).astype(np.float32))
    g["rad_idx"], g["rad_dist"] = bi, bd
    sizes = [0, 1, 2, 5, 9, 33, 70]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    dd = rng.integers(0, 256, (offs[-1], 32), dtype=np.uint8)
    g["distinctive"] = api.distinctive_descriptors(dd, offs)
    # pyramidal LK (f2)
    pts = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)[:300]
    np_, st, er = api.calc_optical_flow_pyr_lk(img1, img2, pts, None, 23, 1, 10, 0.03, 0)
    g["klt_pts"], g["klt_status"], g["klt_err"] = np_, st, er
    return g
