"""Seeded synthetic stand-ins for the reference's datasets.

Real EvETHZ / MVSEC recordings are not available offline; these generators reproduce their SHAPES:
240x180 DAVIS event slices (integer raw pixels, optionally LUT-undistorted with the EvETHZ
intrinsics of Examples/Event/EvETHZ.yaml:62-70, as the reference's loader does at
src/Event/EventLoader.cpp:111-125 / src/Utils/MyCalibrator.cpp:164-179), textured grey frames and
256-bit descriptor sets.  Pure numpy, deterministic per seed.
"""
import numpy as np

# layout of one event at the drop-in boundary: include/Event/EventData.h:36-58 (24 B, AoS)
EVENT_DTYPE = np.dtype([("ts", "<f8"), ("x", "<f4"), ("y", "<f4"), ("p", "u1"), ("pad", "u1", (7,))])
# eorb_raw_event (16 B): sensor pixel, polarity, timestamp
RAW_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("p", "<u4"), ("t", "<f8")])
# cv::KeyPoint layout (28 B)
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

EVETHZ_K = dict(fx=199.092366542, fy=198.82882047, cx=132.192071378, cy=110.712660011,
                k1=-0.368436311798, k2=0.150947243557, p1=-0.000296130534385, p2=-0.000759431726241)


def undistort_lut(W=240, H=180, K=EVETHZ_K, iters=8):
    """Per-raw-pixel undistorted position (radtan model inverted by fixed-point iteration, then
    re-projected with the same K: what cv::undistortPoints(src, K, D, R=I, P=K) yields)."""
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    x0 = (u - K["cx"]) / K["fx"]; y0 = (v - K["cy"]) / K["fy"]
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icd = 1.0 / (1.0 + (K["k2"] * r2 + K["k1"]) * r2)
        dx = 2 * K["p1"] * x * y + K["p2"] * (r2 + 2 * x * x)
        dy = K["p1"] * (r2 + 2 * y * y) + 2 * K["p2"] * x * y
        x = (x0 - dx) * icd; y = (y0 - dy) * icd
    return (x * K["fx"] + K["cx"]).astype(np.float32), (y * K["fy"] + K["cy"]).astype(np.float32)


def shapes_events(n, W=240, H=180, seed=1, undistort=False, n_poly=3, noise_frac=0.05, t0=0.0, dt=1e-6,
                  motion=1.0, return_raw=False):
    """`n` events from the edges of `n_poly` moving quadrilaterals (K = 4*n_poly = 12 edges by
    default) plus uniform noise.  Raw coordinates are integer pixels; with undistort=True they are
    mapped through the EvETHZ LUT and events leaving the image are dropped and re-drawn, like the
    loader's checkInImage (src/Event/EventLoader.cpp:295-296).  Returns EVENT_DTYPE[n]."""
    rng = np.random.default_rng(seed)
    lut = undistort_lut(W, H) if undistort else None
    out = np.zeros(n, EVENT_DTYPE)
    rawx = np.zeros(n, np.uint16); rawy = np.zeros(n, np.uint16)
    filled = 0
    # polygon vertices at slice start and their displacement over the slice
    ctr = rng.uniform([0.25 * W, 0.25 * H], [0.75 * W, 0.75 * H], size=(n_poly, 1, 2))
    rad = rng.uniform(0.12 * H, 0.33 * H, size=(n_poly, 4, 1))
    ang = np.sort(rng.uniform(0, 2 * np.pi, size=(n_poly, 4)), axis=1)[..., None]
    v0 = ctr + rad * np.concatenate([np.cos(ang), np.sin(ang)], axis=2)
    vel = rng.uniform(-6.0, 6.0, size=(n_poly, 1, 2)) * motion
    while filled < n:
        m = int((n - filled) * 1.15) + 64
        tt = rng.uniform(0, 1, m)
        poly = rng.integers(0, n_poly, m); edge = rng.integers(0, 4, m); s = rng.uniform(0, 1, m)
        a = v0[poly, edge] + vel[poly, 0] * tt[:, None]
        b = v0[poly, (edge + 1) % 4] + vel[poly, 0] * tt[:, None]
        pt = a + (b - a) * s[:, None] + rng.normal(0, 0.6, size=(m, 2))
        noise = rng.uniform(0, 1, m) < noise_frac
        pt[noise] = rng.uniform([0, 0], [W, H], size=(int(noise.sum()), 2))
        xi = np.floor(pt[:, 0]).astype(np.int64); yi = np.floor(pt[:, 1]).astype(np.int64)
        ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        xi, yi, tt = xi[ok], yi[ok], tt[ok]
        if lut is not None:
            xf = lut[0][yi, xi]; yf = lut[1][yi, xi]
            ok = (xf >= 0) & (xf < W) & (yf >= 0) & (yf < H)       # MyCalibrator::isInImage on floats
            xf, yf, tt, xi, yi = xf[ok], yf[ok], tt[ok], xi[ok], yi[ok]
        else:
            xf = xi.astype(np.float32); yf = yi.astype(np.float32)
        k = min(len(xf), n - filled)
        out["x"][filled:filled + k] = xf[:k]; out["y"][filled:filled + k] = yf[:k]
        out["ts"][filled:filled + k] = tt[:k]
        rawx[filled:filled + k] = xi[:k]; rawy[filled:filled + k] = yi[:k]
        filled += k
    order = np.argsort(out["ts"], kind="stable")
    out = out[order]
    out["ts"] = t0 + np.arange(n) * dt                      # monotone, us resolution
    out["p"] = rng.integers(0, 2, n).astype(np.uint8)
    if not return_raw:
        return out
    raw = np.zeros(n, RAW_DTYPE)
    raw["x"] = rawx[order]; raw["y"] = rawy[order]; raw["p"] = out["p"]; raw["t"] = out["ts"]
    return out, raw


def random_raw_events(n, W=240, H=180, seed=0):
    """Uniform sensor-pixel events (eorb_raw_event records)."""
    rng = np.random.default_rng(seed)
    raw = np.zeros(n, RAW_DTYPE)
    raw["x"] = rng.integers(0, W, n); raw["y"] = rng.integers(0, H, n)
    raw["p"] = rng.integers(0, 2, n); raw["t"] = 1e6 + np.arange(n) * 3.0
    return raw


def random_events(n, W=240, H=180, seed=0, frac=True, margin=4.0):
    """Uniform float events, including positions slightly outside the image (stamp clipping)."""
    rng = np.random.default_rng(seed)
    ev = np.zeros(n, EVENT_DTYPE)
    x = rng.uniform(-margin, W + margin, n); y = rng.uniform(-margin, H + margin, n)
    if not frac:
        x = np.floor(x); y = np.floor(y)
    ev["x"] = x.astype(np.float32); ev["y"] = y.astype(np.float32)
    ev["ts"] = np.arange(n) * 1e-6
    ev["p"] = rng.integers(0, 2, n).astype(np.uint8)
    return ev


def texture_image(W=240, H=180, seed=3, octaves=4):
    """Band-limited noise + a few hard-edged rectangles: gives FAST corners on every level."""
    rng = np.random.default_rng(seed)
    img = np.zeros((H, W), np.float64)
    for o in range(octaves):
        gh, gw = max(2, H >> (o + 2)), max(2, W >> (o + 2))
        g = rng.uniform(0, 1, (gh, gw))
        yy = np.linspace(0, gh - 1, H); xx = np.linspace(0, gw - 1, W)
        y0 = np.floor(yy).astype(int); x0 = np.floor(xx).astype(int)
        y1 = np.minimum(y0 + 1, gh - 1); x1 = np.minimum(x0 + 1, gw - 1)
        fy = (yy - y0)[:, None]; fx = (xx - x0)[None, :]
        up = (g[y0][:, x0] * (1 - fy) * (1 - fx) + g[y0][:, x1] * (1 - fy) * fx +
              g[y1][:, x0] * fy * (1 - fx) + g[y1][:, x1] * fy * fx)
        img += up * (0.5 ** (octaves - 1 - o))
    img = (img - img.min()) / (img.max() - img.min())
    for _ in range(40):
        w, h = rng.integers(6, 40), rng.integers(6, 40)
        x, y = rng.integers(0, W - 6), rng.integers(0, H - 6)
        img[y:y + h, x:x + w] = np.clip(img[y:y + h, x:x + w] + rng.uniform(-0.5, 0.5), 0, 1)
    return np.round(img * 255).astype(np.uint8)


def random_descriptors(n, seed=4, width=32):
    return np.random.default_rng(seed).integers(0, 256, (n, width), dtype=np.uint8)


def planted_descriptors(train, seed=5, max_flips=40):
    """Each query = a (permuted) train row with k ~ U[0, max_flips] flipped bits (first 32 B)."""
    rng = np.random.default_rng(seed)
    n = len(train)
    perm = rng.permutation(n)
    q = train[perm].copy()
    for i in range(n):
        k = rng.integers(0, max_flips + 1)
        bits = rng.choice(256, size=k, replace=False)
        for b in bits:
            q[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return q, perm


def random_keypoints(n, W=240, H=180, nlevels=4, scale=1.2, seed=6):
    """Keypoints shaped like extractor output (level-coordinates scaled to the image)."""
    rng = np.random.default_rng(seed)
    kp = np.zeros(n, KP_DTYPE)
    kp["x"] = rng.uniform(10, W - 10, n).astype(np.float32)
    kp["y"] = rng.uniform(10, H - 10, n).astype(np.float32)
    kp["octave"] = rng.integers(0, nlevels, n)
    kp["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    kp["size"] = (31 * scale ** kp["octave"]).astype(np.int32).astype(np.float32)
    kp["response"] = rng.integers(1, 120, n).astype(np.float32)
    kp["class_id"] = -1
    return kp


def random_vocabulary(k=10, L=3, seed=0, ragged=False, stop_frac=0.02):
    """A DBoW2-shaped vocabulary tree with random 256-bit node descriptors: k children per node and L levels (ragged=True:
    2..k children and some branches ending early).  Children of a node are near copies of their parent, so descents are
    meaningful.  Returns dict(L, child_off, child_ids, node_desc, word_id, weight) with node 0 = root."""
    rng = np.random.default_rng(seed)
    desc = [rng.integers(0, 256, 32, dtype=np.uint8)]
    children = [[]]
    level = [0]
    frontier = [0]
    for lv in range(1, L + 1):
        nxt = []
        for u in frontier:
            if ragged and lv > 1 and rng.uniform() < 0.15:
                continue                                                # this branch ends early: u stays a word
            nc = int(rng.integers(2, k + 1)) if ragged else k
            for _ in range(nc):
                flip = rng.uniform(size=256) < (0.25 / lv)
                d = desc[u] ^ np.packbits(flip)
                desc.append(d); children.append([]); level.append(lv)
                children[u].append(len(desc) - 1); nxt.append(len(desc) - 1)
        frontier = nxt
    n = len(desc)
    # DBoW2 stores children in creation order but node ids need not be contiguous per parent: shuffle ids (root stays 0)
    perm = np.concatenate([[0], 1 + rng.permutation(n - 1)])
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    child_off = [0]; child_ids = []
    for new in range(n):
        old = perm[new]
        child_ids.extend(int(inv[c]) for c in children[old]); child_off.append(len(child_ids))
    node_desc = np.stack([desc[perm[i]] for i in range(n)])
    is_leaf = np.array([len(children[perm[i]]) == 0 for i in range(n)])
    word_id = np.full(n, -1, np.int32); word_id[is_leaf] = rng.permutation(int(is_leaf.sum())).astype(np.int32)
    weight = np.zeros(n, np.float64); weight[is_leaf] = rng.uniform(0.5, 9.0, int(is_leaf.sum()))
    stop = is_leaf & (rng.uniform(size=n) < stop_frac)
    weight[stop] = 0.0                                                  # stopWords()
    return dict(L=L, child_off=np.array(child_off, np.int32), child_ids=np.array(child_ids, np.int32), node_desc=node_desc,
                word_id=word_id, weight=weight)


# ---- config 1 as the reference executes it (EvImBuilder::Track): a stream with enough motion for the window-size rule to fire, and
# stand-ins for what the optimisers hand the motion-compensated reconstructions (they are outside the front end) ----
EVETHZ_PINHOLE = (EVETHZ_K["fx"], EVETHZ_K["fy"], EVETHZ_K["cx"], EVETHZ_K["cy"])       # rectified events: a calibrated pinhole camera


def l1_stream(n_chunks=60, chunk=2000, seed=5, motion=14.0, W=240, H=180):
    """One long time-ordered slice of the shapes generator (float EventData, monotone time stamps 1 us apart)."""
    return shapes_events(n_chunks * chunk, W, H, seed=seed, motion=motion, undistort=True)


def l1_mci_poses(window):
    """What resolveLastDPose / resolveLastPoseMap / resolveLastAtt2Params would hand generateMCImage (src/Event/EvImBuilder.cpp:958-1143):
    fixed small motions scaled by the window's length.  Deterministic in the window."""
    n = len(window)
    s = min(n / 6000.0, 2.0)
    return dict(dp=dict(angle=0.010 * s, axis=(0.1, -0.2, 0.97), t=(0.004 * s, -0.003 * s, 0.001), medDepth=1.0),
                ba=dict(angle=0.016 * s, axis=(-0.3, 0.1, 0.95), t=(-0.002 * s, 0.005 * s, 0.0), medDepth=1.3),
                se2=np.array([0.012 * s, 1.5 * s, -0.8 * s], np.float32))


def stereo_pair(seed, W=346, H=260, dmax=14):
    """A rectified stereo pair: the right image is the left one warped by a smooth disparity field in [2, dmax] pixels -- d(x, y) = 2 +
    (dmax - 2) (0.5 + 0.5 sin(y / 37 + x / 91)), bilinear in x, nothing vertical -- plus a little independent noise.  Returns (left, right) u8."""
    rng = np.random.default_rng(seed)
    left = texture_image(W + 32, H, seed=seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    disp = 2.0 + (dmax - 2.0) * (0.5 + 0.5 * np.sin(yy / 37.0 + xx / 91.0))
    xs = xx + disp                                                         # right(x) = left(x + d)
    x0 = np.floor(xs).astype(np.int64); fr = xs - x0
    rows = yy.astype(np.int64)
    right = (1 - fr) * left[rows, x0] + fr * left[rows, np.minimum(x0 + 1, W + 31)]
    right = np.clip(np.rint(right + rng.normal(0, 1.0, right.shape)), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(left[:, :W]), right
