#!/usr/bin/env python3
"""Per-row measurements for the SURVEY §8 rows outside the headline pipeline (f1-f4 and the stand-alone matchers): kernel time from
the library's HIP-event profiler (device work only; the host-buffer entry points also pay PCIe copies) next to the CPU oracle
on one core, on representative sizes.  Prints one JSON object; run on the GPU box:  python tests/fuzz/bench_rows.py > rows.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from eorb_slam_amd import frontend as fe, synth  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402

W, H = 240, 180
CAM = (199.092366542, 198.82882047, 132.192071378, 110.712660011)


def timed(fn, reps=3):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps * 1e3


CALLS = []          # per gpu() call, in row order: the median time of one call through ctypes (host buffers in and out), profiler off


def gpu(ctx, fn, reps=5):
    fn()
    ts = []
    for _ in range(15):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    CALLS.append(sorted(ts)[len(ts) // 2] * 1e3)
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(reps):
        fn()
    ctx.sync(); ctx.prof_enable(False)
    r = ctx.prof_results()
    return {k: v[0] / reps for k, v in r.items()}


def main():
    ctx = fe.Context()
    rows = {}
    rng = np.random.default_rng(0)
    E = fe.EvImConverter
    # ---- f1: motion-compensated image of 1 Mev + focus score ----
    ev = synth.shapes_events(1000000, W, H, seed=3, undistort=True)
    axis = np.array([0.12, -0.3, 0.946484]); axis /= np.linalg.norm(axis); t = np.array([0.013, -0.007, 0.002])
    g = gpu(ctx, lambda: E.ev2mci_gg_f_se3(ev, CAM, 0.031, axis, t, 1.7, W, H, ctx=ctx))
    c = timed(lambda: orc.ev2mci_se3(ev, CAM, 0.031, axis, t, 1.7, W, H), 1)
    rows["f1 ev2mci_gg_f SE3, 1 Mev"] = {"gpu_kernels_ms": sum(g.values()), "cpu_oracle_ms": c, "detail": g}
    img = orc.ev2mci_se3(ev, CAM, 0.031, axis, t, 1.7, W, H)[0]
    g = gpu(ctx, lambda: E.measureImageFocus(img, ctx=ctx)); c = timed(lambda: orc.measure_image_focus(img))
    rows["f1 measureImageFocus 240x180"] = {"gpu_kernels_ms": sum(g.values()), "cpu_oracle_ms": c}
    img4 = np.stack([img, img * 0.5, img + 0.25, img * 1.5])
    g = gpu(ctx, lambda: E.measureImageFocusN(img4, ctx=ctx))
    rows["f1 measureImageFocus x 4 in one call (the MCI contest)"] = {"gpu_kernels_ms": sum(g.values()), "cpu_oracle_ms": 4 * c}
    # ---- f4: loader ----
    raw = synth.random_raw_events(1000000, W, H, seed=5)
    mx, my = synth.undistort_lut(W, H)
    E.set_undistort_maps(mx, my, True, ctx=ctx)
    g = gpu(ctx, lambda: E.undistort_events(raw, W, H, 1e6, ctx=ctx)); c = timed(lambda: orc.undistort_events(raw, mx, my, W, H, True, 1e6))
    rows["f4 rectify 1 M raw events"] = {"gpu_kernels_ms": sum(g.values()), "cpu_oracle_ms": c}
    lines = [b"%d.%06d %d %d %d" % (i // 1000, (i * 37) % 1000000, raw["x"][i], raw["y"][i], raw["p"][i]) for i in range(1000000)]
    text = b"\n".join(lines) + b"\n"
    g = gpu(ctx, lambda: E.parse_events_text(text, ctx=ctx), 3); c = timed(lambda: orc.parse_events_text(text), 1)
    rows["f4 parse 1 M text lines (%.1f MB)" % (len(text) / 1e6)] = {"gpu_kernels_ms": sum(g.values()), "cpu_oracle_ms": c}
    voc = synth.random_vocabulary(10, 5, seed=4)                          # 111 111 nodes
    leaves = np.nonzero(voc["word_id"] >= 0)[0]
    desc = voc["node_desc"][rng.choice(leaves, 1000)].copy(); desc ^= np.packbits(rng.uniform(size=(1000, 256)) < 0.05, axis=1)
    V = fe.ORBVocabulary(voc, 0, 1, ctx=ctx)
    g = gpu(ctx, lambda: V.transform(desc, 3)); c = timed(lambda: orc.bow_transform(voc, desc, 3, 0, 1))
    rows["f4 BoW transform, 1000 features, k=10 L=5"] = {"gpu_kernels_ms": sum(g.values()), "cpu_oracle_ms": c}
    # ---- f2: LK ----
    img1 = synth.texture_image(W, H, seed=21); img2 = np.roll(img1, (2, -3), axis=(0, 1))
    e = orc.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=19)
    _, k1, d1, _ = e.extract(img1); _, k2, d2, _ = e.extract(img2)
    pts = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)[:800]
    trk = fe.ELK_Tracker(23, 1, 10, 0.03, ctx=ctx)
    g = gpu(ctx, lambda: trk.calcOpticalFlowPyrLK(img1, img2, pts)); c = timed(lambda: orc.calc_optical_flow_pyr_lk(img1, img2, pts))
    rows["f2 calcOpticalFlowPyrLK, %d points" % len(pts)] = {"gpu_kernels_ms": sum(g.values()), "cpu_oracle_ms": c}
    # ---- f3: KeyFrame-side matchers on a ~900-keypoint frame pair ----
    _, _, fv1, _, _ = orc.bow_transform(voc, d1, 3, 0, 1); _, _, fv2, _, _ = orc.bow_transform(voc, d2, 3, 0, 1)
    h1 = np.ones(len(k1), np.uint8); h2 = np.ones(len(k2), np.uint8)
    g = gpu(ctx, lambda: fe.SearchByBoW_KF(k1, d1, h1, fv1, k2, d2, h2, fv2, 0.8, True, ctx=ctx))
    c = timed(lambda: orc.search_by_bow_kf(k1, d1, h1, fv1, k2, d2, h2, fv2, 0.8, True))
    rows["f3 SearchByBoW(KF,KF), %d x %d" % (len(k1), len(k2))] = {"gpu_kernels_ms": sum(g.values()), "cpu_oracle_ms": c}
    scale = (1.2 ** np.arange(4)).astype(np.float32); sig2 = scale * scale
    F = np.array([[0, 0, .7], [0, 0, .7], [-.7, -.7, 0]], np.float32)
    g = gpu(ctx, lambda: fe.SearchForTriangulation(k1, d1, h1, fv1, k2, d2, h2, fv2, (120., 90.), F, scale, sig2, False, True, ctx=ctx))
    c = timed(lambda: orc.search_for_triangulation(k1, d1, h1, fv1, k2, d2, h2, fv2, (120., 90.), F, scale, sig2, False, True))
    rows["f3 SearchForTriangulation"] = {"gpu_kernels_ms": sum(g.values()), "cpu_oracle_ms": c}
    M = 2000
    pick = rng.integers(0, len(k2), M)
    uv = np.stack([k2["x"][pick] + 3, k2["y"][pick] - 3], axis=1).astype(np.float32); level = k2["octave"][pick].astype(np.int32)
    sc6 = (1.2 ** np.arange(6)).astype(np.float32); radius = (3.0 * sc6[level]).astype(np.float32); valid = np.ones(M, np.uint8)
    gb = fe.grid_bounds(W, H); F1 = orc.Frame(k1, d1, W, H)
    g = gpu(ctx, lambda: fe.KeyFrameRadiusMatch(k1, d1, gb, valid, uv, radius, level, d2[pick], ctx=ctx))
    c = timed(lambda: orc.kf_radius_match(F1, valid, uv, radius, level, d2[pick]))
    rows["f3 Fuse / SearchBySim3 core, %d map points" % M] = {"gpu_kernels_ms": sum(g.values()), "cpu_oracle_ms": c}
    offs = (np.arange(2001) * 12).astype(np.int32); dd = rng.integers(0, 256, (offs[-1], 32), dtype=np.uint8)
    g = gpu(ctx, lambda: fe.ComputeDistinctiveDescriptors(dd, offs, ctx=ctx)); c = timed(lambda: orc.distinctive_descriptors(dd, offs))
    rows["f3 ComputeDistinctiveDescriptors, 2000 map points x 12"] = {"gpu_kernels_ms": sum(g.values()), "cpu_oracle_ms": c}
    # ---- a15 / C4: brute force 2000 x 2000 ----
    tdesc = synth.random_descriptors(2000, seed=4); q, _ = synth.planted_descriptors(tdesc, seed=5)
    bf = fe.BFMatcher(ctx)
    g = gpu(ctx, lambda: bf.knnMatch2(q, tdesc)); c = timed(lambda: orc.bf_knn2(q, tdesc), 1)
    rows["a15 BFMatcher knn2 %d x %d" % (len(q), len(tdesc))] = {"gpu_kernels_ms": sum(g.values()), "cpu_oracle_ms": c}
    # ---- f5: a stereo frame (EuRoC shape: 752x480, 1 200 features, 8 levels): both extractions + ComputeStereoMatches ----
    SW, SH = 752, 480
    left = synth.texture_image(SW + 32, SH, seed=9)
    right = np.ascontiguousarray(left[:, 9:9 + SW]); left = np.ascontiguousarray(left[:, :SW])
    ge = fe.ORBextractor(1200, 1.2, 8, 20, 7, 19, (SW, SH))
    oL = orc.OrbExtractor(1200, 1.2, 8, 20, 7, edgeTh=19, imWidth=SW); oR = orc.OrbExtractor(1200, 1.2, 8, 20, 7, edgeTh=19, imWidth=SW)

    def cpu_stereo():
        _, kl, dl, _ = oL.extract(left, (0, 0)); _, kr, dr, _ = oR.extract(right, (0, 0))
        return oL.compute_stereo_matches(oR, kl, dl, kr, dr, 0.11, 47.9)
    g = gpu(ge.ctx, lambda: ge.stereo(left, right, 0.11, 47.9)); c = timed(cpu_stereo, 1)
    t_call = timed(lambda: ge.stereo(left, right, 0.11, 47.9), 20)
    _, kl, dl, _ = oL.extract(left, (0, 0)); _, kr, dr, _ = oR.extract(right, (0, 0))
    c_match = timed(lambda: oL.compute_stereo_matches(oR, kl, dl, kr, dr, 0.11, 47.9), 3)
    rows["f5 stereo frame 752x480: 2 x extract + ComputeStereoMatches"] = {"gpu_kernels_ms": sum(g.values()), "cpu_oracle_ms": c, "detail": g, "call_ms_through_ctypes": t_call,
                                                                             "cpu_oracle_ms_stereo_matches_alone": c_match}
    ge.ctx.close()
    assert len(CALLS) == len(rows)
    for (k, v), t in zip(rows.items(), CALLS):
        v["speedup"] = v["cpu_oracle_ms"] / max(v["gpu_kernels_ms"], 1e-9)
        v.setdefault("call_ms_through_ctypes", t)
        v["speedup_per_call"] = v["cpu_oracle_ms"] / max(v["call_ms_through_ctypes"], 1e-9)
    print(json.dumps(rows, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
