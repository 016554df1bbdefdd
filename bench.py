#!/usr/bin/env python3
"""bench.py -- front-end frames/s on MI355X (BASELINE.json metric).

Default (`--workload w2`, the driver's line): one "step" = one pass of the hot path (event accumulation -> ORB-1000 extraction ->
frame-to-frame Hamming matching) over one batch of `--batch` synthetic time-slices of 1 000 000 events on a 240x180 sensor
(BASELINE.json configs[1]), all inputs resident in HBM before the timed region.  Rank r of N works on its own independent slices
(weak scaling, no data-path collective); the per-slice keypoint records are gathered to rank 0 over RCCL at the end of every step
(north_star's "final keypoint gather").  `python bench.py --gpus N` with N > 1 starts its own N ranks as child processes when it
was not launched by torch.distributed.run (WORLD_SIZE unset).

Other workloads (each prints the same kind of JSON line with `roofline` and `cpu_baseline`):
  w1  the reference's live operating point: 2 000-event L1 slices (6 000 with --events 6000) -> event image -> FAST detection of
      400 (800) points, 1 level, edge 9 -- batched throughput plus the single-slice latency of the two seams as the reference
      calls them (host buffers, one slice per call)
  w1full  config 1 as EvImBuilder::Track executes it: a stream of chunks through ev2im_gauss -> detect (INIT) / LK against the
      reference frame (TRACKING) -> window-size rule -> on a dispatch the four-way reconstruction contest + L2 detection, one chunk per
      call through the one-call seams, the CPU chain beside it
  w3  texture frames: ORB-1000 + 500 AKAZE-like rows, mixed SearchForInitialization + SearchByProjection, one frame per call
  w4  346x260, 8 levels, 2 000 features + brute-force 2-NN 2000 x 2000, one frame per call

Prints ONE JSON line (rank 0): metric/value/unit + `roofline` for the dominant kernel (live HIP-event timing on the launch stream)
+ `cpu_baseline` (the CPU oracle, -O3 -march=native build, on a bounded sample of the same workload: 1 thread with p50/p95, and
all cores).
"""
import argparse, gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "front-end frames/s (event-accumulate + extract + match) per GPU; HBM GB/s vs roofline"
# HBM bytes and issue-pipe utilisation of the accumulation kernels come from rocprofv3 --pmc passes (bench.py cannot collect PMC counters
# itself): tools/profile_round.sh writes them, with the hash of the library sources they were measured on, to PROFILE_INPUTS; they are
# reported only while that hash equals the hash of the sources this run uses (otherwise null + "stale")
PROFILE_INPUTS = "profiles/roofline_inputs.json"
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8 TB/s (spec)
CPU_POOL = 16              # worker processes of the all-cores CPU baseline (a one-GPU box's CPU share)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["w1", "w2", "w3", "w4", "w1full"], default="w2")
    ap.add_argument("--batch", type=int, default=0, help="time-slices per step per GPU (0 = the workload's default: 128 for w2, 1024 for w1)")
    ap.add_argument("--events", type=int, default=0, help="events per slice (0 = the workload's default: 1 000 000 for w2, 2 000 for w1)")
    ap.add_argument("--cpu-slices", type=int, default=-1, help="slices / frames timed for the 1-thread CPU baseline (0 = skip, -1 = workload default)")
    ap.add_argument("--cpu-pool", type=int, default=CPU_POOL, help="processes of the all-cores CPU baseline (0 = skip)")
    ap.add_argument("--no-prof", action="store_true", help="skip per-kernel HIP-event timing")
    ap.add_argument("--sequences", type=int, default=1,
                    help="independent event sequences processed concurrently per GPU, each on its own context / HIP stream; "
                         "steps alternate between them (a step is still one batch of one sequence)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N > 1: nccl (= RCCL over xGMI); gloo only to rehearse the multi-rank logic "
                         "on a box with fewer GPUs than ranks (records staged through host memory)")
    ap.add_argument("--input", choices=["raw", "float"], default="raw",
                    help="raw: sensor-pixel events (x,y,t,p) + the calibrator's undistortion maps resolved on the GPU; "
                         "float: events already undistorted by the loader (EventData)")
    ap.add_argument("--stream", action="store_true",
                    help="w2: also time the batch with its events arriving from pinned host memory: double-buffered H2D on a copy stream "
                         "overlapped with the previous batch on the compute stream, for the 16-byte and the 4-byte wire record")
    ap.add_argument("--latency-calls", type=int, default=300, help="w1: single-slice calls timed for the latency figures (0 = skip)")
    ap.add_argument("--no-side", action="store_true",
                    help="default w2 line only: skip the `side` object (short verified runs of w1, w3, w4, the float input and the streamed "
                         "input, so that one line carries a number for every BASELINE config)")
    a = ap.parse_args()
    # the `side` object rides on the default line: w2, raw input, one GPU, one sequence
    a.side = (not a.no_side) and a.workload == "w2" and a.input == "raw" and a.gpus == 1 and a.sequences == 1 and not a.no_prof
    if a.side:
        a.stream = True
    return a


# --------------------------------------------------------------------------------------------------------------------------
# CPU baseline helpers (the oracle is the checker / baseline only: never on the measured GPU path)
def _pcts(ts):
    a = np.sort(np.asarray(ts, np.float64)) * 1e3
    return {"mean_ms": float(a.mean()), "p50_ms": float(np.percentile(a, 50)), "p95_ms": float(np.percentile(a, 95))}


class _Unit:
    """One sample unit of a workload on the CPU oracle: run(i) processes unit i (one slice / one frame) and returns its outputs (what
    the GPU run is verified against), reset() forgets the frame-to-frame state."""
    def __init__(self, run, reset):
        self.run, self.reset = run, reset


def _cpu_batch_unit(spec, data=None):
    """data: the (float events, raw events) pairs of a batch workload when the caller already holds them."""
    from oracle import oracle_py
    from eorb_slam_amd import synth
    kind = spec["kind"]
    if kind == "batch":
        W, H, NEV, orb = spec["W"], spec["H"], spec["events"], spec["orb"]
        oe = oracle_py.OrbExtractor(fast=True, imWidth=W, **orb)
        use_raw = spec["input"] == "raw"
        lx, ly = synth.undistort_lut(W, H)
        if data is None:
            data = [synth.shapes_events(NEV, W, H, seed=spec["seed0"] + b, motion=0.5, undistort=True, return_raw=True) for b in range(spec["nunits"])]
        state = {"prev": None}

        def run(i):
            f, r = data[i % len(data)]
            evs = oracle_py.undistort_events(r, lx, ly, W, H, True, 1.0) if use_raw else f
            f32, u8, mm = oracle_py.ev2im_gauss(evs, W, H, spec["sigma"], False, True, fast=True)
            _, kps, desc, _ = oe.extract(u8, (0, 1000), bool(spec["want_desc"]))
            out = {"f32": f32, "u8": u8, "mm": np.asarray(mm, np.float32), "kps": kps, "desc": desc, "nm": None, "m12": None}
            if spec["match"] and spec["want_desc"]:
                F = oracle_py.Frame(kps, desc, W, H, fast=True)
                if state["prev"] is not None:
                    pm = np.stack([state["prev"].kps["x"], state["prev"].kps["y"]], axis=1)
                    out["nm"], out["m12"], _ = oracle_py.search_for_initialization(state["prev"], F, pm, 100, 0.9, True)
                state["prev"] = F
            return out
        return _Unit(run, lambda: state.update(prev=None))
    if kind == "w3":
        frames = _w3_frames(spec["nunits"], spec["seed0"])
        oe = oracle_py.OrbExtractor(fast=True, imWidth=240, **spec["orb"])
        st = {"prev": None}

        def run(i):
            img = frames[i % len(frames)]
            _, kps, desc, _ = oe.extract(img)
            k, d, o = _w3_mix(kps, desc, i % len(frames))
            F = oracle_py.Frame(k, d, 240, 180, o, fast=True)
            out = {"kps": kps, "desc": desc, "n1": None, "m12": None, "n2": None, "cur_mp": None}
            if st["prev"] is not None:
                P, Pd = st["prev"]
                pm = np.stack([P.kps["x"], P.kps["y"]], axis=1)
                out["n1"], out["m12"], _ = oracle_py.search_for_initialization(P, F, pm, 100, 0.9, True)
                a = _w3_proj_args(P.kps, Pd, P.is_orb, len(k))
                out["n2"], out["cur_mp"] = oracle_py.search_by_projection_last(F, P, a["valid"], a["uv"], a["mp_desc"], a["mp_obs"], a["cur_mp"], 15.0, a["ls"], 0, True)
            st["prev"] = (F, d)
            return out
        return _Unit(run, lambda: st.update(prev=None))
    if kind == "w4":
        frames = _w4_frames(spec["nunits"], spec["seed0"])
        oe = oracle_py.OrbExtractor(fast=True, imWidth=346, **spec["orb"])
        q, t = _w4_desc()

        def run(i):
            _, kps, desc, _ = oe.extract(frames[i % len(frames)])
            idx, dist = oracle_py.bf_knn2(q, t, fast=True)
            return {"kps": kps, "desc": desc, "idx": idx, "dist": dist}
        return _Unit(run, lambda: None)
    raise ValueError(kind)


def _cpu_worker(args):
    """All-cores baseline: one process = one core working through its own units; returns (units, t_start, t_end)."""
    spec, idx, reps, barrier = args
    spec = dict(spec); spec["seed0"] = spec["seed0"] + 100 * (idx + 1)
    run = _cpu_batch_unit(spec).run
    run(0)                                               # warm-up (page cache, branch predictors)
    barrier.wait(timeout=600)                            # all workers start their timed loop together
    t0 = time.time()
    for i in range(reps):
        run(i)
    return reps, t0, time.time()


def cpu_baseline(spec, n1, pool, unit_name, data=None, keep=0):
    """1-thread throughput + p50/p95 per unit, and the all-cores throughput (one unit per core at a time).  Returns (the `cpu_baseline`
    object, the oracle's outputs of the first `keep` units: what the GPU's results for the same units are verified against)."""
    out, kept = {}, []
    if n1 > 0:
        unit = _cpu_batch_unit(spec, data)
        unit.run(0)
        unit.reset()                                     # (the timed pass starts like the GPU run: no previous frame)
        ts = []
        t_all = time.perf_counter()
        for i in range(n1):
            t0 = time.perf_counter(); r = unit.run(i); ts.append(time.perf_counter() - t0)
            if i < keep:
                kept.append(r)
        tcpu = time.perf_counter() - t_all
        out = {"value": n1 / tcpu, "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "%d %s (same generator / seeds as the GPU run), oracle built -O3 -march=native -ffp-contract=off, %.1f s"
                         % (n1, unit_name, tcpu), "host_cpus": os.cpu_count()}
        out.update(_pcts(ts))
    if pool > 0 and n1 > 0:
        import multiprocessing as mp
        nproc = min(pool, os.cpu_count() or 1)
        big = spec["kind"] == "batch" and spec["events"] >= 100000
        sp = dict(spec); sp["nunits"] = 4 if big else min(spec["nunits"], 16)
        reps = int(max(8, min(4000, round(2.5 / max(tcpu / n1, 1e-6)))))        # about 2.5 s of wall time per process
        ctx = mp.get_context("spawn")                    # never fork a process that holds a HIP context
        with ctx.Manager() as mgr:
            bar = mgr.Barrier(nproc)
            with ctx.Pool(nproc) as p:
                res = p.map(_cpu_worker, [(sp, i, reps, bar) for i in range(nproc)])
        units = sum(r[0] for r in res)
        wall = max(r[2] for r in res) - min(r[1] for r in res)
        out["all_cores"] = {"value": units / wall, "unit": "frames/s", "cores": nproc,
                            "sample": "%d processes x %d %s each (one unit per core at a time), wall %.1f s" % (nproc, reps, unit_name, wall)}
    return out, kept


def _bits_equal(a, b):
    """same bytes (structured records against their byte rows included)"""
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.nbytes == b.nbytes and np.array_equal(a.view(np.uint8).ravel(), b.view(np.uint8).ravel())


def _verified(checked, mismatches, units, what):
    return {"units": units, "mismatches": len(mismatches), "compared": checked, "what": what,
            **({"first_mismatches": mismatches[:8]} if mismatches else {})}


def _gen_slice(args):
    """One synthetic slice (a worker of the generator pool): the 16-byte HBM record of the chosen input."""
    NEV, W, H, seed, use_raw = args
    from eorb_slam_amd import synth
    f, r = synth.shapes_events(NEV, W, H, seed=seed, motion=0.5, undistort=True, return_raw=True)
    if use_raw:
        return r
    from eorb_slam_amd import frontend
    return frontend.pack_events(f)


def unpack_events(ev16):
    """eorb_event16 records -> the reference's EventData (what the oracle takes): polarity back out of t's sign bit"""
    from eorb_slam_amd import synth
    out = np.zeros(len(ev16), synth.EVENT_DTYPE)
    out["x"] = ev16["x"]; out["y"] = ev16["y"]
    tb = ev16["t"].view(np.uint64)
    out["p"] = ((tb >> np.uint64(63)) == 0).astype(np.uint8)
    out["ts"] = (tb & np.uint64(0x7fffffffffffffff)).view(np.float64)
    return out


def raw_to_float16(raw, lx, ly):
    """eorb_raw_event -> eorb_event16 of the same events: the loader's undistortion (EventLoader.cpp:111-125: the maps at the sensor
    pixel; the generator only keeps events that stay inside the image) and eorb_pack_events' polarity-in-the-sign-bit"""
    from eorb_slam_amd import frontend
    out = np.zeros(len(raw), frontend.EV16_DTYPE)
    yi = raw["y"].astype(np.intp); xi = raw["x"].astype(np.intp)
    out["x"] = lx[yi, xi]; out["y"] = ly[yi, xi]
    tb = raw["t"].astype(np.float64).view(np.uint64).copy()
    tb[raw["p"] == 0] |= np.uint64(0x8000000000000000)
    out["t"] = tb.view(np.float64)
    return out


def gen_slices(NEV, W, H, seeds, use_raw, world):
    """The batch's slices as 16-byte records, one array per seed.  Large batches (the 128 x 1 M events of w2 take minutes on one core)
    are drawn by a pool of processes -- same seeds, same events as the serial loop -- started before this process touches the GPU
    any further; N > 1 ranks share the host's cores."""
    jobs = [(NEV, W, H, int(sd), use_raw) for sd in seeds]
    # (under rocprofv3 the tool library is preloaded into every child process and initialises the GPU there: no pool then)
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or bool(os.environ.get("ROCP_TOOL_LIBRARIES"))
    if NEV * len(seeds) < 8000000 or profiled:
        out = []
        for i, j in enumerate(jobs):
            out.append(_gen_slice(j))
            if profiled and i % 16 == 15:
                print("bench.py: drew %d of %d slices" % (i + 1, len(jobs)), file=sys.stderr, flush=True)
        return out
    import multiprocessing as mp
    nproc = max(2, min(CPU_POOL, (os.cpu_count() or 2) // max(world, 1), len(jobs)))
    with mp.get_context("spawn").Pool(nproc) as p:
        return p.map(_gen_slice, jobs, chunksize=1)


# --------------------------------------------------------------------------------------------------------------------------
# synthetic inputs of w3 / w4
def _w3_frames(n, seed0):
    from eorb_slam_amd import synth
    base = synth.texture_image(240, 180, seed=seed0)
    return [np.ascontiguousarray(np.roll(base, (i % 7, -(2 * i % 11)), axis=(0, 1))) for i in range(n)]


_W3_CACHE = {}        # the synthetic parts of a frame (random rows, noise) are drawn once per frame index / size, not inside the timed calls


def _w3_mix(kps, desc, i):
    """MixedFrame stand-in: the extractor's ORB keypoints + 500 AKAZE-like rows (61 bytes, the first 32 compared)."""
    from eorb_slam_amd import synth
    n, na = len(kps), 500
    key = ("mix", i % 64)
    if key not in _W3_CACHE:
        rng = np.random.default_rng(1000 + i % 64)
        ak = synth.random_keypoints(na, 240, 180, nlevels=3, scale=1.26, seed=2000 + i % 64)
        ak["class_id"] = 0
        _W3_CACHE[key] = (ak, rng.integers(0, 256, (na, 61)).astype(np.uint8))
    ak, ad = _W3_CACHE[key]
    k = np.concatenate([kps, ak])
    d = np.zeros((n + na, 61), np.uint8); d[:n, :32] = desc; d[n:] = ad
    o = np.concatenate([np.ones(n, np.uint8), np.zeros(na, np.uint8)])
    return k, d, o


def _w3_proj_args(pk, pd, po, ncur):
    n = len(pk)
    key = ("proj", n)
    if key not in _W3_CACHE:
        rng = np.random.default_rng(7)
        _W3_CACHE[key] = ((rng.uniform(size=n) < 0.8).astype(np.uint8), rng.normal(0, 1, n), rng.normal(0, 1, n))
    valid, nx, ny = _W3_CACHE[key]
    sf = np.float32(1.2) ** np.arange(4, dtype=np.float32)
    asf = np.float32(1.26) ** np.arange(4, dtype=np.float32)
    octv = np.clip(pk["octave"], 0, 3)
    return dict(valid=valid,
                uv=np.stack([pk["x"] + nx, pk["y"] + ny], axis=1).astype(np.float32),
                mp_desc=np.ascontiguousarray(pd[:, :32]), mp_obs=np.ones(n, np.uint8),
                cur_mp=np.full(ncur, -1, np.int32), ls=np.where(po == 1, sf[octv], asf[octv]).astype(np.float32))


def _w4_frames(n, seed0):
    from eorb_slam_amd import synth
    base = synth.texture_image(346, 260, seed=seed0)
    return [np.ascontiguousarray(np.roll(base, (i % 5, -(3 * i % 13)), axis=(0, 1))) for i in range(n)]


def _w4_desc():
    from eorb_slam_amd import synth
    t = synth.random_descriptors(2000, seed=4)
    return synth.planted_descriptors(t, seed=5)[0], t


# --------------------------------------------------------------------------------------------------------------------------
def source_hash():
    """Hash of the library sources (the same function as tools/roofline_inputs.py)."""
    import glob, hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "eorb_slam_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "eorb_slam_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "eorb_fe.h")])
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def profile_inputs(key):
    """The rocprofv3-derived entry of this workload (tools/profile_round.sh), or (None, reason)."""
    path = os.path.join(ROOT, PROFILE_INPUTS)
    try:
        doc = json.load(open(path))
    except Exception:
        return None, "no %s" % PROFILE_INPUTS
    if doc.get("src_hash") != source_hash():
        return None, "stale: %s was measured on sources %s, this run uses %s" % (PROFILE_INPUTS, doc.get("src_hash"), source_hash())
    e = doc.get("entries", {}).get(key)
    return (e, "%s (sources %s)" % (PROFILE_INPUTS, doc["src_hash"])) if e else (None, "%s has no entry %s" % (PROFILE_INPUTS, key))


# HIP-event scope -> the kernel it brackets (one launch per scope where a single name is given)
SCOPE_KERNEL = {"ev_gather": "eorb::sl_gather_kernel", "ev_count": "eorb::sl_count_lds_kernel", "ev_scan": "eorb::sl_scan_kernel",
                "ev_scatter": "eorb::sl_scatter_rank_kernel", "ev_dedupe": "eorb::dd_insert_kernel", "orb_octree": "eorb::octree_kernel",
                "orb_fast_cells": "eorb::fast_cells_kernel", "orb_blur": "eorb::blur_kernel", "orb_brief": "eorb::brief_kernel",
                "orb_orient": "eorb::orient_kernel", "orb_assemble": "eorb::assemble_kernel", "bf_knn2": "eorb::bf_knn2_kernel",
                "klt_track": "eorb::klt_track_kernel (+ pyramid kernels)", "search_init": "eorb::win_cand_kernel<0> + eorb::win_resolve_kernel<0>",
                "search_proj_last": "eorb::win_cand_kernel<1> + eorb::win_resolve_kernel<1>", "orb_pyr": "eorb::pyr_level0_kernel + eorb::pyr_resize_kernel"}
GPU_CLOCK_HZ = 2.4e9       # MI355X_MICROARCH.md: peak engine clock
N_CU = 256


def issue_floor(entries, hot_entries, measured_gather_ms):
    """What the two gather kernels' binding pipes allow for the entries they walked: the LDS-row gather reads one 256-byte
    row per entry = 2 LDS-array cycles per entry and CU (ds_read_b32 of 64 lanes at 128 B/clk; tools/mb/slot_loop.hip measures 2.05);
    the register-row kernel issues per 16-bit entry one scalar instruction (s_mov_b32 / s_lshr_b32 to M0; + 6 per 64 entries of loop) on the
    CU's one scalar ALU and one 64-lane v_add_f32 = 4 cycles of one of the CU's four vector ALUs: both floors are 1 cycle per entry and CU."""
    lds_entries = max(entries - hot_entries, 0)
    f_lds = lds_entries * 2.0 / (N_CU * GPU_CLOCK_HZ) * 1e3
    salu_per_entry = 1.0 + 6.0 / 64.0
    f_salu = hot_entries * salu_per_entry / (N_CU * GPU_CLOCK_HZ) * 1e3
    f_valu = hot_entries * 4.0 / (4 * N_CU * GPU_CLOCK_HZ) * 1e3
    return {"entries_per_step": int(entries), "register_row_entries": int(hot_entries), "lds_row_entries": int(lds_entries),
            "sl_gather_kernel": {"lds_cycles_per_entry": 2, "floor_ms": f_lds},
            "sl_hot_kernel": {"salu_instructions_per_entry": salu_per_entry, "salu_floor_ms": f_salu, "valu_cycles_per_entry_and_simd": 4, "valu_floor_ms": f_valu,
                              "floor_ms": max(f_salu, f_valu)},
            "floor_ms": max(f_lds, f_salu, f_valu), "measured_ms": measured_gather_ms, "clock_hz": GPU_CLOCK_HZ,
            "note": "the two kernels run side by side (LDS array / scalar + vector ALU): the stage cannot beat the larger floor; measured = the live launch time of sl_gather_kernel, "
                    "which the register-row kernel overlaps; floors at the peak engine clock (the counters' GRBM_GUI_ACTIVE puts the sustained clock near 1.9 GHz: x 1.26)"}


def roofline_of(prof, steps, unit_bytes_by_kernel, units_per_launch, inputs_key=None):
    """`roofline` of the dominant kernel: algorithmic bytes per launch / its live HIP-event launch time; HBM traffic and the
    issue-pipe utilisation of that kernel from the rocprofv3 passes of the same sources (see PROFILE_INPUTS)."""
    tot = {k: v[0] for k, v in prof.items()}
    dom = max(tot, key=tot.get)                     # (slot form: every accumulation scope brackets ONE kernel's launch)
    ms, launches = prof[dom]
    avg_ms = ms / max(launches, 1)
    ub = unit_bytes_by_kernel(dom)
    achieved = ub * units_per_launch / (avg_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "kernel": SCOPE_KERNEL.get(dom, dom), "scope": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": None, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": ub * units_per_launch}
    if inputs_key:
        e, src = profile_inputs(inputs_key)
        r["traffic_source"] = src
        sc = e["scopes"].get(dom) if e else None
        if sc:
            if sc.get("kernels"):                   # (the scope's longest kernel in the rocprofv3 trace of the same sources: the name the profile files use)
                r["kernel"] = max(sc["kernels"].items(), key=lambda kv: kv[1].get("avg_ms", 0.0))[0]
            r["traffic"] = sc["traffic_bytes"]
            r["rocprof_avg_launch_ms"] = sc["rocprof_ms"]
            if "issue" in sc:
                i = sc["issue"]
                r["issue"] = {"valu_frac": i["valu_frac"], "lds_frac": i["lds_frac"], "salu_frac": i.get("salu_frac"), "wait_frac": i["wait_frac"],
                              "issue_stall_frac": i["issue_stall_frac"], "kernel": i["kernel"],
                              "source": "rocprofv3 --pmc, cycles = GRBM_GUI_ACTIVE / 8 XCDs: SQ_INSTS_VALU x 4 / (cycles x 1024 SIMDs), SQ_LDS_IDX_ACTIVE / (cycles x 256 CUs), SQ_INSTS_SALU / (cycles x 256 CUs), "
                                        "SQ_WAIT_ANY / SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; " + src}
            stage = sum(v["traffic_bytes"] for k, v in e["scopes"].items() if k.startswith("ev_"))
            alg = unit_bytes_by_kernel("ev_") * units_per_launch
            if stage > 0 and dom.startswith("ev_"):
              r["stage"] = {"what": "all accumulation kernels (ev_*) of a step", "traffic": stage, "algorithmic": alg, "ratio": stage / alg,
                          "note": "of this, ~1.5 GB per 128 Mev are the count pass's ranked records (8 B per event written once, read once by the scatter, which then neither "
                                  "counts nor ranks: 3 % more frames/s); EORB_SLOT_PRERANK=0 trades them back: 2.33 x algorithmic, the scatter ranking by itself",
                          "scopes": {k: {"live_ms": prof[k][0] / max(prof[k][1], 1), "rocprof_ms": v["rocprof_ms"], "traffic": v["traffic_bytes"],
                                         "algorithmic_GBps": alg / (prof[k][0] / max(prof[k][1], 1) * 1e-3) / 1e9,
                                         **({"issue_by_kernel": v["issue_by_kernel"]} if "issue_by_kernel" in v else {}),
                                         **({"issue": {kk: vv for kk, vv in v["issue"].items() if kk != "counters"}} if "issue" in v else {})}
                                     for k, v in e["scopes"].items() if k.startswith("ev_") and k in prof}}
    return r, {k: v[0] / steps for k, v in sorted(prof.items())}


def level_pixels(W, H, sf, nlevels):
    return sum(int(round(W / sf ** l)) * int(round(H / sf ** l)) for l in range(nlevels))


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started from a bare shell: launch one child process per rank BEFORE anything in this process touches the GPU
        from eorb_slam_amd import shard
        sys.exit(shard.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus))
    if a.gpus != world:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world), file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    from eorb_slam_amd import frontend, shard, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the front end has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    local_rank %= max(torch.cuda.device_count(), 1)     # (a rehearsal with more ranks than GPUs shares devices; no effect on a full node)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            import datetime
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=600))
        else:
            import datetime
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
    env = dict(a=a, world=world, rank=rank, local_rank=local_rank, dev=dev, torch=torch, dist=dist, frontend=frontend, shard=shard,
               synth=synth)
    out = {"w1": run_batch, "w2": run_batch, "w3": run_frames, "w4": run_frames, "w1full": run_chain}[a.workload](env)
    failed = 0
    if rank == 0:
        if a.side and world == 1:
            out["side"] = side_runs(env, out)
        failed = int(out.get("verified", {}).get("mismatches", 0)) + sum(int((v.get("verified") or {}).get("mismatches", 0)) for v in out.get("side", {}).values() if isinstance(v, dict))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if failed:
        print("bench.py: %d outputs of the timed path differ from the CPU oracle's (see `verified`)" % failed, file=sys.stderr)
        sys.exit(3)


def _compact(o):
    """A side run's line, reduced to what a reader needs to place it: value, the workload, the dominant kernel's roofline, the CPU
    sample and the verification."""
    if not o:
        return None
    r = o.get("roofline") or {}
    c = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"], "warmup": o["warmup"],
         "workload": o["config"]["workload"], "config": {k: v for k, v in o["config"].items() if k != "workload"},
         "roofline": {k: r.get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms", "traffic", "traffic_source") if k in r} or None,
         "cpu_baseline": {k: o["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "sample", "p50_ms") if k in o.get("cpu_baseline", {})} or None,
         "verified": o.get("verified")}
    for k in ("latency", "batched", "speedup_vs_cpu_1thread", "kernels_ms_per_step"):
        if k in o:
            c[k] = o[k]
    return c


def run_stereo(env):
    """A stereo frame per call (the stereo constructor's hot path, src/Frame.cc:97-152 + :869-1048): both extractions and
    ComputeStereoMatches through eorb_frame_stereo, host buffers in and out, on a synthetic rectified pair of the EuRoC shape (752x480, 1 200
    features, 8 levels, FAST 20 / 7: Examples/Stereo/EuRoC.yaml); verified against the oracle; the oracle timed on one core beside it."""
    from eorb_slam_amd import frontend as fe, synth
    from oracle import oracle_py as orc
    a = env["a"]
    W, H, NF, NL, TH, MB, MBF = 752, 480, 1200, 8, (20, 7), 0.11, 47.9
    pairs = [synth.stereo_pair(200 + k, W, H) for k in range(4)]
    ge = fe.ORBextractor(NF, 1.2, NL, TH[0], TH[1], 19, (W, H))
    for k in range(max(a.warmup, 1) * 2):
        ge.stereo(pairs[k % 4][0], pairs[k % 4][1], MB, MBF)
    calls = max(a.steps, 1) * 8
    ts = []
    for k in range(calls):
        t0 = time.perf_counter(); ge.stereo(pairs[k % 4][0], pairs[k % 4][1], MB, MBF); ts.append(time.perf_counter() - t0)
    ts = np.array(ts)
    # verification + the CPU sample: every pair once through the oracle
    oL = orc.OrbExtractor(NF, 1.2, NL, TH[0], TH[1], edgeTh=19, imWidth=W); oR = orc.OrbExtractor(NF, 1.2, NL, TH[0], TH[1], edgeTh=19, imWidth=W)
    bad = 0; cpu = []; matched = 0
    for L, R in pairs:
        t0 = time.perf_counter()
        _, kl, dl, _ = oL.extract(L, (0, 0)); _, kr, dr, _ = oR.extract(R, (0, 0))
        ur, dp, nm = oL.compute_stereo_matches(oR, kl, dl, kr, dr, MB, MBF)
        cpu.append(time.perf_counter() - t0)
        g = ge.stereo(L, R, MB, MBF)
        same = (len(g["kpsL"]) == len(kl) and len(g["kpsR"]) == len(kr) and g["nmatches"] == nm and _bits_equal(g["kpsL"], kl) and _bits_equal(g["kpsR"], kr)
                and _bits_equal(g["descL"], dl) and _bits_equal(g["descR"], dr) and _bits_equal(g["uRight"], ur) and _bits_equal(g["depth"], dp))
        bad += 0 if same else 1
        matched += int((ur > 0).sum())
    ge.ctx.close()
    return {"value": float(1.0 / ts.mean()), "unit": "stereo frames/s", "calls": int(calls), "p50_ms": float(np.median(ts) * 1e3), "p95_ms": float(np.percentile(ts, 95) * 1e3),
            "workload": "stereo: %dx%d pair -> ORB-%d x 2 (%d levels) -> ComputeStereoMatches, one frame per call through ctypes (eorb_frame_stereo)" % (W, H, NF, NL),
            "mean_stereo_matches": matched / len(pairs),
            "cpu_baseline": {"value": float(1.0 / np.mean(cpu)), "unit": "stereo frames/s", "cores": 1, "kind": "port", "p50_ms": float(np.median(cpu) * 1e3),
                             "sample": "%d pairs through the oracle (2 x extract + compute_stereo_matches)" % len(pairs)},
            "verified": {"units": len(pairs), "mismatches": int(bad), "compared": ["keypoint records of both images", "descriptors of both images", "mvuRight bits", "mvDepth bits", "matches before the median cut"]}}


def side_runs(env, main_out):
    """Short runs of the other BASELINE configs and input forms inside the default command, each with its own CPU sample and its own
    verification against the oracle, so that ONE recorded line carries a number for configs[0]..[3] (w1, w2, w3, w4), for the
    reference's own float EventData input and for events streamed from the host."""
    import argparse
    a = env["a"]
    side = {"what": "short runs of the other workloads / input forms by the same command (fewer steps, a smaller CPU sample, no all-cores "
                    "leg); `python bench.py --workload w1|w3|w4` and `--input float` print their full lines"}
    plan = (("w1", run_batch, dict(workload="w1", input="raw", steps=10, warmup=2, cpu_slices=400, cpu_pool=0, latency_calls=100)),
            ("w1full", run_chain, dict(workload="w1full", steps=3, warmup=1, cpu_slices=1, cpu_pool=0)),
            ("w3", run_frames, dict(workload="w3", steps=6, warmup=1, cpu_slices=16, cpu_pool=0)),
            ("w4", run_frames, dict(workload="w4", steps=6, warmup=1, cpu_slices=8, cpu_pool=0)),
            ("w2_float_input", run_batch, dict(workload="w2", input="float", steps=6, warmup=2, cpu_slices=4, cpu_pool=0)))
    for name, fn, over in plan:
        v = dict(vars(a)); v.update(over); v.update(side=False, stream=False, batch=0, events=0)
        env2 = dict(env); env2["a"] = argparse.Namespace(**v)
        t0 = time.perf_counter()
        try:
            side[name] = _compact(fn(env2))
            side[name]["wall_s"] = time.perf_counter() - t0
        except Exception as e:                                   # a side run must not cost the main line
            side[name] = {"error": "%s: %s" % (type(e).__name__, e), "verified": {"mismatches": 1, "what": "the side run failed"}}
    t0 = time.perf_counter()
    try:
        v = dict(vars(a)); v.update(steps=6, warmup=1)
        env2 = dict(env); env2["a"] = argparse.Namespace(**v)
        side["stereo"] = run_stereo(env2)
        side["stereo"]["wall_s"] = time.perf_counter() - t0
    except Exception as e:
        side["stereo"] = {"error": "%s: %s" % (type(e).__name__, e), "verified": {"mismatches": 1, "what": "the side run failed"}}
    if "streaming" in main_out:
        side["streaming"] = main_out["streaming"]
    return side


def timed_steps(env, step, sync_extra=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
    a, torch, dist, world = env["a"], env["torch"], env["dist"], env["world"]
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    # The interpreter's cyclic collector stays on, but the objects that exist by now (torch's import alone leaves millions) move to the
    # permanent generation: a full collection inside the timed steps otherwise walks all of them -- a 40-60 ms host stall once per run
    # (gpurun_out/gap_float.txt: the GPU idle between two calls), visible wherever a call waits for its results (float input, w3, w4).
    gc.collect(); gc.freeze()
    if sync_extra:
        sync_extra()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        gdev = env["dev"] if a.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([dt], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def base_line(env, value, dt, workload, config):
    a, world = env["a"], env["world"]
    return {"metric": METRIC, "value": value, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": dict(workload=workload, **config)}


def stream_ingest(env, seq, comp_s, ev16, offsets, B, NEV, resident_s):
    """The w2 step with its input streamed: two device buffers, batch i + 1 copied from pinned host memory on a copy stream while
    batch i runs on the compute stream (events order copy -> compute -> reuse of the buffer).  Steady-state slices/s including the
    link, for the 16-byte eorb_raw_event, the 4-byte eorb_raw_event4 (the images never read the time stamp) and the 2-byte
    eorb_raw_event2 (polarity-free images read nothing but the sensor pixel: 43 200 < 65 536 on a DAVIS 240x180)."""
    a, torch, dev, frontend = env["a"], env["torch"], env["dev"], env["frontend"]
    ctx_i, fb_i, bf = seq
    res = {"what": "pinned host buffer -> HBM per batch on a copy stream, overlapped with the previous batch's kernels (double buffer); "
                   "value includes the link", "resident_ms_per_step": resident_s * 1e3}
    for name, rec_bytes, fmt in (("raw16", 16, True), ("raw4", 4, 4), ("raw2", 2, 2)):
        host_np = ev16 if fmt is True else (frontend.pack_raw_events4(ev16) if fmt == 4 else frontend.pack_raw_events2(ev16, 240))
        host = torch.from_numpy(host_np.view(np.uint8)).pin_memory()
        dbuf = [torch.empty(host.numel(), dtype=torch.uint8, device=dev) for _ in range(2)]
        copy_s = torch.cuda.Stream(device=dev)
        copied = [torch.cuda.Event() for _ in range(2)]; freed = [torch.cuda.Event() for _ in range(2)]

        def issue_copy(i):
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(freed[i % 2])
                dbuf[i % 2].copy_(host, non_blocking=True)
                copied[i % 2].record(copy_s)

        def issue_compute(i):
            with torch.cuda.stream(comp_s):
                comp_s.wait_event(copied[i % 2])
                fb_i.run_dev(dbuf[i % 2].data_ptr(), offsets, bf["img"].data_ptr(), bf["kp"].data_ptr(), bf["desc"].data_ptr(), bf["n"].data_ptr(),
                             bf["m"].data_ptr(), bf["nm"].data_ptr(), raw=fmt)
                freed[i % 2].record(comp_s)
        for e in freed:
            e.record(comp_s)
        nsteps = max(4, min(a.steps, 12))
        issue_copy(0)
        for i in range(2):                         # warm-up
            issue_copy(i + 1); issue_compute(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        issue_copy(0)
        for i in range(nsteps):
            if i + 1 < nsteps:
                issue_copy(i + 1)
            issue_compute(i)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / nsteps
        # the link alone
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for i in range(4):
            with torch.cuda.stream(copy_s):
                dbuf[i % 2].copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        tl = (time.perf_counter() - t1) / 4
        res[name] = {"value": B / t, "unit": "frames/s", "ms_per_step": t * 1e3, "record_bytes": rec_bytes, "bytes_per_step": int(host.numel()),
                     "h2d_GBps": host.numel() / tl / 1e9, "h2d_ms_per_step": tl * 1e3}
        del dbuf, host
    return res


# --------------------------------------------------------------------------------------------------------------------------
def run_batch(env):
    """w2 (BASELINE.json configs[1]) and w1 (configs[0] stand-in): the batched HBM-resident pipeline."""
    a, world, rank, dev, torch = env["a"], env["world"], env["rank"], env["dev"], env["torch"]
    frontend, shard, synth = env["frontend"], env["shard"], env["synth"]
    W, H = 240, 180
    if a.workload == "w2":
        B, NEV = a.batch or 128, a.events or 1000000
        orb = dict(nfeatures=1000, scaleFactor=1.2, nlevels=4, iniThFAST=10, minThFAST=0, edgeTh=19)
        want_desc, match = 1, 1
        ncpu = 64 if a.cpu_slices < 0 else a.cpu_slices
    else:
        B, NEV = a.batch or 1024, a.events or 2000
        # Event.fts.* of Examples/Event/EvETHZ.yaml:197-203: FAST detection (ORB extractor forced to 1 level @ 1.0,
        # EvBaseTracker.cpp:157-161), threshold 0, 400 points (L2 window of 6 000 events: 800), Features.imMargin 9
        orb = dict(nfeatures=800 if NEV >= 6000 else 400, scaleFactor=1.0, nlevels=1, iniThFAST=0, minThFAST=0, edgeTh=9)
        want_desc, match = 0, 0
        ncpu = 2000 if a.cpu_slices < 0 else a.cpu_slices
    use_raw = a.input == "raw"
    # ---- synthetic input: B independent slices per rank (seeded), packed to the 16 B HBM record ----
    seed0 = 2 + 1000 * rank
    # N = 1: B distinct slices (the line the driver records).  N > 1: every rank of a node draws its events on the same host, so a
    # rank draws 16 distinct slices and tiles them -- the scaling run measures the GPUs and the gather, not 8 x 128 x 1 M draws
    ndist = B if (world == 1 or NEV * B < 8000000) else min(B, 16)
    cache = env.setdefault("_slices", {})
    key = (a.workload, NEV, B, seed0, ndist)
    if key not in cache:
        cache[key] = gen_slices(NEV, W, H, [seed0 + b for b in range(ndist)], True, world)
    recs = cache[key]                                   # raw sensor records; the float form = the loader's map lookup of the same events
    if not use_raw:
        lx, ly = synth.undistort_lut(W, H)
        recs = [raw_to_float16(r, lx, ly) for r in recs]
    ev16 = np.concatenate([recs[b % ndist] for b in range(B)])
    offsets = np.arange(B + 1, dtype=np.int64) * NEV
    # the CPU baseline / the verification (rank 0 at N = 1) work on the first slices of the batch
    ncmp = min(ncpu, 64, B) if world == 1 else 0
    if use_raw:
        cpu_pairs = [(None, recs[b]) for b in range(ncmp)]
    else:
        cpu_pairs = [(unpack_events(recs[b]), None) for b in range(ncmp)]
    lat_pairs = []
    if a.workload == "w1" and a.latency_calls > 0 and world == 1:
        lat_pairs = [synth.shapes_events(NEV, W, H, seed=seed0 + b, motion=0.5, undistort=True, return_raw=True) for b in range(min(B, 64))]

    S = max(1, a.sequences)
    d_ev = torch.from_numpy(ev16.view(np.uint8)).to(dev)
    # every sequence, the first included, runs on an explicit stream of its own: the front end's launches, the HIP-event timing and
    # (N > 1) the gather are all ordered on it
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    seqs = []
    mx, my = synth.undistort_lut(W, H)
    for si in range(S):
        ctx_i = frontend.Context(device=env["local_rank"], stream=streams[si].cuda_stream)
        fb_i = frontend.FrontEndBatch(W, H, 1.0, False, max_batch=B, max_events=NEV, match=bool(match), want_desc=bool(want_desc),
                                      windowSize=100, nnratio=0.9, checkOri=True, ctx=ctx_i, **orb)
        if use_raw:
            frontend.EvImConverter.set_undistort_maps(mx, my, True, ctx=ctx_i)
        cap = fb_i.cap
        lay = shard.RecordLayout(B, cap)
        rec = lay.alloc(dev)
        n_v, kp_v, desc_v = lay.views(rec)
        bufs = dict(img=torch.empty(B * W * H, dtype=torch.uint8, device=dev), rec=rec, n=n_v, kp=kp_v, desc=desc_v,
                    m=torch.empty(B * cap, dtype=torch.int32, device=dev), nm=torch.zeros(B, dtype=torch.int32, device=dev))
        seqs.append((ctx_i, fb_i, bufs))
    gdev = dev if a.backend == "nccl" else torch.device("cpu")
    recv = [torch.empty(seqs[0][2]["rec"].shape, dtype=torch.uint8, device=gdev) for _ in range(world)] if (world > 1 and rank == 0) else None
    step_no = [0]

    dbg = os.environ.get("EORB_BENCH_DEBUG")

    def step():
        si = step_no[0] % S; step_no[0] += 1
        _, fb_i, bf = seqs[si]
        t_dbg = time.perf_counter()
        with torch.cuda.stream(streams[si]):
            fb_i.run_dev(d_ev.data_ptr(), offsets, bf["img"].data_ptr(), bf["kp"].data_ptr(), bf["desc"].data_ptr(), bf["n"].data_ptr(),
                         bf["m"].data_ptr(), bf["nm"].data_ptr(), raw=use_raw)
            if world > 1:       # final keypoint gather (RCCL over xGMI), fixed-capacity records, on the producing stream
                shard.gather_packed(bf["rec"], 0, recv, a.backend)
        if dbg:
            print("step %d: %.2f ms inside the call" % (step_no[0], (time.perf_counter() - t_dbg) * 1e3), file=sys.stderr)

    # Inside the timed steps only the accumulation scopes are bracketed by HIP events (the roofline's kernels: two event records per
    # scope and call cost a step of 14 scopes 0.1-0.18 ms of 3.6); the other scopes are timed in a few further steps afterwards.
    TIMED_SCOPES = ("ev_count", "ev_scan", "ev_scatter", "ev_bin", "ev_gather", "ev_dedupe")

    def arm_prof():
        if not a.no_prof:
            for c_i, _, _ in seqs:
                c_i.prof_reset(); c_i.prof_only(TIMED_SCOPES); c_i.prof_enable(True)

    def collect_prof():
        out = {}
        for c_i, _, _ in seqs:
            c_i.prof_enable(False)
            for k, (ms, n) in c_i.prof_results().items():
                a0, n0 = out.get(k, (0.0, 0))
                out[k] = (a0 + ms, n0 + n)
        return out

    dt = timed_steps(env, step, arm_prof)
    prof = {}
    if not a.no_prof:
        prof = collect_prof()
        extra = min(a.steps, 5)
        for c_i, _, _ in seqs:
            c_i.prof_reset(); c_i.prof_only(()); c_i.prof_enable(True)
        for _ in range(extra):
            step()
        torch.cuda.synchronize()
        for k, (ms, n) in collect_prof().items():
            if k not in prof:
                prof[k] = (ms * a.steps / extra, n * a.steps // extra)      # (scaled to the timed steps: roofline_of divides by them)
    for c_i, _, _ in seqs:
        c_i.sync()                                   # raises if a batch overflowed an internal capacity (sticky status)
    nk = seqs[0][2]["n"].cpu().numpy(); nm = seqs[0][2]["nm"].cpu().numpy()
    # ---- what the timed steps computed, for the verification below (rank 0, N = 1): the LAST step's outputs of sequence 0 for the
    #      first `ncmp` slices -- float images, extremes, u8 images, keypoints, descriptors, matches -- downloaded outside the timed region
    gpu_out = None
    if rank == 0 and ncmp > 0:
        _, fb0, bf0 = seqs[(step_no[0] - 1) % S]
        cap0 = fb0.cap
        p32, mm = fb0.last_f32(B)
        f32 = np.zeros((ncmp, H, W), np.float32); fb0.ctx.download(f32, p32)
        gpu_out = dict(f32=f32, mm=mm[:ncmp], u8=bf0["img"][:ncmp * W * H].cpu().numpy().reshape(ncmp, H, W), n=bf0["n"][:ncmp].cpu().numpy(),
                       kp=bf0["kp"][:ncmp * cap0 * 28].cpu().numpy().reshape(ncmp, cap0, 28), desc=bf0["desc"][:ncmp * cap0 * 32].cpu().numpy().reshape(ncmp, cap0, 32),
                       m=bf0["m"][:ncmp * cap0].cpu().numpy().reshape(ncmp, cap0), nm=bf0["nm"][:ncmp].cpu().numpy(), cap=cap0)
    out = None
    if rank == 0:
        frames = world * B * a.steps
        if a.workload == "w2":
            wl = ("BASELINE.json configs[1]: synthetic %d events/slice on %dx%d (shapes generator, DAVIS sensor pixels; %s), "
                  "ev2im_gauss sigma=1 -> ORB-1000 (1.2, 4 levels, FAST 10/0, edge 19) -> SearchForInitialization vs previous slice"
                  % (NEV, W, H, "raw (x,y,t,p) records resolved on the GPU through the calibrator's undistortion maps built from the EvETHZ "
                     "intrinsics, as the reference loader does on the CPU" if use_raw else
                     "already undistorted by the loader through the EvETHZ maps (EventData floats)"))
        else:
            wl = ("BASELINE.json configs[0] stand-in (W1): synthetic %d-event slices on %dx%d (EvETHZ.yaml l1ChunkSize; shapes generator, %s), "
                  "ev2im_gauss sigma=1 -> FAST detection of %d points (ORB extractor, 1 level, threshold 0, edge 9; no descriptors: the "
                  "event path tracks with KLT)" % (NEV, W, H, a.input, orb["nfeatures"]))
        out = base_line(env, frames / dt, dt, wl, dict(
            slices_per_step_per_gpu=B, events_per_slice=NEV, image=[W, H], input=a.input, sequences_per_gpu=S,
            parallelism="1 process/GPU, independent slices, RCCL gather of keypoints" if world > 1 else "1 GPU",
            mean_keypoints=float(nk.mean()), mean_matches=float(nm[1:].mean()) if (B > 1 and match) else 0.0))
        if prof:
            acc_bytes = 16.0 * NEV + W * H * 9.0            # 16 B x N events + W*H*(4 write + 1 write + 4 read)
            P = level_pixels(W, H, orb["scaleFactor"], orb["nlevels"])
            ext_bytes = 7.0 * P + 1321.0 * float(nk.mean())  # 7*P + 1321*K per frame
            r, per_step = roofline_of(prof, a.steps, lambda k: acc_bytes if k.startswith("ev_") else ext_bytes, B,
                                      "%s:%s:%d:%d" % (a.workload, a.input, B, NEV))
            if a.workload == "w2" and use_raw:
                c0 = seqs[0][0]
                ent, hot = c0.debug_counter("slot_entries"), c0.debug_counter("slot_hot_entries")
                if ent > 0 and "ev_gather" in prof:
                    r["issue_floor"] = issue_floor(ent, max(hot, 0), prof["ev_gather"][0] / max(prof["ev_gather"][1], 1))
            r["note"] = ("the accumulation is bound by instruction issue (vector ALU / LDS), not by HBM: every pixel adds its taps in event "
                         "order (49 taps per 16 B event, DESIGN.md section 4); `issue` carries the utilisation of the binding pipes, the HBM "
                         "fraction is reported as the contract asks")
            out["roofline"] = r
            out["kernels_ms_per_step"] = per_step
            out["kernels_ms_per_step_source"] = ("HIP events on the launch stream: %s over the timed steps, the other scopes over %d further steps after them"
                                                 % (", ".join(k for k in TIMED_SCOPES if k in per_step), min(a.steps, 5)))
        # ---- streaming ingest: the events arrive from the host (src/Event/EventLoader.cpp:535-577 hands chunks to the tracker) ----
        if a.stream and a.workload == "w2" and world == 1 and use_raw:
            out["streaming"] = stream_ingest(env, seqs[0], streams[0], ev16, offsets, B, NEV, dt / a.steps)
        # ---- single-slice latency of the seams as the reference calls them (host buffers, one slice per call): W1 only ----
        if a.workload == "w1" and a.latency_calls > 0 and world == 1:
            c = frontend.Context(device=env["local_rank"])
            ge = frontend.ORBextractor(imSize=(W, H), ctx=c, **orb)
            lat = {"ev2im_gauss": [], "orb_detect": [], "total": []}
            for i in range(a.latency_calls + 20):
                ev = lat_pairs[i % len(lat_pairs)][0]
                t0 = time.perf_counter()
                u8 = frontend.EvImConverter.ev2im_gauss(ev, W, H, 1.0, False, True, ctx=c)
                t1 = time.perf_counter()
                ge(u8, (0, 1000), False)
                t2 = time.perf_counter()
                if i >= 20:
                    lat["ev2im_gauss"].append(t1 - t0); lat["orb_detect"].append(t2 - t1); lat["total"].append(t2 - t0)
            # the same slice as the sensor delivers it: eorb_ev2im_gauss_raw resolves the raw events through the maps on the GPU
            # (what the reference's loader + ev2im_gauss do together)
            frontend.EvImConverter.set_undistort_maps(mx, my, True, ctx=c)
            lat["ev2im_gauss_raw"] = []; lat["total_raw"] = []
            for i in range(a.latency_calls + 20):
                rw = lat_pairs[i % len(lat_pairs)][1]
                t0 = time.perf_counter()
                u8 = frontend.EvImConverter.ev2im_gauss_raw(rw, W, H, 1.0, False, True, ctx=c)
                t1 = time.perf_counter()
                ge(u8, (0, 1000), False)
                t2 = time.perf_counter()
                if i >= 20:
                    lat["ev2im_gauss_raw"].append(t1 - t0); lat["total_raw"].append(t2 - t0)
            out["latency"] = {"what": "one %d-event slice per call, host buffers in and out (PCIe and binding overhead included): "
                                      "eorb_ev2im_gauss (undistorted float events, the reference's seam) or eorb_ev2im_gauss_raw (sensor "
                                      "events, loader fused) -> eorb_orb_extract(detect-only)" % NEV, "calls": a.latency_calls,
                              **{k: _pcts(v) for k, v in lat.items()}}
            c.close()
        # ---- CPU baseline: the oracle (port), bounded sample of the same workload; its outputs for the first slices are what the
        #      GPU's last timed step is verified against, bit for bit ----
        if ncpu > 0 and world == 1:
            spec = dict(kind="batch", W=W, H=H, events=NEV, orb=orb, sigma=1.0, input=a.input, want_desc=want_desc, match=match,
                        seed0=seed0, nunits=min(ncpu, 64))
            out["cpu_baseline"], ref = cpu_baseline(spec, ncpu, a.cpu_pool, "slices of %d events" % NEV, cpu_pairs, keep=ncmp)
            out["speedup_vs_cpu_1thread"] = out["value"] / out["cpu_baseline"]["value"]
            out["verified"] = verify_batch(gpu_out, ref, want_desc, match)
    for c_i, _, _ in seqs:
        c_i.close()
    return out


def verify_batch(g, ref, want_desc, match):
    """The GPU's outputs of the last timed step against the oracle's for the same slices (the `for k` order of
    src/Event/EventConversion.cc:231-263 decides every float): float image, running extremes, u8 image, keypoint records,
    descriptor bytes, SearchForInitialization matches against the previous slice -- all compared as bytes."""
    bad = []
    compared = ["f32 image bits", "min/max bits", "u8 image", "keypoint count", "keypoint records (28 B)"] + \
               (["descriptors (32 B)"] if want_desc else []) + (["matches12 + nmatches vs the previous slice (slices 1..)"] if (want_desc and match) else [])
    for b, r in enumerate(ref):
        n = len(r["kps"])
        if not _bits_equal(r["f32"], g["f32"][b]): bad.append("slice %d: f32 image (%d px)" % (b, int((r["f32"].view(np.uint32) != g["f32"][b].view(np.uint32)).sum())))
        if not _bits_equal(r["mm"], g["mm"][b]): bad.append("slice %d: extremes %s vs %s" % (b, r["mm"].tolist(), g["mm"][b].tolist()))
        if not np.array_equal(r["u8"], g["u8"][b]): bad.append("slice %d: u8 image" % b)
        if n != int(g["n"][b]): bad.append("slice %d: %d keypoints vs %d" % (b, n, int(g["n"][b]))); continue
        if not _bits_equal(r["kps"], g["kp"][b, :n]): bad.append("slice %d: keypoint records" % b)
        if want_desc and not np.array_equal(r["desc"], g["desc"][b, :n]): bad.append("slice %d: descriptors" % b)
        if want_desc and match and b >= 1:
            npv = len(ref[b - 1]["kps"])
            if int(r["nm"]) != int(g["nm"][b]) or not np.array_equal(r["m12"], g["m"][b, :npv]): bad.append("slice %d: matches (%s vs %d)" % (b, r["nm"], int(g["nm"][b])))
    return _verified(compared, bad, len(ref), "the last timed step's outputs for the first %d slices of the batch against the CPU oracle's (the cpu_baseline "
                     "leg's own results), compared as bytes outside the timed region" % len(ref))
    for c_i, _, _ in seqs:
        c_i.close()
    return out


# --------------------------------------------------------------------------------------------------------------------------
def run_frames(env):
    """w3 / w4: one frame per call through the host-buffer entry points, as Tracking::GrabImage* / Frame construction call them."""
    a, world, rank, torch = env["a"], env["world"], env["rank"], env["torch"]
    fe = env["frontend"]
    c = fe.Context(device=env["local_rank"])
    nfr = 32
    if a.workload == "w3":
        W, H = 240, 180
        orb = dict(nfeatures=1000, scaleFactor=1.2, nlevels=4, iniThFAST=10, minThFAST=0, edgeTh=19)
        frames = _w3_frames(nfr, 3 + rank)
        ge = fe.ORBextractor(imSize=(W, H), ctx=c, **orb)
        m = fe.ORBmatcher(0.9, True, c)
        st = {"prev": None, "i": 0, "nm": [], "nk": []}

        def one(keep=False):
            i = st["i"]; st["i"] += 1
            _, kps, desc, _ = ge(frames[i % nfr])
            k, d, o = _w3_mix(kps, desc, i % nfr)
            F = fe.FrameView(k, d, W, H, o)
            res = {"kps": kps, "desc": desc} if keep else None
            if st["prev"] is not None:
                P, Pd = st["prev"]
                pm = np.stack([P.kps["x"], P.kps["y"]], axis=1)
                n1, m12, _ = m.SearchForInitialization(P, F, pm, 100)
                pa = _w3_proj_args(P.kps, Pd, P.is_orb, len(k))
                n2, cmp_ = m.SearchByProjectionLast(F, P, pa["valid"], pa["uv"], pa["mp_desc"], pa["mp_obs"], pa["cur_mp"], 15.0, pa["ls"], 0)
                st["nm"].append((n1, n2))
                if keep:
                    res.update(n1=n1, m12=m12, n2=n2, cur_mp=cmp_)
            st["nk"].append(len(kps))
            st["prev"] = (F, d)
            return res
        wl = ("BASELINE.json configs[2] stand-in (W3): 240x180 texture frames, ORB-1000 (1.2, 4 levels, FAST 10/0, edge 19) + 500 AKAZE-like "
              "61-byte rows per frame, MixedMatcher SearchForInitialization (window 100) + SearchByProjection(cur, last) with the type "
              "gate; one frame per call, host buffers")
        P = level_pixels(W, H, 1.2, 4)
        ncpu = 40 if a.cpu_slices < 0 else a.cpu_slices
    else:
        W, H = 346, 260
        orb = dict(nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=10, minThFAST=0, edgeTh=15)
        frames = _w4_frames(nfr, 4 + rank)
        ge = fe.ORBextractor(imSize=(W, H), ctx=c, **orb)
        bf = fe.BFMatcher(c)
        q, t = _w4_desc()
        st = {"i": 0, "nk": [], "nm": []}

        def one(keep=False):
            i = st["i"]; st["i"] += 1
            _, kps, desc, _ = ge(frames[i % nfr])
            idx, dist = bf.knnMatch2(q, t)
            st["nk"].append(len(kps))
            return {"kps": kps, "desc": desc, "idx": idx, "dist": dist} if keep else None
        wl = ("BASELINE.json configs[3] stand-in (W4): 346x260 texture frames, ORB extraction of 2 000 features on 8 levels @1.2 (edge 15) "
              "+ cv::BFMatcher-style Hamming 2-NN of 2000 x 2000 descriptors (planted matches); one frame per call, host buffers")
        P = level_pixels(W, H, 1.2, 8)
        ncpu = 24 if a.cpu_slices < 0 else a.cpu_slices
    FR = 16                                   # frames per step

    def step():
        for _ in range(FR):
            one()

    def arm():
        st["nk"].clear(); st["nm"].clear()
    dt = timed_steps(env, step, arm)
    nk, nm = list(st["nk"]), list(st["nm"])
    # the per-kernel times come from a second pass over the same steps: with one frame per call the profiling scopes (two event
    # records per kernel group) are a visible share of a call, and the value above is the product's, not the profiler's
    prof = {}
    if not a.no_prof:
        c.prof_reset(); c.prof_enable(True)
        for _ in range(a.steps):
            step()
        c.prof_enable(False); prof = c.prof_results()
    st["nk"], st["nm"] = nk, nm
    out = None
    if rank == 0:
        K = float(np.mean(st["nk"]))
        cfg = dict(frames_per_step_per_gpu=FR, image=[W, H], mean_keypoints=K,
                   parallelism="1 process/GPU, independent replicas" if world > 1 else "1 GPU")
        if st["nm"]:
            cfg["mean_matches_init"] = float(np.mean([x[0] for x in st["nm"]])); cfg["mean_matches_proj"] = float(np.mean([x[1] for x in st["nm"]]))
        out = base_line(env, world * FR * a.steps / dt, dt, wl, cfg)
        if prof:
            ext_bytes = 7.0 * P + 1321.0 * K
            bfb = (2000 + 2000) * 32 + 2000 * 16.0
            r, per_step = roofline_of(prof, a.steps, lambda k: bfb if k.startswith("bf_") else ext_bytes, 1, "%s:frames" % a.workload)
            r["note"] = ("one frame per call: the working set (< 1 MB) lives in L2 / Infinity Cache and every kernel is launch- or "
                         "latency-bound; the HBM fraction is reported as the contract asks; kernel times from a second, profiled "
                         "pass over the same steps")
            out["roofline"] = r
            out["kernels_ms_per_step"] = per_step
        if world == 1:
            # throughput mode of the same extraction (+ SearchForInitialization of frame b against b-1, ORB rows only): the frames
            # resident in HBM, one call per batch of 32 (eorb_fe_run_batch_images_dev)
            fbi = fe.FrontEndBatch(W, H, 1.0, False, max_batch=nfr, max_events=1, match=True, want_desc=True, windowSize=100, nnratio=0.9,
                                   checkOri=True, **orb)
            d_fr = torch.from_numpy(np.concatenate([f.ravel() for f in frames])).to(env["dev"])
            capb = fbi.cap
            t_kp = torch.empty(nfr * capb * 28, dtype=torch.uint8, device=env["dev"]); t_d = torch.empty(nfr * capb * 32, dtype=torch.uint8, device=env["dev"])
            t_n = torch.zeros(nfr, dtype=torch.int32, device=env["dev"]); t_m = torch.empty(nfr * capb, dtype=torch.int32, device=env["dev"])
            t_nm = torch.zeros(nfr, dtype=torch.int32, device=env["dev"])
            def bstep():
                fbi.run_images_dev(d_fr.data_ptr(), nfr, t_kp.data_ptr(), t_d.data_ptr(), t_n.data_ptr(), t_m.data_ptr(), t_nm.data_ptr())
            for _ in range(3):
                bstep()
            fbi.ctx.sync()
            t0 = time.perf_counter()
            nrep = 20
            for _ in range(nrep):
                bstep()
            fbi.ctx.sync()
            tb = time.perf_counter() - t0
            out["batched"] = {"value": nfr * nrep / tb, "unit": "frames/s", "frames_per_call": nfr, "ms_per_call": tb / nrep * 1e3,
                              "mean_keypoints": float(t_n.cpu().numpy().mean()),
                              "what": "the same frames resident in HBM, %d per call: extraction + SearchForInitialization vs the previous frame "
                                      "(eorb_fe_run_batch_images_dev)" % nfr}
            fbi.ctx.close()
        if ncpu > 0 and world == 1:
            spec = dict(kind=a.workload, orb=orb, seed0=(3 if a.workload == "w3" else 4), nunits=nfr, events=0)
            nver = min(ncpu, nfr)
            out["cpu_baseline"], ref = cpu_baseline(spec, ncpu, a.cpu_pool, "frames", keep=nver)
            out["speedup_vs_cpu_1thread"] = out["value"] / out["cpu_baseline"]["value"]
            # the same calls the timed steps made, from a fresh start, their results kept: frame i against the oracle's frame i
            st["i"] = 0
            if "prev" in st:
                st["prev"] = None
            bad = []
            for i in range(nver):
                g = one(keep=True); r = ref[i]
                if len(g["kps"]) != len(r["kps"]) or not _bits_equal(g["kps"], r["kps"]): bad.append("frame %d: keypoints" % i)
                elif not np.array_equal(g["desc"], r["desc"]): bad.append("frame %d: descriptors" % i)
                if a.workload == "w3" and i >= 1:
                    if g["n1"] != r["n1"] or not np.array_equal(g["m12"], r["m12"]): bad.append("frame %d: SearchForInitialization (%s vs %s)" % (i, g["n1"], r["n1"]))
                    if g["n2"] != r["n2"] or not np.array_equal(g["cur_mp"], r["cur_mp"]): bad.append("frame %d: SearchByProjection (%s vs %s)" % (i, g["n2"], r["n2"]))
                if a.workload == "w4" and (not np.array_equal(g["idx"], r["idx"]) or not np.array_equal(g["dist"], r["dist"])): bad.append("frame %d: 2-NN" % i)
            cmp_ = ["keypoint records (28 B)", "descriptors (32 B)"] + (["SearchForInitialization matches12 + count", "SearchByProjection(cur, last) map-point indices + count"]
                                                                        if a.workload == "w3" else ["brute-force 2-NN indices + distances (2000 x 2000)"])
            out["verified"] = _verified(cmp_, bad, nver, "the calls of the timed steps repeated from a fresh start for the first %d frames, their results against "
                                        "the CPU oracle's (the cpu_baseline leg's own results), compared as bytes" % nver)
    c.close()
    return out


def run_chain(env):
    """w1full: config 1 as the reference executes it.  One step = one pass of EvImBuilder::Track's chunk loop over a synthetic stream
    (128 000 events: about 70 chunks of 2 000 ... 5 000 events, about 20 dispatches), one chunk per call through the one-call seams."""
    a, world, rank, torch = env["a"], env["world"], env["rank"], env["torch"]
    fe, synth = env["frontend"], env["synth"]
    W, H, CH, NCH = 240, 180, 2000, 64
    evs = synth.l1_stream(n_chunks=NCH, chunk=CH, seed=5 + rank)
    g = fe.EvImBuilder(W, H, cam=synth.EVETHZ_PINHOLE, keep_images=False)
    last = {}

    def step():
        g.resetAll()
        last["res"] = fe.feed_chunks(g, evs, synth.l1_mci_poses)
    dt = timed_steps(env, step)
    nchunks = len(last["res"])
    ndisp = sum(r["dispatched"] for r in last["res"])
    prof = {}
    if not a.no_prof:
        for c in (g.ctx, g.ctx_l2):
            c.prof_reset(); c.prof_only(()); c.prof_enable(True)
        step()
        for c in (g.ctx, g.ctx_l2):
            c.prof_enable(False)
            for k, (ms, n) in c.prof_results().items():
                a0, n0 = prof.get(k, (0.0, 0)); prof[k] = (a0 + ms, n0 + n)
    out = None
    if rank == 0:
        wl = ("BASELINE.json configs[0] as the reference executes it (W1 full): EvImBuilder::Track over a synthetic %d-event stream on %dx%d -- per chunk "
              "(l1ChunkSize 2 000, adapted by the window-size rule) ev2im_gauss sigma=1 -> INIT: FAST detection of 400 points + LK reference / "
              "TRACKING: pyramidal LK (win 23, 2 levels, 10 iterations) against the reference frame; on a dispatch the four-way reconstruction "
              "contest (2 x ev2mci SE3, SE2, event histogram; focus; cv::normalize) + FAST detection of 800 points on the winner; one chunk per "
              "call, host buffers" % (len(evs), W, H))
        out = base_line(env, world * nchunks * a.steps / dt, dt, wl, dict(chunks_per_step=nchunks, dispatches_per_step=ndisp, image=[W, H],
                        parallelism="1 process/GPU, independent sequences" if world > 1 else "1 GPU"))
        out["ms_per_chunk"] = dt / a.steps / nchunks * 1e3
        if prof:
            acc_bytes = 16.0 * CH + W * H * 9.0
            r, per_step = roofline_of(prof, 1, lambda k: acc_bytes, 1, "w1full:chain")
            r["note"] = ("one chunk per call: every kernel is launch- or latency-bound (a 2 000-event chunk is 0.4 MB of algorithmic traffic); "
                         "the HBM fraction is reported as the contract asks; kernel times from a second, profiled pass")
            out["roofline"] = r
            out["kernels_ms_per_step"] = per_step
        if a.cpu_slices != 0 and world == 1:
            from oracle import orc_chain               # the CPU baseline / the checker, outside the timed region
            o = orc_chain.L1Chain(W, H, cam=synth.EVETHZ_PINHOLE, fast=True)
            orc_chain.run_sequence(o.track, evs[:8 * CH], CH, synth.l1_mci_poses)            # warm-up
            o = orc_chain.L1Chain(W, H, cam=synth.EVETHZ_PINHOLE, fast=True)
            t0 = time.perf_counter()
            ref = orc_chain.run_sequence(o.track, evs, CH, synth.l1_mci_poses)
            tcpu = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": len(ref) / tcpu, "unit": "frames/s", "cores": 1, "kind": "port", "host_cpus": os.cpu_count(),
                                   "sample": "the same stream through the oracle's chain (oracle/orc_chain.py), %d chunks, %.2f s" % (len(ref), tcpu)}
            out["speedup_vs_cpu_1thread"] = out["value"] / out["cpu_baseline"]["value"]
            # the last timed pass against the oracle's chain, chunk by chunk
            gk = fe.EvImBuilder(W, H, cam=synth.EVETHZ_PINHOLE, keep_images=True)
            gres = fe.feed_chunks(gk, evs, synth.l1_mci_poses)
            gk.close()
            bad = []
            if len(gres) != len(ref): bad.append("%d chunks vs %d" % (len(gres), len(ref)))
            for i, (x, y) in enumerate(zip(gres, ref)):
                if x["state"] != y["state"] or x["dispatched"] != y["dispatched"] or x.get("chunk_size") != y.get("chunk_size"): bad.append("chunk %d: state / dispatch / chunk size" % i); break
                if "image" in y and not np.array_equal(x["image"], y["image"]): bad.append("chunk %d: event image" % i)
                if "kps" in y and not _bits_equal(x["kps"], y["kps"]): bad.append("chunk %d: INIT keypoints" % i)
                if "pts" in y and not (_bits_equal(x["pts"], y["pts"]) and np.array_equal(x["status"], y["status"]) and _bits_equal(x["err"], y["err"])): bad.append("chunk %d: LK points / status / error" % i)
                if y["dispatched"]:
                    xm, ym = x["mci"], y["mci"]
                    if xm["winner"] != ym["winner"] or not _bits_equal(xm["focus"], ym["focus"]): bad.append("chunk %d: contest focus / winner" % i)
                    if not np.array_equal(xm["image"], ym["image"]) or not _bits_equal(xm["l2_kps"], ym["l2_kps"]): bad.append("chunk %d: winner image / L2 keypoints" % i)
            out["verified"] = _verified(["state machine (state, dispatch, next chunk size)", "u8 event image per chunk", "INIT keypoint records", "LK points / status / error bits",
                                         "contest focus bits + winner", "winner image", "L2 keypoint records"], bad, len(ref),
                                        "the same stream through the GPU chain once more with the images kept, chunk by chunk against the oracle's chain")
    g.close()
    return out


if __name__ == "__main__":
    main()
