#!/usr/bin/env python3
"""bench.py -- front-end frames/s on MI355X (BASELINE.json metric, config[1]).

One "step" = one pass of the hot path (event accumulation -> ORB-1000 extraction -> frame-to-frame
Hamming matching) over one batch of `--batch` synthetic time-slices of `--events` events on a
240x180 sensor, all inputs resident in HBM before the timed region.  Rank r of N works on its own
independent slices (weak scaling, no data-path collective); the per-slice keypoint records are
gathered to rank 0 over RCCL at the end of every step (north_star's "final keypoint gather").

Prints ONE JSON line (rank 0): metric/value/unit + `roofline` for the dominant kernel (live HIP-event
timing on the launch stream) + `cpu_baseline` (the CPU oracle, -O3 -march=native build, 1 thread, on a
bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "front-end frames/s (event-accumulate + extract + match) per GPU; HBM GB/s vs roofline"
# HBM bytes one ev_gather launch moves per million events (rocprofv3 --pmc FETCH_SIZE x2 (gfx950) + WRITE_SIZE, see
# profiles/r01_v8_pmc_traffic.txt; float input: r01_v6); bench.py cannot collect PMC counters itself
GATHER_TRAFFIC_PER_MEV = {"raw": 31.8e6, "float": 30.9e6}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="time-slices per step per GPU")
    ap.add_argument("--events", type=int, default=1000000, help="events per slice")
    ap.add_argument("--cpu-slices", type=int, default=64, help="slices timed for the CPU baseline (0 = skip)")
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
    return ap.parse_args()


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from eorb_slam_amd import frontend, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            print("bench.py: --gpus %d needs torch.distributed.run with %d ranks" % (a.gpus, a.gpus), file=sys.stderr)
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the front end has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    local_rank %= max(torch.cuda.device_count(), 1)     # (a rehearsal with more ranks than GPUs shares devices; no effect on a full node)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    W, H, B, NEV = 240, 180, a.batch, a.events
    orb = dict(nfeatures=1000, scaleFactor=1.2, nlevels=4, iniThFAST=10, minThFAST=0, edgeTh=19)
    # ---- synthetic input: B independent slices per rank (seeded), packed to the 16 B HBM record ----
    pairs = [synth.shapes_events(NEV, W, H, seed=2 + 1000 * rank + b, motion=0.5, undistort=True, return_raw=True) for b in range(B)]
    slices = [p[0] for p in pairs]
    use_raw = a.input == "raw"
    if use_raw:
        ev16 = np.concatenate([p[1] for p in pairs])               # eorb_raw_event, 16 B
    else:
        ev16 = np.concatenate([frontend.pack_events(s) for s in slices])
    offsets = np.arange(B + 1, dtype=np.int64) * NEV
    # host copies are only needed again by the CPU baseline (rank 0 at N = 1): keep those slices, drop the rest (40 MB per slice)
    keep = min(a.cpu_slices, B) if (a.cpu_slices > 0 and world == 1) else 0
    pairs = pairs[:keep]; slices = slices[:keep]

    S = max(1, a.sequences)
    d_ev = torch.from_numpy(ev16.view(np.uint8)).to(dev)
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(S - 1)]
    seqs = []
    for si in range(S):
        ctx_i = frontend.Context(device=local_rank, stream=streams[si].cuda_stream)
        fb_i = frontend.FrontEndBatch(W, H, 1.0, False, max_batch=B, max_events=NEV, match=True, windowSize=100, nnratio=0.9,
                                      checkOri=True, ctx=ctx_i, **orb)
        if use_raw:
            mx, my = synth.undistort_lut(W, H)
            frontend.EvImConverter.set_undistort_maps(mx, my, True, ctx=ctx_i)
        cap = fb_i.cap
        # keypoint records of a batch = one packed buffer {n[B] | kp[B][cap] | desc[B][cap][32]}: ONE gather per step at N > 1
        off_kp = (B * 4 + 255) & ~255
        off_desc = (off_kp + B * cap * 28 + 255) & ~255
        rec = torch.zeros(off_desc + B * cap * 32, dtype=torch.uint8, device=dev)
        bufs = dict(img=torch.empty(B * W * H, dtype=torch.uint8, device=dev),
                    rec=rec, n=rec[:B * 4].view(torch.int32), kp=rec[off_kp:off_kp + B * cap * 28], desc=rec[off_desc:],
                    m=torch.empty(B * cap, dtype=torch.int32, device=dev),
                    nm=torch.zeros(B, dtype=torch.int32, device=dev))
        seqs.append((ctx_i, fb_i, bufs))
    ctx, fb = seqs[0][0], seqs[0][1]
    d_n, d_nm = seqs[0][2]["n"], seqs[0][2]["nm"]
    gdev = dev if a.backend == "nccl" else torch.device("cpu")
    gather_rec = [torch.empty_like(seqs[0][2]["rec"], device=gdev) for _ in range(world)] if (world > 1 and rank == 0) else None
    step_no = [0]

    def step():
        si = step_no[0] % S; step_no[0] += 1
        _, fb_i, bf = seqs[si]
        with torch.cuda.stream(streams[si]):
            fb_i.run_dev(d_ev.data_ptr(), offsets, bf["img"].data_ptr(), bf["kp"].data_ptr(), bf["desc"].data_ptr(), bf["n"].data_ptr(),
                         bf["m"].data_ptr(), bf["nm"].data_ptr(), raw=use_raw)
            if world > 1:       # final keypoint gather (RCCL over xGMI), fixed-capacity records
                dist.gather(bf["rec"] if a.backend == "nccl" else bf["rec"].cpu(), gather_rec, dst=0)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if not a.no_prof:
        for c_i, _, _ in seqs:
            c_i.prof_reset(); c_i.prof_enable(True)
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
        t = torch.tensor([dt], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = {}
    if not a.no_prof:
        for c_i, _, _ in seqs:
            c_i.prof_enable(False)
            for k, (ms, n) in c_i.prof_results().items():
                a0, n0 = prof.get(k, (0.0, 0))
                prof[k] = (a0 + ms, n0 + n)
    nk = d_n.cpu().numpy(); nm = d_nm.cpu().numpy()

    if rank == 0:
        frames = world * B * a.steps
        ms_per_step = dt / a.steps * 1e3
        out = {
            "metric": METRIC, "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: synthetic %d events/slice on %dx%d (shapes generator, DAVIS sensor "
                                   "pixels; %s), ev2im_gauss sigma=1 -> ORB-1000 (1.2, 4 levels, FAST 10/0, "
                                   "edge 19) -> SearchForInitialization vs previous slice"
                                   % (NEV, W, H, "raw (x,y,t,p) records resolved on the GPU through the calibrator's undistortion maps built "
                                      "from the EvETHZ intrinsics, as the reference loader does on the CPU" if use_raw else
                                      "already undistorted by the loader through the EvETHZ maps (EventData floats)"),
                       "slices_per_step_per_gpu": B, "events_per_slice": NEV, "image": [W, H], "input": a.input, "sequences_per_gpu": S,
                       "parallelism": "1 process/GPU, independent slices, RCCL gather of keypoints" if world > 1 else "1 GPU",
                       "mean_keypoints": float(nk.mean()), "mean_matches": float(nm[1:].mean()) if B > 1 else 0.0},
        }
        # ---- roofline of the dominant kernel (live HIP-event timing on the launch stream) ----
        if prof:
            tot = {k: v[0] for k, v in prof.items()}
            dom = max(tot, key=tot.get)
            ms, launches = prof[dom]
            avg_ms = ms / max(launches, 1)
            # accumulate = 16 B x N events + W*H*(4 write + 1 write + 4 read) per slice
            acc_bytes = 16.0 * NEV + W * H * 9.0
            # extract = 7*P + 1321*K per frame (P = sum of level pixels, K = keypoints)
            P = 240 * 180 + 200 * 150 + 167 * 125 + 139 * 104
            ext_bytes = 7.0 * P + 1321.0 * float(nk.mean())
            unit_bytes = acc_bytes if dom.startswith("ev_") else ext_bytes
            achieved = unit_bytes * B / (avg_ms * 1e-3) / 1e9
            # HBM traffic of the dominant kernel per launch: measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this workload
            # (profiles/r01_v8_pmc_traffic.txt: ev_gather_raw 2 x 2022 MB fetched + 28 MB written per 128 slices of 1 Mev with the
            # gfx950 x2 FETCH correction), scaled to this launch; bench.py cannot collect PMC counters itself
            traffic = GATHER_TRAFFIC_PER_MEV[a.input] * (NEV / 1e6) * B if dom == "ev_gather" else None
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "traffic_source": "profiles/r01_v8_pmc_traffic.txt",
                               "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": unit_bytes * B,
                               "note": "ev_gather (ev_gather_raw_kernel for raw events) is issue/latency-bound by construction: every pixel "
                                       "adds its taps in event order (49 taps per 16 B event, DESIGN.md section 4); the HBM fraction is "
                                       "reported as the contract asks"}
            out["kernels_ms_per_step"] = {k: v[0] / a.steps for k, v in sorted(prof.items())}
        # ---- CPU baseline: the oracle (port), 1 thread, bounded sample of the same workload ----
        if a.cpu_slices > 0 and world == 1:       # reported at N=1 only (the contract); N>1 runs stay short
            from oracle import oracle_py
            oe = oracle_py.OrbExtractor(fast=True, imWidth=W, **orb)
            ns = min(a.cpu_slices, B)
            # warm the page cache / branch predictors on one small slice
            oracle_py.ev2im_gauss(slices[0][:20000], W, H, 1.0, False, True, fast=True)
            prev = None
            if use_raw:
                lx, ly = synth.undistort_lut(W, H)
            tc = time.perf_counter()
            done = 0
            reps = 0
            while done < a.cpu_slices:
                for b in range(ns):
                    evs = oracle_py.undistort_events(pairs[b][1], lx, ly, W, H, True, 1.0) if use_raw else slices[b]
                    _, u8, _ = oracle_py.ev2im_gauss(evs, W, H, 1.0, False, True, fast=True)
                    _, kps, desc, _ = oe.extract(u8)
                    F = oracle_py.Frame(kps, desc, W, H, fast=True)
                    if prev is not None:
                        pm = np.stack([prev.kps["x"], prev.kps["y"]], axis=1)
                        oracle_py.search_for_initialization(prev, F, pm, 100, 0.9, True)
                    prev = F
                    done += 1
                    if done >= a.cpu_slices:
                        break
                reps += 1
            tcpu = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": done / tcpu, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "%d slices of %d events (same generator/seeds as the GPU run), oracle "
                                             "built -O3 -march=native -ffp-contract=off, %.1f s" % (done, NEV, tcpu),
                                   "host_cpus": os.cpu_count()}
            out["speedup_vs_cpu_1thread"] = out["value"] / out["cpu_baseline"]["value"] / world
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    for c_i, _, _ in seqs:
        c_i.close()


if __name__ == "__main__":
    main()
