#!/usr/bin/env python3
"""Turns rocprofv3 output directories of ONE bench.py command into profiles/roofline_inputs.json, the file bench.py reads
`roofline.traffic` / `roofline.issue` from (it cannot collect PMC counters itself).  The file carries the hash of the library
sources it was measured on; bench.py reports the numbers only while that hash equals the hash of the sources it runs.

usage: roofline_inputs.py <out.json> <key> <stats_dir> <fetch_dir> <write_dir> <sq1_dir> <sq2_dir> -- <bench command line>
key = "<workload>:<input>:<batch>:<events>".  Counters (separate --pmc passes, MI355X_MICROARCH.md):
  traffic  = FETCH_SIZE x 1024 x f + WRITE_SIZE x 1024, f = 2 for the kernels whose reads are wide coalesced streams (gfx950: FETCH_SIZE counts
             their 128-byte requests as 64 bytes; list WIDE_READ below), 1 for gathers / list walkers / per-keypoint kernels
  cycles    = GRBM_GUI_ACTIVE / 8                                          (summed over the 8 XCDs)
  valu_frac = SQ_INSTS_VALU x 4 cycles / (cycles x 1024 SIMDs)             (a wave64 VALU instruction holds its SIMD's ALU for 4 cycles)
  lds_frac  = SQ_LDS_IDX_ACTIVE / (cycles x 256 CUs)                       (LDS-array cycles)
  salu_frac = SQ_INSTS_SALU / (cycles x 256 CUs)                           (one scalar ALU per CU, one instruction per cycle)
  wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES                                 (wave-cycles parked at s_waitcnt / barrier)
"""
import collections, csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCOPES = {   # HIP-event scope of bench.py -> kernels it brackets (substrings of the demangled names)
    "ev_gather": ["sl_gather_kernel", "sl_hot_kernel", "ev_gather_"],
    # slot form: one scope per kernel of the binning (the plan kernels run on their own stream beside the scatter)
    "ev_count": ["sl_chunks_kernel", "sl_count_kernel", "sl_count_lds_kernel"], "ev_scan": ["sl_scan_kernel"],
    "ev_scatter": ["sl_scatter", "sl_plan_kernel", "sl_tasks_kernel"],
    # batch pipeline (polarity, wide stamps, sparse slices)
    "ev_bin": ["ev_count_kernel", "ev_scan_kernel", "ev_scatter", "ev_tile_hist", "ev_tile_order"],
    "ev_normalize": ["ev_normalize_kernel"], "ev_dedupe": ["dd_insert_kernel"], "ev_stamp_tables": ["ev_pre_kernel", "ev_stamp_kernel", "ev_src_info_kernel"],
    "orb_pyr": ["pyr_level0_kernel", "pyr_resize_kernel"], "orb_fast_cells": ["fast_cells_kernel"], "orb_octree": ["octree_kernel"],
    "orb_orient": ["orient_kernel"], "orb_blur": ["blur_kernel"], "orb_brief": ["brief_kernel"], "orb_assemble": ["assemble_kernel"],
    "search_init": ["win_cand_kernel<0>", "win_resolve_kernel<0>", "win_histo_kernel<0>"],
    "search_proj_last": ["win_cand_kernel<1>", "win_resolve_kernel<1>", "win_histo_kernel<1>"],
    "bf_knn2": ["bf_knn2_kernel"], "klt_track": ["klt_"],
    "ev_focus": ["ev_focus_"], "ev_cvnormalize": ["ev_mm_reset", "ev_minmax_final", "ev_cvnormalize"], "ev_warp_se3": ["ev_warp_se3_kernel"], "ev_warp_se2": ["ev_warp_se2_kernel"],
}
# FETCH_SIZE counts the 128-byte requests of WIDE COALESCED reads as 64 bytes on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section): the
# x 2 applies to the kernels whose reads are such streams -- the passes over the event records and over whole images -- not to the
# gathers / list walkers / per-keypoint kernels, whose reads are scalar loads, dword gathers or L2-resident tables
WIDE_READ = ["sl_count", "sl_scatter", "sl_chunks", "ev_count_kernel", "ev_scatter", "dd_insert", "ev_unpack4", "ev_normalize", "ev_pre_kernel",
             "pyr_level0", "pyr_resize", "blur_kernel", "ev_undistort", "txt_", "ev_minmax_final", "ev_cvnormalize", "ev_warp", "fe_publish"]
N_SIMD, N_CU, N_XCD = 1024, 256, 8


def source_hash():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "eorb_slam_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "eorb_slam_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "eorb_fe.h")])
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def counters(d):
    """{kernel name: {counter: average per dispatch}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k, c = r["Kernel_Name"], r["Counter_Name"]
            acc[k][c] += float(r["Counter_Value"]); n[k][c].add(r["Dispatch_Id"])
    return {k: {c: v / max(len(n[k][c]), 1) for c, v in cs.items()} for k, cs in acc.items()}


def stats(d):
    out = {}
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Name"]] = (float(r["AverageNs"]) * 1e-6, int(r["Calls"]))
    return out


def main():
    out_path, key, d_stats, d_fetch, d_write, d_sq1, d_sq2 = sys.argv[1:8]
    cmd = " ".join(sys.argv[9:]) if len(sys.argv) > 9 else ""
    st = stats(d_stats); cf = counters(d_fetch); cw = counters(d_write); c1 = counters(d_sq1); c2 = counters(d_sq2)
    scopes = {}
    for scope, pats in SCOPES.items():
        ks = [k for k in st if any(p in k for p in pats)]
        if not ks:
            continue
        e = {"kernels": {}, "traffic_bytes": 0.0, "rocprof_ms": 0.0}
        for k in ks:
            ff = 2 if any(p in k for p in WIDE_READ) else 1
            fetch = cf.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 * ff; write = cw.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
            e["kernels"][k.split("(")[0]] = {"avg_ms": st[k][0], "calls": st[k][1], "fetch_bytes": fetch, "fetch_factor": ff, "write_bytes": write}
            e["traffic_bytes"] += fetch + write; e["rocprof_ms"] += st[k][0]
        def issue_of(k):
            a, b = c1.get(k, {}), c2.get(k, {})
            cyc = b.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD      # (the counter is summed over the 8 XCDs: 4.38e7 for a 2.36 ms launch at 2.3 GHz)
            if not (cyc > 0 and a):
                return None
            return {"kernel": k.split("(")[0], "valu_frac": a.get("SQ_INSTS_VALU", 0.0) * 4 / (cyc * N_SIMD),
                    "salu_frac": a.get("SQ_INSTS_SALU", 0.0) / (cyc * N_CU),
                    "lds_frac": b.get("SQ_LDS_IDX_ACTIVE", 0.0) / (cyc * N_CU),
                    "wait_frac": a.get("SQ_WAIT_ANY", 0.0) / max(a.get("SQ_WAVE_CYCLES", 0.0), 1.0),
                    "issue_stall_frac": a.get("SQ_WAIT_INST_ANY", 0.0) / max(a.get("SQ_WAVE_CYCLES", 0.0), 1.0),
                    "counters": {**{kk: v for kk, v in a.items()}, **{kk: v for kk, v in b.items()}}}
        dom = max(ks, key=lambda k: st[k][0])
        i = issue_of(dom)
        if i:
            e["issue"] = i
        if scope == "ev_gather" and len(ks) > 1:
            # sl_gather_kernel and sl_hot_kernel run side by side on two streams: the scope lasts as long as the longer one.  (The
            # counters of a pass are collected with the kernels serialised by the profiler: per-kernel fractions, not a sum.)
            e["concurrent"] = True
            e["rocprof_ms"] = max(st[k][0] for k in ks)
            e["issue_by_kernel"] = {k.split("(")[0]: {kk: vv for kk, vv in (issue_of(k) or {}).items() if kk != "counters"} for k in ks}
        if scope == "ev_scatter" and any("sl_scatter" in k for k in ks):
            # sl_plan_kernel and sl_tasks_kernel run on their own stream beside the scatter: their traffic counts, their time does not
            e["rocprof_ms"] = sum(st[k][0] for k in ks if "sl_plan_kernel" not in k and "sl_tasks_kernel" not in k)
            e["beside_the_scatter"] = [k.split("(")[0] for k in ks if "sl_plan_kernel" in k or "sl_tasks_kernel" in k]
        scopes[scope] = e
    doc = {}
    if os.path.exists(out_path):
        try:
            doc = json.load(open(out_path))
        except Exception:
            doc = {}
    if doc.get("src_hash") != source_hash():
        doc = {"src_hash": source_hash(), "entries": {}}
    doc["note"] = ("rocprofv3 --kernel-trace --stats and separate --pmc passes of the command of each entry; formulas in tools/roofline_inputs.py; "
                   "bench.py uses an entry only while src_hash equals the hash of the sources it runs")
    doc["entries"][key] = {"command": cmd, "scopes": scopes}
    json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps({key: {s: {"traffic_MB": round(e["traffic_bytes"] / 1e6, 1), "rocprof_ms": round(e["rocprof_ms"], 3),
                                "issue": {k: round(v, 3) for k, v in e.get("issue", {}).items() if k.endswith("_frac")}} for s, e in scopes.items()}}, indent=1))


if __name__ == "__main__":
    main()
