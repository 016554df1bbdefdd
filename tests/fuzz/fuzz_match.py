"""Re-runs the parity tests (matchers, tracker, vocabulary, accumulation, extractor, batched pipeline) of tests/test_gpu_parity.py on fresh random data: every integer seed the
tests pass to numpy or to the synthetic generators is offset by the round number.  Parity assertions compare against the CPU
oracle; the tests' sanity assertions (`assert n > 10`) can fail on an unlucky seed and are reported separately.
Run on the GPU box: python tests/fuzz/fuzz_match.py [rounds]"""
import inspect, os, sys, traceback
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from eorb_slam_amd import frontend as fe, synth
from oracle import oracle_py as orc
import test_gpu_parity as T

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
orc.build()
ctx = fe.Context()
NAMES = ["test_bf_knn2", "test_search_for_initialization", "test_search_for_initialization_mixed_gate", "test_search_by_projection_last",
         "test_search_by_projection_map", "test_search_by_projection_last_mixed_gate", "test_search_by_projection_map_mixed_gate",
         "test_window_matchers_state_chains", "test_ev2mci_kannala_brandt8", "test_unsynced_ragged_batches_keep_their_own_tables", "test_tracked_descriptors_and_level_assignment", "test_search_by_bow", "test_search_by_bow_keyframes",
         "test_search_for_triangulation", "test_kf_radius_match_fuse_sim3", "test_search_by_projection_keyframe", "test_bow_transform",
         "test_klt_pyr_lk", "test_hamming_window_match", "test_distinctive_descriptors", "test_frontend_batch_matches_oracle_pipeline",
         "test_ev2im_gauss_bit_exact", "test_ev2im_gauss_shapes_lut", "test_ev2im_gauss_hot_pixel_order", "test_frontend_batch_ragged_and_empty_slices",
         "test_raw_undistort_events", "test_raw_ev2im_gauss_equals_loader_then_ev2im_gauss", "test_raw_mvsec_size_and_wide_stamps",
         "test_raw_ev2im_count", "test_frontend_batch_raw_equals_float_path", "test_ev2mci_se3", "test_ev2mci_se2_and_focus_contest",
         "test_orb_extract_texture", "test_orb_extract_event_image", "test_orb_extract_detect_only_fast_mode", "test_orb_extract_mvsec_shape"]
_rng0 = np.random.default_rng
_gens = {n: getattr(synth, n) for n in dir(synth) if callable(getattr(synth, n)) and "seed" in inspect.signature(getattr(synth, n)).parameters}
off = [0]
def rng_patched(seed=None, *a, **k):
    return _rng0(seed + 7919 * off[0] if isinstance(seed, (int, np.integer)) else seed, *a, **k)
np.random.default_rng = rng_patched
def wrap(f):
    sig = inspect.signature(f)
    def g(*a, **k):
        b = sig.bind(*a, **k); b.apply_defaults()
        if isinstance(b.arguments.get("seed"), (int, np.integer)): b.arguments["seed"] = int(b.arguments["seed"]) + 7919 * off[0]
        return f(*b.args, **b.kwargs)
    return g
for n, f in _gens.items():
    setattr(synth, n, wrap(f))
parity = sanity = runs = 0
for r in range(1, rounds + 1):
    off[0] = r
    for name in NAMES:
        fn = getattr(T, name)
        params = [{}]
        for m in getattr(fn, "pytestmark", []):
            if m.name == "parametrize":
                keys = [k.strip() for k in m.args[0].split(",")]
                params = [dict(zip(keys, v if isinstance(v, tuple) else (v,))) for v in m.args[1]]
        for p in params:
            kw = dict(p)
            for a in inspect.signature(fn).parameters:
                if a == "oracle": kw[a] = orc
                elif a == "fe": kw[a] = fe
                elif a == "ctx": kw[a] = ctx
            runs += 1
            try:
                fn(**kw)
            except AssertionError:
                tb = traceback.extract_tb(sys.exc_info()[2])[-1]
                line = tb.line or ""
                is_sanity = ("array_equal" not in line and "==" not in line) or line.strip().startswith("assert on >") or "> " in line and "array_equal" not in line and "==" not in line
                if is_sanity: sanity += 1
                else: parity += 1
                print("round", r, name, p, "line", tb.lineno, "|", line.strip()[:150], "| SANITY" if is_sanity else "| PARITY", flush=True)
            except Exception as e:
                parity += 1
                print("round", r, name, p, "EXC", repr(e)[:200], flush=True)
    print("round", r, "done: runs", runs, "parity failures", parity, "sanity failures", sanity, flush=True)
print("fuzz_match: %d runs, %d parity failures, %d sanity-only failures" % (runs, parity, sanity))
sys.exit(1 if parity else 0)
