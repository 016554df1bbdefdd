"""W1 batch (1 024 slices x 2 000 raw events): step time with the sparse pipeline the dispatcher picks (gather_form 0) against the slot
form forced (gather_form 4).  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from eorb_slam_amd import frontend as fe, synth


def main():
    W, H, B, NEV = 240, 180, int(os.environ.get("B", "1024")), int(os.environ.get("NEV", "2000"))
    recs = bench.gen_slices(NEV, W, H, [2 + b for b in range(B)], True, 1)
    ev = np.concatenate(recs); del recs
    fb = fe.FrontEndBatch(W, H, 1.0, False, 400, 1.0, 1, 0, 0, 9, max_batch=B, max_events=NEV, want_desc=False, match=False) if "want_desc" in fe.FrontEndBatch.__init__.__code__.co_varnames else fe.FrontEndBatch(W, H, 1.0, False, 400, 1.0, 1, 0, 0, 9, max_batch=B, max_events=NEV)
    c, cap = fb.ctx, fb.cap
    mx, my = synth.undistort_lut(W, H)
    fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=c)
    d_ev = c.dev_alloc(ev.nbytes); c.upload(d_ev, ev)
    d_img = c.dev_alloc(B * W * H); d_kp = c.dev_alloc(B * cap * 28); d_desc = c.dev_alloc(B * cap * 32); d_n = c.dev_alloc(B * 4); d_m = c.dev_alloc(B * cap * 4); d_nm = c.dev_alloc(B * 4)
    off = np.arange(B + 1, dtype=np.int64) * NEV
    run = lambda: fb.run_dev(d_ev, off, d_img, d_kp, d_desc, d_n, d_m, d_nm, raw=True)
    ref = None
    for form in (0, 4, 0, 4):
        c.debug_option("gather_form", form)
        for _ in range(3): run()
        c.sync()
        t0 = time.perf_counter()
        for _ in range(20): run()
        c.sync()
        dt = (time.perf_counter() - t0) / 20
        img = np.zeros(B * W * H, np.uint8); c.download(img, d_img)
        if ref is None: ref = img
        print("gather_form %d: %.3f ms per step of %d x %d events (%.0f frames/s)  %s" % (form, dt * 1e3, B, NEV, B / dt, "same images" if np.array_equal(ref, img) else "IMAGES DIFFER"), flush=True)
    c.debug_option("gather_form", 0)


if __name__ == "__main__":
    main()
