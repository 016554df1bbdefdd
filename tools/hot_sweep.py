"""Step time of the w2 batch (128 x 1 Mev, raw events) over the register-row kernel's list-length threshold and its number of
wavefronts (test hooks slot_hot_min / slot_hot_waves): where the two gather kernels balance.  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from eorb_slam_amd import frontend as fe, synth

def main():
    W, H, B, NEV = 240, 180, int(os.environ.get("B", "128")), 1000000
    recs = bench.gen_slices(NEV, W, H, [2 + b for b in range(B)], True, 1)
    ev = np.concatenate(recs); del recs
    fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=B, max_events=NEV)
    c, cap = fb.ctx, fb.cap
    mx, my = synth.undistort_lut(W, H)
    fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=c)
    d_ev = c.dev_alloc(ev.nbytes); c.upload(d_ev, ev)
    d_img = c.dev_alloc(B * W * H); d_kp = c.dev_alloc(B * cap * 28); d_desc = c.dev_alloc(B * cap * 32); d_n = c.dev_alloc(B * 4); d_m = c.dev_alloc(B * cap * 4); d_nm = c.dev_alloc(B * 4)
    off = np.arange(B + 1, dtype=np.int64) * NEV
    run = lambda: fb.run_dev(d_ev, off, d_img, d_kp, d_desc, d_n, d_m, d_nm, raw=True)
    ref = None
    print("hot_min hot_waves  ms/step  hot lists", flush=True)
    for hw in (1536, 2048, 3072):
        for hm in (1000, 2000, 4000, 6000, 8000, 12000, 16000, 24000):
            c.debug_option("slot_hot_min", hm); c.debug_option("slot_hot_waves", hw)
            for _ in range(3): run()
            c.sync()
            t0 = time.perf_counter()
            for _ in range(10): run()
            c.sync()
            dt = (time.perf_counter() - t0) / 10
            img = np.zeros(B * W * H, np.uint8); c.download(img, d_img)
            if ref is None: ref = img
            print("%7d %9d  %7.3f  %6d  %s" % (hm, hw, dt * 1e3, c.debug_counter("slot_hot_items"), "same images" if np.array_equal(ref, img) else "IMAGES DIFFER"), flush=True)


if __name__ == "__main__":
    main()
