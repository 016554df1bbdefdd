import os, sys, subprocess
sys.path.insert(0, os.getcwd())
import numpy as np
if len(sys.argv) > 1:
    from eorb_slam_amd import frontend as fe, synth
    W, H = 240, 180
    mx, my = synth.undistort_lut(W, H)
    c = fe.Context()
    fe.EvImConverter.set_undistort_maps(np.ascontiguousarray(mx), np.ascontiguousarray(my), True, ctx=c)
    out = {}
    for n in (50, 100, 500, 5000, 90000):
        raw = synth.random_raw_events(n, W, H, seed=9)
        out[str(n)] = fe.EvImConverter.ev2im_gauss_raw(raw, W, H, 1.0, False, False, ctx=c)
    np.savez(sys.argv[1], **out)
else:
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "/tmp/new.npz"], env=env)
    env["EORB_OLD_RAW_GATHER"] = "1"
    subprocess.check_call([sys.executable, __file__, "/tmp/old.npz"], env=env)
    a = np.load("/tmp/new.npz"); b = np.load("/tmp/old.npz")
    for k in a.files:
        d = a[k].view(np.uint32) != b[k].view(np.uint32)
        ys, xs = np.nonzero(d)
        print(k, "mismatches", d.sum(), "tiles", sorted(set(zip((ys // 8).tolist(), (xs // 8).tolist())))[:10])
        for y, x in list(zip(ys, xs))[:6]:
            print("   ", y, x, a[k][y, x], b[k][y, x])
