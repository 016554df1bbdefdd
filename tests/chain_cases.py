"""Shared inputs of the L1-chain tests (CPU: tests/test_oracle_chain.py, GPU: tests/test_gpu_chain.py) and of bench.py's w1full
workload: a synthetic event stream with enough motion for the window-size rule to fire, and stand-ins for the optimisers' poses."""
import numpy as np

from eorb_slam_amd import synth

CAM = (synth.EVETHZ_K["fx"], synth.EVETHZ_K["fy"], synth.EVETHZ_K["cx"], synth.EVETHZ_K["cy"])       # Pinhole (rectified events)


def stream(n_chunks=60, chunk=2000, seed=5, motion=14.0, W=240, H=180):
    """One long time-ordered slice of the shapes generator (float EventData, monotone time stamps, 1 us apart)."""
    return synth.shapes_events(n_chunks * chunk, W, H, seed=seed, motion=motion, undistort=True)


def mci_poses(window):
    """What resolveLastDPose / resolveLastPoseMap / resolveLastAtt2Params would hand generateMCImage: fixed small motions scaled by the
    window's length (the optimisers themselves are outside the front end).  Deterministic in the window."""
    n = len(window)
    s = min(n / 6000.0, 2.0)
    return dict(dp=dict(angle=0.010 * s, axis=(0.1, -0.2, 0.97), t=(0.004 * s, -0.003 * s, 0.001), medDepth=1.0),
                ba=dict(angle=0.016 * s, axis=(-0.3, 0.1, 0.95), t=(-0.002 * s, 0.005 * s, 0.0), medDepth=1.3),
                se2=np.array([0.012 * s, 1.5 * s, -0.8 * s], np.float32))
