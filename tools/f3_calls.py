"""call latency (through ctypes, host buffers) of a KeyFrame-side matcher entry; run on the GPU box: python tools/f3_calls.py"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from eorb_slam_amd import frontend as fe, synth
from oracle import oracle_py as orc
W,H=240,180
ctx=fe.Context()
rng=np.random.default_rng(0)
def frame(seed):
    img=synth.texture_image(W,H,seed=seed)
    oe=orc.OrbExtractor(1000,1.2,4,10,0,edgeTh=19)
    _,k,d,_=oe.extract(img)
    return k,d
k1,d1=frame(1); k2,d2=frame(2)
def timed(fn, reps=200):
    for _ in range(20): fn()
    ts=[]
    for _ in range(reps):
        t=time.perf_counter(); fn(); ts.append(time.perf_counter()-t)
    ts.sort(); return ts[len(ts)//2]*1e3
M=2000
pick=rng.integers(0,len(k2),M)
uv=np.stack([k2["x"][pick]+3,k2["y"][pick]-3],axis=1).astype(np.float32); level=k2["octave"][pick].astype(np.int32)
sc6=(1.2**np.arange(6)).astype(np.float32); radius=(3.0*sc6[level]).astype(np.float32); valid=np.ones(M,np.uint8)
gb=fe.grid_bounds(W,H)
qd=d2[pick].copy()
print("KeyFrameRadiusMatch 2000 map points: %.4f ms per call" % timed(lambda: fe.KeyFrameRadiusMatch(k1,d1,gb,valid,uv,radius,level,qd,ctx=ctx)))
