"""Per-role busy cycles per batch of ev_gather on the hot tiles.  Needs a library built with -DEORB_DIAG
(EORB_FE_LIB=<that .so> python tools/gather_diag.py); knock-out variants: -DEORB_KO_VALS / -DEORB_KO_ADDS."""
import ctypes as C, os, sys, subprocess, json
sys.path.insert(0, os.getcwd())
import numpy as np
from eorb_slam_amd import frontend, synth, _lib
L=_lib.lib()
W,H,B,N=240,180,int(os.environ.get("DIAG_B","8")),1000000
raw="--raw" in sys.argv
pairs=[synth.shapes_events(N,W,H,seed=2+b,motion=0.5,undistort=True,return_raw=True) for b in range(B)]
fb=frontend.FrontEndBatch(W,H,1.0,False,max_batch=B,max_events=N)
c=fb.ctx
if raw:
    mx,my=synth.undistort_lut(W,H); frontend.EvImConverter.set_undistort_maps(mx,my,True,ctx=c)
    ev16=np.concatenate([p[1] for p in pairs])
else:
    ev16=np.concatenate([frontend.pack_events(p[0]) for p in pairs])
d=c.dev_alloc(ev16.nbytes); c.upload(d,ev16)
offs=np.arange(B+1,dtype=np.int64)*N
for it in range(3):
    fb.run_dev(d,offs,raw=raw); c.sync()
    out=(C.c_ulonglong*16)()
    L.eorb_diag_read(out)
    o=list(out)
    for r,name in enumerate(["adds(w0)","setup+vals(w1)","vals(w2+)"]):
        w,t,nb,st=o[r*4:r*4+4]
        if nb: print(name,"work/batch %.0f"%(w/nb),"total/batch %.0f"%(t/nb),"setup/batch %.0f"%(st/nb),"batches",nb)
