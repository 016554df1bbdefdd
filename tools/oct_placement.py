"""Where the octree's working set sits for the configurations the benches use (eorb_debug_counter oct_*): LDS only or partly in global scratch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eorb_slam_amd import frontend as fe

for (W, H, nf, sf, nl, edge, name) in ((240, 180, 400, 1.0, 1, 9, "w1 L1 detect"), (240, 180, 800, 1.0, 1, 9, "w1 L2 detect"), (240, 180, 1000, 1.2, 4, 19, "w2/w3"),
                                       (346, 260, 2000, 1.2, 8, 19, "w4"), (752, 480, 1000, 1.2, 8, 19, "EuRoC")):
    e = fe.ORBextractor(nf, sf, nl, 10 if nl > 1 else 0, 0, edge, (W, H))
    c = e.ctx
    dc = c.debug_counter("oct_direct_cap")
    print("%-14s lds_only(single|batch)=%d  lds %6d B  scratch %6d B  direct passes up to %d | %d candidates per level" % (name, c.debug_counter("oct_lds_only"), c.debug_counter("oct_lds_bytes"), c.debug_counter("oct_scratch_bytes"), dc & 0xffff, dc >> 16), flush=True)
