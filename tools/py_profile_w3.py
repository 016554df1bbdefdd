"""cProfile of bench.py's w3 step (one frame per call through ctypes): where the binding's time goes.  Run on the GPU box."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--workload", "w3", "--cpu-slices", "0", "--cpu-pool", "0", "--no-prof", "--steps", "30"]
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
