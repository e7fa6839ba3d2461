import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--train", "--steps", "20", "--warmup", "3"]
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
pr = cProfile.Profile(); pr.enable()
bench.main()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
