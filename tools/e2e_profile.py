import cProfile, pstats, sys, io
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import e2e_predict
pr = cProfile.Profile()
pr.enable()
e2e_predict.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
print(s.getvalue()[:6000])
