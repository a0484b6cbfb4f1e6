"""Host-side profile (cProfile) of the config-5 harness with 8 realisations: where the wall time outside the kernels goes."""
import cProfile, pstats, sys, io, runpy
sys.argv = ["tools/gpu_checks/many_realizations_fullsize.py", "8", "hip"]
pr = cProfile.Profile()
pr.enable()
runpy.run_path("tools/gpu_checks/many_realizations_fullsize.py", run_name="__main__")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
print(s.getvalue()[:7000])
