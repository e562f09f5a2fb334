import cProfile, pstats, sys, runpy, io, re
sys.argv = ["tools/profile_config5.py", "nbody", "30"]
pr = cProfile.Profile()
pr.enable()
runpy.run_path("tools/profile_config5.py", run_name="__main__")
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s).sort_stats("cumulative")
st.print_stats("montecosmo_amd|growth|background", 60)
print(s.getvalue()[:12000])
