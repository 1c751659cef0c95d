#!/usr/bin/env python3
"""cProfile of the on-the-fly MD example: where an update step spends its time.
usage: python3 tools/otf_profile.py [md_nvt_otf args]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["md_nvt_otf.py"] + sys.argv[1:]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "md_nvt_otf.py"), run_name="__main__")
finally:
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
