"""lab/probes/coop_phases.py against a profiling build of the library (lab/build_dbg.sh coopprof -DMPST_COOP_PROF)."""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpstime_jl_amd as mt
mt._lib.LIB_PATH = os.environ["MPST_LIB"]
os.environ["MPST_BIG_SYNC"] = "1"
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lab", "probes", "coop_phases.py"), run_name="__main__")
