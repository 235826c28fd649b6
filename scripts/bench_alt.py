"""Run bench.py against an alternative build of the library (experiments): python scripts/bench_alt.py <so> [bench args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maaco-path-planing_amd"))
from pathfit import _lib
_lib._SO = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
