"""Run the A* / decode GPU parity tests against an alternative build of the library (A/B builds:
stress builds): PF_LIB=path/to/lib.so python scripts/run_parity_with_lib.py"""
import os, sys
sys.path[:0] = ["maaco-path-planing_amd", "tests", "oracle"]
from pathfit import _lib
_lib._SO = os.path.abspath(os.environ["PF_LIB"])
import pytest
sys.exit(pytest.main(["tests/test_gpu_parity.py", "-m", "gpu", "-x", "-q", "-k", "golden_both or random_512 or sealed or decode"]))
