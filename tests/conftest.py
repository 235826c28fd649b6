import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "oracle"), ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.exists("/root/reference/astar.py")
    skip_ref = pytest.mark.skip(reason="/root/reference not present (GPU box)")
    for it in items:
        if "reference" in it.keywords and not have_ref:
            it.add_marker(skip_ref)
