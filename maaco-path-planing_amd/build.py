"""Build libpathfit.so (hand-written HIP for gfx950) in-tree with hipcc.

    python maaco-path-planing_amd/build.py [--force]

hipcc cross-compiles gfx950 without a GPU; the .so travels to the GPU box with
the repo snapshot.  PF_EXTRA_FLAGS adds compiler flags: -DPF_TWO_WAVE compiles the
two-wavefronts-per-search engine (csrc/pf_astar_pr.h; off by default, measured 0.90x),
-DPF_TRACE / -DPF_STAMPS / -DPF_WALK_PROBE the diagnostic builds of scripts/.  -ffp-contract=off: the reference's arithmetic is unfused
IEEE double and bit-exact path parity depends on it.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "pathfit.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(HERE, "csrc"))) if f.endswith(".h")] + \
       [os.path.join(os.path.dirname(HERE), "include", "pathfit.h")]
OUT = os.path.join(HERE, "lib", "libpathfit.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


def build(force=False, verbose=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = [HIPCC] + FLAGS + os.environ.get("PF_EXTRA_FLAGS", "").split() + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
