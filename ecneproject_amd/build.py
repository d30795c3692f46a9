"""Builds libecne_hip.so (gfx950) in-tree with hipcc. Run: python -m ecneproject_amd.build"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libecne_hip.so")
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))   # every file under csrc/ is a dependency


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(os.path.dirname(HERE), "include", "ecne.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return SO
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-pthread", "-shared", "-fPIC",
           "-o", SO, os.path.join(CSRC, "ecne_engine.hip")] + os.environ.get("ECNE_BUILD_FLAGS", "").split()   # e.g. -DECNE_POPPROF (developer builds)
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv)
