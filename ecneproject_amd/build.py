"""Builds libecne_hip.so (gfx950) in-tree with hipcc. Run: python -m ecneproject_amd.build"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, os.environ.get("ECNE_BUILD_SO", "libecne_hip.so"))      # (developer builds beside the product: ECNE_BUILD_SO=libecne_hip_jitter.so ECNE_BUILD_FLAGS=-DECNE_JITTER)
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


TUS = ["ecne_engine.hip", "ecne_frontend.hip"]      # translation units (everything else under csrc/ is a header of one or both)


def build(force=False, verbose=True):
    """hipcc --offload-arch=gfx950: the two translation units are compiled side by side, then linked into libecne_hip.so."""
    if not force and not needs_build():
        return SO
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-pthread", "-fPIC"] + os.environ.get("ECNE_BUILD_FLAGS", "").split()   # e.g. -DECNE_POPPROF (developer builds)
    objdir = os.path.join(HERE, "build" if "ECNE_BUILD_SO" not in os.environ else "build_" + os.path.splitext(os.environ["ECNE_BUILD_SO"])[0])
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for tu in TUS:
        obj = os.path.join(objdir, tu.replace(".hip", ".o"))
        cmd = [hipcc()] + flags + ["-c", os.path.join(CSRC, tu), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, obj, subprocess.Popen(cmd)))
    for cmd, obj, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", SO] + [obj for _c, obj, _p in procs]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv)
