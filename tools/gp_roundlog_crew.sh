#!/bin/bash
# GPU box: per-call log of the crew / level rounds of one circuit, crew rounds on and off   usage: gp_roundlog_crew.sh <fixture relpath>
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cp ecneproject_amd/libecne_hip.so /tmp/libecne_hip.so.keep
ECNE_BUILD_FLAGS="-DECNE_ROUNDLOG -DECNE_FINE_TICKS" python -m ecneproject_amd.build --force > /tmp/build.log 2>&1 || { tail -20 /tmp/build.log; exit 1; }
for c in 1 0; do ECNE_CREW=$c timeout 600 python tools/solve_case.py "$1" 0 > gpurun_out/rlcrew_$c.txt 2>&1 || true; done
cp /tmp/libecne_hip.so.keep ecneproject_amd/libecne_hip.so
