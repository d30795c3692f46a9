#!/bin/bash
# GPU box: per-round log of one solve_case.py configuration   usage: gp_roundlog_case.sh secp
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cp ecneproject_amd/libecne_hip.so /tmp/libecne_hip.so.keep
ECNE_BUILD_FLAGS="-DECNE_ROUNDLOG -DECNE_FINE_TICKS" python -m ecneproject_amd.build --force > /tmp/build.log 2>&1 || { tail -20 /tmp/build.log; exit 1; }
N=$(basename "$1" | tr "@." "__")
timeout 600 python tools/solve_case.py "$1" 0 > gpurun_out/roundlog_$N.txt 2>&1 || true
python tools/round_log.py gpurun_out/roundlog_$N.txt --seq > gpurun_out/roundlog_${N}_summary.txt 2>&1 || true
head -11 gpurun_out/roundlog_${N}_summary.txt
cp /tmp/libecne_hip.so.keep ecneproject_amd/libecne_hip.so
