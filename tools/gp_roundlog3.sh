#!/bin/bash
# GPU box: per-round log of one solve_case.py case (ROUNDLOG build into a scratch copy of the .so, then the normal build again)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cp ecneproject_amd/libecne_hip.so /tmp/libecne_hip.so.keep
ECNE_BUILD_FLAGS="-DECNE_ROUNDLOG -DECNE_FINE_TICKS" python -m ecneproject_amd.build --force > /tmp/build.log 2>&1 || { tail -20 /tmp/build.log; exit 1; }
timeout 600 python tools/solve_case.py "$@" > gpurun_out/roundlog3.txt 2>&1 || true
cp /tmp/libecne_hip.so.keep ecneproject_amd/libecne_hip.so
python tools/round_log.py gpurun_out/roundlog3.txt --seq | cut -c1-6000
