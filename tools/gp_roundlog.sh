#!/bin/bash
# GPU box: per-round log of the bench solve (ROUNDLOG build into a scratch copy of the .so, then the normal build again)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cp ecneproject_amd/libecne_hip.so /tmp/libecne_hip.so.keep
ECNE_BUILD_FLAGS="-DECNE_ROUNDLOG -DECNE_FINE_TICKS" python -m ecneproject_amd.build --force > /tmp/build.log 2>&1 || { tail -20 /tmp/build.log; exit 1; }
timeout 600 python tools/solve_case.py ecdsa 0 > gpurun_out/roundlog_ecdsa.txt 2>&1 || true
python tools/round_log.py gpurun_out/roundlog_ecdsa.txt --seq > gpurun_out/roundlog_ecdsa_summary.txt 2>&1 || true
grep -a "LANECUT" gpurun_out/roundlog_ecdsa.txt | sort | uniq -c | sort -rn | head -30
cp /tmp/libecne_hip.so.keep ecneproject_amd/libecne_hip.so
