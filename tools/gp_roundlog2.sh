#!/bin/bash
# GPU box: per-round log of one solve_file.py case (ROUNDLOG build into a scratch copy of the .so, then the normal build again)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cp ecneproject_amd/libecne_hip.so /tmp/libecne_hip.so.keep
ECNE_BUILD_FLAGS="-DECNE_ROUNDLOG -DECNE_FINE_TICKS" python -m ecneproject_amd.build --force > /tmp/build.log 2>&1 || { tail -20 /tmp/build.log; exit 1; }
timeout 600 python tools/solve_file.py "$@" > gpurun_out/roundlog2.txt 2>&1 || true
cp /tmp/libecne_hip.so.keep ecneproject_amd/libecne_hip.so
grep -a "^RL" gpurun_out/roundlog2.txt | awk '{k=$2; c[k]++; t[k]+=$10; r[k]+=$8} END {for (k in c) printf "%s rounds %d rows %d ms %.2f\n", k, c[k], r[k], t[k]*1e-5}'
grep -a "^RL" gpurun_out/roundlog2.txt | sed -n 2000,2060p | awk '{printf "%s a%s n%s c%s %dus | ", substr($2,1,2), $4, $6, $8, $10/100} END {print ""}'
