#!/bin/bash
# GPU box: A/B of one compile-time flag on the bench solve (two builds, alternating runs on the same box)   usage: gp_ab_flag.sh -DFLAG
cd "$GRAFT_REPO_ROOT"
cp ecneproject_amd/libecne_hip.so /tmp/base.so
ECNE_BUILD_FLAGS="$1" python -m ecneproject_amd.build --force > /tmp/b.log 2>&1 || { tail /tmp/b.log; exit 1; }
cp ecneproject_amd/libecne_hip.so /tmp/flag.so
for i in 1 2 3; do
  cp /tmp/base.so ecneproject_amd/libecne_hip.so; echo -n "base  "; timeout 300 python tools/solve_case.py ecdsa 0 2>&1 | grep -a "dev_ms" | cut -c70-90
  cp /tmp/flag.so ecneproject_amd/libecne_hip.so; echo -n "$1  "; timeout 300 python tools/solve_case.py ecdsa 0 2>&1 | grep -a "dev_ms" | cut -c70-90
done
cp /tmp/base.so ecneproject_amd/libecne_hip.so
