#!/bin/bash
# GPU box: A/B of this tree's library against old_build/ (a copy of an older commit, built) on the bench solve and the scale variants, alternating runs on the same box
cd "$GRAFT_REPO_ROOT"
cp ecneproject_amd/libecne_hip.so /tmp/new.so
cp old_build/ecneproject_amd/libecne_hip.so /tmp/old.so
for i in 1 2 3; do
  cp /tmp/old.so ecneproject_amd/libecne_hip.so; echo -n "old  "; timeout 300 python tools/solve_case.py ecdsa 0 2>&1 | grep -a "dev_ms" | cut -c60-130
  cp /tmp/new.so ecneproject_amd/libecne_hip.so; echo -n "new  "; timeout 300 python tools/solve_case.py ecdsa 0 2>&1 | grep -a "dev_ms" | cut -c60-130
done
cp /tmp/new.so ecneproject_amd/libecne_hip.so
