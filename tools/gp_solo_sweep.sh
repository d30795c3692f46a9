#!/bin/bash
# GPU box: sweep of the solo-drain trigger (build variants into scratch copies)
cd "$GRAFT_REPO_ROOT"
cp ecneproject_amd/libecne_hip.so /tmp/keep.so
for v in "32 8" "16 8" "8 4" "6 4" "6 2"; do
  set -- $v
  ECNE_BUILD_FLAGS="-DECNE_SOLO_AVAIL=$1 -DECNE_SOLO_RATIO=$2" python -m ecneproject_amd.build --force > /tmp/b.log 2>&1 || { tail /tmp/b.log; exit 1; }
  echo "== avail>=$1 ratio $2"
  timeout 300 python tools/solve_case.py ecdsa 0 2>&1 | grep -a "dev_ms" | cut -c60-140
  timeout 300 python tools/solve_file.py eddsa:3 0 0 2>&1 | tail -1 | cut -c1-110
  timeout 300 python tools/solve_file.py eddsa:45 0 0 2>&1 | tail -1 | cut -c1-110
  timeout 300 python tools/solve_case.py secp 0 2>&1 | grep -a "dev_ms" | cut -c1-90
done
cp /tmp/keep.so ecneproject_amd/libecne_hip.so
