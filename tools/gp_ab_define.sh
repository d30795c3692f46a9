#!/bin/bash
# GPU box: kernel times of a few workloads under one compile-time define   usage: gp_ab_define.sh "-DX=.." [more flag sets...]
cd "$GRAFT_REPO_ROOT"
cp ecneproject_amd/libecne_hip.so /tmp/base.so
echo "== base"; python tools/variant_times.py 2>&1 | tail -3; python tools/solve_case.py ecdsa:6 0 2>&1 | grep -a dev_ms | sed "s/.*dev_ms/ecdsa_like(6) dev_ms/" | cut -c1-40
for F in "$@"; do
  ECNE_BUILD_FLAGS="$F" python -m ecneproject_amd.build --force > /tmp/b.log 2>&1 || { tail /tmp/b.log; exit 1; }
  echo "== $F"; python tools/variant_times.py 2>&1 | tail -3; python tools/solve_case.py ecdsa:6 0 2>&1 | grep -a dev_ms | sed "s/.*dev_ms/ecdsa_like(6) dev_ms/" | cut -c1-40
done
cp /tmp/base.so ecneproject_amd/libecne_hip.so
