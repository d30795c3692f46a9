#!/bin/bash
# GPU box: bench.py lines (ms per step, kernel ms) of some workloads under compile-time defines, the base build first
# usage: gp_ab_flags_bench.sh "ecdsa dag" "-DX=1" ["-DX=2" ...]
cd "$GRAFT_REPO_ROOT"
W=$1; shift
cp ecneproject_amd/libecne_hip.so /tmp/base.so
run() {
  for w in $W; do
    if [ $w = ecdsa ]; then A=""; else A="--workload $w"; fi
    for i in 1 2; do timeout 300 python bench.py $A --no-cpu-baseline --no-cold 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $w', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; done
  done
}
run base
for F in "$@"; do
  ECNE_BUILD_FLAGS="$F" python -m ecneproject_amd.build --force > /tmp/b.log 2>&1 || { tail /tmp/b.log; exit 1; }
  run "$F"
done
cp /tmp/base.so ecneproject_amd/libecne_hip.so
