#!/bin/bash
# GPU box: this tree's library against old_build.so (an older commit's build, copied to the repo root -- git-ignored, it travels with the
# snapshot) on the job workloads, alternating runs on the same box.   usage: bash tools/gp_ab_old.sh "secp poseidon" [reps]
cd "$GRAFT_REPO_ROOT"
W=${1:-"secp poseidon"}; N=${2:-2}
cp ecneproject_amd/libecne_hip.so /tmp/new.so
for i in $(seq $N); do
  for which in old new; do
    if [ $which = old ]; then cp old_build.so ecneproject_amd/libecne_hip.so; else cp /tmp/new.so ecneproject_amd/libecne_hip.so; fi
    for w in $W; do
      if [ $w = ecdsa ]; then A=""; else A="--workload $w"; fi
      timeout 300 python bench.py $A --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which $w', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"
    done
  done
done
cp /tmp/new.so ecneproject_amd/libecne_hip.so
