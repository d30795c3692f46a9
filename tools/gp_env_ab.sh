#!/bin/bash
# GPU box: bench.py lines of some workloads with an environment switch off / on, alternating   usage: gp_env_ab.sh "dag ecdsa" ECNE_SIDE_XCD 0 1 [reps]
cd "$GRAFT_REPO_ROOT"
W=$1; V=$2; A0=$3; A1=$4; N=${5:-2}
for i in $(seq $N); do
  for a in $A0 $A1; do
    for w in $W; do
      if [ $w = ecdsa ]; then A=""; else A="--workload $w"; fi
      env $V=$a timeout 300 python bench.py $A --no-cpu-baseline --no-cold 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$a $w', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), d['config'].get('verdicts_true'))"
    done
  done
done
