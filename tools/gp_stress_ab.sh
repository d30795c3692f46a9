#!/bin/bash
# GPU box: the stress fuzz (every frontier drained) over and over on this tree's library and on old_build/'s   usage: gp_stress_ab.sh [runs] [seed]
cd "$GRAFT_REPO_ROOT"
N=${1:-10}; S=${2:-92000}
for i in $(seq $N); do
  echo -n "NEW $i: "; ECNE_DRAIN=2 timeout 600 python tests/tools/stress_fuzz.py $S 60 4 2>&1 | grep -a "FAIL" | head -3 | cut -c1-900; echo
  echo -n "OLD $i: "; AB_PKG=old_build ECNE_DRAIN=2 timeout 600 python tests/tools/stress_fuzz.py $S 60 4 2>&1 | grep -a "FAIL" | head -3 | cut -c1-900; echo
done
