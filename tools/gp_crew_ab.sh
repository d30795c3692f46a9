#!/bin/bash
# GPU box: crew rounds (crew.hip.hpp) on / off on the chained circuits, then the parity tests that drive single-workgroup jobs
cd "$GRAFT_REPO_ROOT"
P="ecne_circomlib_tests/Poseidon@poseidon.r1cs"
S="ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs"
EP="ecne_circomlib_tests/EdDSAPoseidonVerifier@eddsaposeidon.r1cs"
B="ecne_circomlib_tests/BabyPbk@babyjub.r1cs"
for f in "$P" "$S" "$EP" "$B" secp; do
  for c in 1 0; do
    echo -n "crew=$c  "; ECNE_CREW=$c timeout 300 python tools/solve_case.py "$f" 0 2>&1 | grep -a "dev_ms" | sed 's/.*rows/rows/' | cut -c1-110
  done
done
