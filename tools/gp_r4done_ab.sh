#!/bin/bash
# GPU box: long_r4_done (fastrow.hip.hpp) on / off, times and per-call round logs
cd "$GRAFT_REPO_ROOT"
S="ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs"
EP="ecne_circomlib_tests/EdDSAPoseidonVerifier@eddsaposeidon.r1cs"
for f in "$S" "$EP" secp; do for c in 1 0; do echo -n "r4done=$c  "; ECNE_R4DONE=$c timeout 300 python tools/solve_case.py "$f" 0 2>&1 | grep -a "dev_ms" | sed 's/.*rows/rows/' | cut -c1-110; done; done
cp ecneproject_amd/libecne_hip.so /tmp/libecne_hip.so.keep
ECNE_BUILD_FLAGS="-DECNE_ROUNDLOG -DECNE_FINE_TICKS" python -m ecneproject_amd.build --force > /tmp/build.log 2>&1 || { tail -20 /tmp/build.log; exit 1; }
for c in 1 0; do ECNE_R4DONE=$c timeout 600 python tools/solve_case.py "$S" 0 > gpurun_out/rlr4_$c.txt 2>&1 || true; done
cp /tmp/libecne_hip.so.keep ecneproject_amd/libecne_hip.so
