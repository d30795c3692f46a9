#!/bin/bash
# GPU box: stage clocks of the fast wavefront round on single-workgroup solves, with chain bursts switched off (every level is a fast round)
cd "$GRAFT_REPO_ROOT"
cp ecneproject_amd/libecne_hip.so /tmp/base.so
ECNE_BUILD_FLAGS="-DECNE_W2PROF -DECNE_CHAIN_BURST_C=0" python -m ecneproject_amd.build --force > /tmp/b.log 2>&1 || { tail /tmp/b.log; exit 1; }
python tools/w2_stages_file.py "ecne_circomlib_tests/Poseidon@poseidon.r1cs" "ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs"
ECNE_BUILD_FLAGS="-DECNE_W2PROF" python -m ecneproject_amd.build --force > /tmp/b.log 2>&1 || { tail /tmp/b.log; exit 1; }
python tools/w2_stages_file.py "ecne_circomlib_tests/Poseidon@poseidon.r1cs" "ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs"
cp /tmp/base.so ecneproject_amd/libecne_hip.so
