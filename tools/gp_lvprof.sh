#!/bin/bash
cd "$GRAFT_REPO_ROOT"
cp ecneproject_amd/libecne_hip.so /tmp/base.so
ECNE_BUILD_FLAGS="-DECNE_LVPROF" python -m ecneproject_amd.build --force > /tmp/b.log 2>&1 || { tail /tmp/b.log; exit 1; }
python tools/lv_stages.py "ecne_circomlib_tests/Poseidon@poseidon.r1cs" "ecne_circomlib_tests/EdDSAMiMCSpongeVerifier@eddsamimcsponge.r1cs" "ecne_circomlib_tests/EdDSAPoseidonVerifier@eddsaposeidon.r1cs" secp
cp /tmp/base.so ecneproject_amd/libecne_hip.so
