#!/bin/bash
# GPU box: which rows of a case reach the general executor from the chain executor (round log build)   usage: gp_secp_gen.sh secp
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cp ecneproject_amd/libecne_hip.so /tmp/libecne_hip.so.keep
ECNE_BUILD_FLAGS="-DECNE_ROUNDLOG -DECNE_FINE_TICKS" python -m ecneproject_amd.build --force > /tmp/build.log 2>&1 || { tail -20 /tmp/build.log; exit 1; }
timeout 600 python tools/solve_case.py "$1" 0 > gpurun_out/gen_$1.txt 2>&1 || true
grep "^RG" gpurun_out/gen_$1.txt | awk '{print $5, $9, $11}' | sort | uniq -c | sort -rn | head -40
grep -c "^RG" gpurun_out/gen_$1.txt
cp /tmp/libecne_hip.so.keep ecneproject_amd/libecne_hip.so
python tools/shape_hist.py secp256k1.r1cs
