#!/bin/bash
# GPU box: k_classify_rows under library variants (var_*.so at the repo root, built here with ECNE_BUILD_FLAGS) and ECNE_CLS_BLOCKS, S = 104
cd "$GRAFT_REPO_ROOT"
cp ecneproject_amd/libecne_hip.so /tmp/new.so
S=${1:-104}
for so in /tmp/new.so var_*.so; do
  cp $so ecneproject_amd/libecne_hip.so
  for b in 4096 1024 16384; do
    echo -n "$so blocks=$b: "; ECNE_CLS_BLOCKS=$b timeout 300 python tools/classify_time.py $S 2>/dev/null | tail -1 | cut -c1-200
  done
done
cp /tmp/new.so ecneproject_amd/libecne_hip.so
