#!/bin/bash
# GPU box: stage clocks of the level rounds on the master of a team (ecdsa_like(26)), -DECNE_LVPROF build
cd "$GRAFT_REPO_ROOT"
cp ecneproject_amd/libecne_hip.so /tmp/base.so
ECNE_BUILD_FLAGS="-DECNE_LVPROF" python -m ecneproject_amd.build --force > /tmp/b.log 2>&1 || { tail /tmp/b.log; exit 1; }
python tools/lv_stages.py ecdsa:26
cp /tmp/base.so ecneproject_amd/libecne_hip.so
