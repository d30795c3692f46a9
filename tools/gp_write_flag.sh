#!/bin/bash
# GPU box: WRITE_SIZE of k_solve_team per launch and the solve time, this tree's library against a build of it with one more compile-time flag   usage: gp_write_flag.sh -DFLAG
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp ecneproject_amd/libecne_hip.so /tmp/base.so
ECNE_BUILD_FLAGS="$1" python -m ecneproject_amd.build --force > /tmp/b.log 2>&1 || { tail /tmp/b.log; exit 1; }
cp ecneproject_amd/libecne_hip.so /tmp/flag.so
for b in base flag base flag; do
  cp /tmp/$b.so ecneproject_amd/libecne_hip.so
  rm -rf gpurun_out/wab; rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/wab -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/wab.err
  echo -n "$b WRITE_SIZE KB: "; python tools/rocpd_summary.py $(find gpurun_out/wab -name "*.db") 2>/dev/null | grep -a "k_solve_team" | grep -a "WRITE_SIZE" | awk '{print $(NF)}'
  echo -n "$b "; timeout 300 python tools/solve_case.py ecdsa 0 2>&1 | grep -a "dev_ms" | cut -c70-90
done
cp /tmp/base.so ecneproject_amd/libecne_hip.so; rm -rf gpurun_out/wab
