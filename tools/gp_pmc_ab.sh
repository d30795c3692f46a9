#!/bin/bash
# GPU box: a set of PMC counters of k_solve_team per launch for this tree's library and the builds under old_build*/   usage: gp_pmc_ab.sh "CTR1 CTR2 ..."
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp ecneproject_amd/libecne_hip.so /tmp/new.so
for b in new old_build old_build2; do
  if [ $b = new ]; then cp /tmp/new.so ecneproject_amd/libecne_hip.so; else [ -f $b/ecneproject_amd/libecne_hip.so ] || continue; cp $b/ecneproject_amd/libecne_hip.so ecneproject_amd/libecne_hip.so; fi
  rm -rf gpurun_out/wab; rocprofv3 --pmc $1 --kernel-trace -d gpurun_out/wab -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/wab.err
  echo "== $b"; python tools/rocpd_summary.py $(find gpurun_out/wab -name "*.db") 2>/dev/null | grep -a "k_solve_team" | tail -n +2 | awk '{print $(NF-3), $(NF)}'
done
cp /tmp/new.so ecneproject_amd/libecne_hip.so; rm -rf gpurun_out/wab
