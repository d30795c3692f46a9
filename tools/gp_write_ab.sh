#!/bin/bash
# GPU box: WRITE_SIZE / FETCH_SIZE of k_solve_team per launch for this tree's library and the builds under old_build*/ (same bench command)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
cp ecneproject_amd/libecne_hip.so /tmp/new.so
for b in new old_build old_build2; do
  if [ $b = new ]; then cp /tmp/new.so ecneproject_amd/libecne_hip.so; else [ -f $b/ecneproject_amd/libecne_hip.so ] || continue; cp $b/ecneproject_amd/libecne_hip.so ecneproject_amd/libecne_hip.so; fi
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf gpurun_out/wab; rocprofv3 --pmc $c --kernel-trace -d gpurun_out/wab -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/wab.err
    echo -n "$b $c: "; python tools/rocpd_summary.py $(find gpurun_out/wab -name "*.db") 2>/dev/null | grep -a "k_solve_team" | grep -a "$c" | awk '{print $(NF)}'
  done
done
cp /tmp/new.so ecneproject_amd/libecne_hip.so; rm -rf gpurun_out/wab
