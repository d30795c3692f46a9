#!/bin/bash
# GPU box: soak of the final build -- the GPU suite over and over, stress fuzz on fresh seeds (default and every-frontier-drained), repeated solves
cd "$GRAFT_REPO_ROOT"
fail=0
for i in $(seq 8); do timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -1 | tee -a gpurun_out/soak.log; done
for s in 71000 72000 73000; do timeout 900 python tests/tools/stress_fuzz.py $s 150 1 2>&1 | tail -2 | tee -a gpurun_out/soak.log; done
for s in 81000 82000; do ECNE_DRAIN=2 timeout 900 python tests/tools/stress_fuzz.py $s 60 4 2>&1 | tail -2 | tee -a gpurun_out/soak.log; done
timeout 1200 python tests/tools/repeat_solves.py 150 2>&1 | tail -2 | tee -a gpurun_out/soak.log
