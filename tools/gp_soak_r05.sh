#!/bin/bash
# GPU box: soak of the round-5 build -- stress fuzz on fresh seeds (forced teams of 2 / 5 / 8 / 24: outer iterations on the master alone,
# incremental P3 / P4 passes), the same with every frontier drained, the GPU suite three times, repeated solves of the bench system
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; : > gpurun_out/soak_r05.log
for s in 171000 172000 173000 174000; do timeout 900 python tests/tools/stress_fuzz.py $s 150 1 2>&1 | tail -6 | tee -a gpurun_out/soak_r05.log; done
for s in 181000 182000 183000; do ECNE_DRAIN=2 timeout 900 python tests/tools/stress_fuzz.py $s 60 4 2>&1 | tail -6 | tee -a gpurun_out/soak_r05.log; done
for s in 191000 192000; do timeout 900 python tests/tools/stress_fuzz.py $s 60 4 2>&1 | tail -6 | tee -a gpurun_out/soak_r05.log; done
for s in 201000 202000 203000; do timeout 900 python tests/tools/stress_fuzz.py $s 300 0 2>&1 | tail -6 | tee -a gpurun_out/soak_r05.log; done
for i in 1 2 3; do timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -1 | tee -a gpurun_out/soak_r05.log; done
timeout 900 python tests/tools/soak_determinism.py 26 100 2>&1 | tail -2 | tee -a gpurun_out/soak_r05.log
timeout 900 python tests/tools/soak_crew.py 40 2>&1 | tail -2 | tee -a gpurun_out/soak_r05.log
