#!/bin/bash
# GPU box: soak of the -DECNE_JITTER build (libecne_hip_jitter.so: pseudo-random delays in front of every barrier arrival and behind every
# release, a random start per workgroup, helpers held back past the master's first commands; job_barrier.hip.hpp). Fuzz systems on forced
# teams of 0 / 2 / 5 / 8 / 24 workgroups (stress_fuzz runs all five), the long decompositions, the circomlib suite as one batch (team + side
# kernel), the split family barrier, repeated solves of the bench system; several seeds. Every line of the log ends in ok / FAIL; an
# ECNE_ETIMEOUT would show as a status difference against the oracle.
#   ECNE_BUILD_SO=libecne_hip_jitter.so ECNE_BUILD_FLAGS=-DECNE_JITTER python -m ecneproject_amd.build --force      (here, before gpurun)
cd "$GRAFT_REPO_ROOT"
export ECNE_LIB=libecne_hip_jitter.so
L=gpurun_out/soak_jitter.log
mkdir -p gpurun_out; : > $L
run() { echo "== $*" | tee -a $L; timeout -s KILL 900 "$@" 2>&1 | tail -7 | tee -a $L; echo "rc=${PIPESTATUS[0]}" | tee -a $L; }
for js in 1 2 3; do
  export ECNE_JITTER_SEED=$js
  run python tests/tools/stress_fuzz.py $((310000 + js * 1000)) 150 1
  run python tests/tools/stress_fuzz.py $((320000 + js * 1000)) 120 0
  run python tests/tools/stress_fuzz.py $((330000 + js * 1000)) 40 4
done
export ECNE_JITTER_SEED=7
ECNE_DRAIN=2 run python tests/tools/stress_fuzz.py 341000 60 4
run python -m pytest tests/test_gpu_parity.py tests/test_gpu_jobs.py tests/test_gpu_split.py tests/test_gpu_team_loop.py tests/test_gpu_drain.py tests/test_gpu_chain.py -x -q
export ECNE_JITTER_SEED=11
run python -m pytest tests/test_gpu_ecdsa_like.py tests/test_gpu_long_r4.py tests/test_gpu_soak.py -x -q
run python tests/tools/soak_determinism.py 26 40
run python tests/tools/soak_side.py 20
