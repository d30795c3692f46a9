#!/bin/bash
# GPU box: soak of the -DECNE_JITTER build (libecne_hip_jitter.so: pseudo-random delays in front of every barrier arrival and behind every
# release, a random start per workgroup, helpers held back past the master's first commands; job_barrier.hip.hpp). Fuzz systems on forced
# teams of 0 / 2 / 5 / 8 / 24 workgroups (stress_fuzz runs all five), the long decompositions, the circomlib suite as one batch (team + side
# kernel), the split family barrier, repeated solves of the bench system; several seeds. A barrier that does not complete ends in
# ECNE_ETIMEOUT = a status difference against the oracle (and an "ECNE TIMEOUT" line from the device).
# Then the PRODUCT library under rocprofv3: 12 kernel traces + 8 TCC counter passes of `bench.py --workload suite` (the pass that once ended in
# "HW Exception ... GPU Hang", profiles/README.md), each under a kill timeout.
#   ECNE_BUILD_SO=libecne_hip_jitter.so ECNE_BUILD_FLAGS=-DECNE_JITTER python -m ecneproject_amd.build --force      (here, before gpurun)
cd "$GRAFT_REPO_ROOT"
export ECNE_LIB=libecne_hip_jitter.so
L=gpurun_out/soak_jitter.log
mkdir -p gpurun_out; : > $L
run() { echo "== $*" | tee -a $L; timeout -s KILL 900 "$@" 2>&1 | tail -7 | tee -a $L; echo "rc=${PIPESTATUS[0]}" | tee -a $L; }
for js in 1 2 3 4 5 6; do
  export ECNE_JITTER_SEED=$js
  run python tests/tools/stress_fuzz.py $((410000 + js * 1000)) 150 1
  run python tests/tools/stress_fuzz.py $((420000 + js * 1000)) 120 0
  run python tests/tools/stress_fuzz.py $((430000 + js * 1000)) 40 4
  ECNE_DRAIN=2 run python tests/tools/stress_fuzz.py $((440000 + js * 1000)) 40 4
done
export ECNE_JITTER_SEED=7
run python -m pytest tests/test_gpu_parity.py tests/test_gpu_jobs.py tests/test_gpu_split.py tests/test_gpu_team_loop.py tests/test_gpu_drain.py tests/test_gpu_chain.py -x -q
export ECNE_JITTER_SEED=11
run python -m pytest tests/test_gpu_ecdsa_like.py tests/test_gpu_long_r4.py tests/test_gpu_soak.py tests/test_gpu_level.py tests/test_gpu_crew.py -x -q
run python tests/tools/soak_determinism.py 26 40
run python tests/tools/soak_side.py 20
unset ECNE_LIB ECNE_JITTER_SEED
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  rm -rf /tmp/jt_trace
  ( cd $R && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d /tmp/jt_trace -- python bench.py --workload suite --steps 5 --warmup 2 --no-cpu-baseline > /tmp/jt_bench.json 2> /tmp/jt_err.txt ); rc=$?
  echo "suite kernel trace $i: rc=$rc $(grep -c 'HW Exception' /tmp/jt_err.txt) hw exceptions, ms_per_step $(python -c "import json;print(round(json.load(open('/tmp/jt_bench.json'))['ms_per_step'],3))" 2>/dev/null)" | tee -a $R/$L
done
for i in 1 2 3 4 5 6 7 8; do
  rm -rf /tmp/jt_trace
  ( cd $R && timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/jt_trace -- python bench.py --workload suite --steps 5 --warmup 2 --no-cpu-baseline > /tmp/jt_bench.json 2> /tmp/jt_err.txt ); rc=$?
  echo "suite TCC pass $i: rc=$rc $(grep -c 'HW Exception' /tmp/jt_err.txt) hw exceptions" | tee -a $R/$L
done
