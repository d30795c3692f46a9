#!/bin/bash
# GPU box: instruction counters of k_solve on one single-workgroup solve (20 solves of the file)   usage: gp_pmc_file.sh <fixture relpath>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_file
rm -rf $O; mkdir -p $O
cat > /tmp/solve20.py <<PY
import sys, os
sys.path.insert(0, "$GRAFT_REPO_ROOT"); sys.path.insert(0, "$GRAFT_REPO_ROOT/tests")
import ecneproject_amd as E
from gpu_common import build_system
s = build_system(sys.argv[1])
for _ in range(20): r = E.solve_batch([s], fetch_states=False)[0]
print("pops", r.summary.pops, "rounds", r.summary.rule_hits[13], "ms", r.summary.device_ms)
PY
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace -d $O/insts -- python /tmp/solve20.py "$1" 2> $O/insts.err | tail -1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $O/cyc -- python /tmp/solve20.py "$1" 2> $O/cyc.err | tail -1
for d in insts cyc; do python tools/rocpd_summary.py $(find $O/$d -name "*.db") 2>&1 | grep -E "k_solve|counter" | head -12; done
