#!/bin/bash
# GPU box: kernel traces of the job workloads (rocprofv3 --kernel-trace --stats, one pass each, under a kill timeout) -> gpurun_out/trace_jobs_r05/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/trace_jobs_r05
mkdir -p $O
for w in secp poseidon dag many; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/$w -- python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$w.json 2> $O/$w.err
  DB=$(find $O/$w -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB > $O/$w.txt 2>&1
  [ -n "$DB" ] && [ $w = dag ] && python tools/rocpd_timeline.py $DB k_solve 10 > $O/dag_timeline.txt 2>&1
  find $O/$w -name "*.db" -delete
  head -6 $O/$w.txt
done
cat $O/dag_timeline.txt
