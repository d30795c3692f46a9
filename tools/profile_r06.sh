# (every pass under `timeout -s KILL`: one run of the round ended in "HW Exception ... GPU Hang" two seconds into the suite trace that follows the
#  TCC pass -- not reproduced by three runs of that command on a fresh box -- and rocprofv3 then sat in its signal handler until the box's limit)
# GPU box: rocprofv3 kernel trace + the PMC passes of bench.py for one workload size (separate runs per counter group:
# MI355X_MICROARCH.md "rocprofv3 PMC slots"; never --pmc together with the hip/hsa trace domains), summarised into gpurun_out/prof_r06_S$S/*.txt
# usage: bash tools/profile_r06.sh [S]     (S = 26: the headline workload; S = 104: the scale-out variant beyond the Infinity Cache)
S=${1:-26}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/prof_r06_S$S
mkdir -p $O
CMD="python bench.py --S $S --steps 5 --warmup 2 --no-cpu-baseline --no-cold"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $O/trace -- $CMD > $O/bench_under_rocprof.json 2> $O/trace.err
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -- $CMD > /dev/null 2> $O/fetch.err
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -- $CMD > /dev/null 2> $O/write.err
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $O/sq -- $CMD > /dev/null 2> $O/sq.err
if [ "$S" = "26" ]; then
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $O/suite -- python bench.py --workload suite --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_suite_under_rocprof.json 2> $O/suite.err
  timeout -s KILL 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace -d $O/insts -- $CMD > /dev/null 2> $O/insts.err
  timeout -s KILL 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/tcc -- $CMD > /dev/null 2> $O/tcc.err
fi
for d in trace fetch write sq insts tcc suite; do [ -d $O/$d ] && python tools/rocpd_summary.py $(find $O/$d -name "*.db") > $O/$d.txt 2>&1; done
tail -1 $O/bench_under_rocprof.json | head -c 300; echo
for d in trace fetch write sq; do echo "== $d"; head -8 $O/$d.txt; done
find $O -name "*.db" -delete
