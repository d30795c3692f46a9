# Developer aid (GPU box): the round-3 profiles -> gpurun_out/prof_r03/*.txt (summaries; the .db files are deleted)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/prof_r03
mkdir -p $O
# 1. counter calibration with kernels of known traffic
rocprofv3 --kernel-trace --stats -d $O/cal_trace -- tools/micro/copy > $O/cal_stdout.txt 2> $O/cal_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/cal_fetch -- tools/micro/copy > /dev/null 2> $O/cal_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/cal_write -- tools/micro/copy > /dev/null 2> $O/cal_write.err
# 2. the device front-end (steady state: three trips)
rocprofv3 --kernel-trace --stats -d $O/fe_trace -- python tools/fe_steady.py 26 > $O/fe_stdout.txt 2> $O/fe_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fe_fetch -- python tools/fe_steady.py 26 > /dev/null 2> $O/fe_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/fe_write -- python tools/fe_steady.py 26 > /dev/null 2> $O/fe_write.err
for d in cal_trace cal_fetch cal_write fe_trace fe_fetch fe_write; do python tools/rocpd_summary.py $(find $O/$d -name "*.db") > $O/$d.txt 2>&1; done
find $O -name "*.db" -delete
cat $O/cal_stdout.txt; head -8 $O/cal_fetch.txt; head -8 $O/cal_write.txt; cat $O/fe_stdout.txt | tail -4; head -30 $O/fe_trace.txt
