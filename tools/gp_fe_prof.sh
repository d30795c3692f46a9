# developer aid (GPU box): phase timeline (ECNE_FE_DEBUG) and rocprofv3 kernel trace of file -> verdict through the device front-end
mkdir -p gpurun_out/prof_fe
ECNE_FE_DEBUG=1 timeout 300 python tools/e2e_timing.py 26 1 device > gpurun_out/prof_fe/timeline.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fe/trace -- python tools/e2e_timing.py 26 1 device > gpurun_out/prof_fe/e2e_under_rocprof.txt 2> gpurun_out/prof_fe/trace.err
python tools/rocpd_summary.py $(find gpurun_out/prof_fe/trace -name "*.db") > gpurun_out/prof_fe/trace.txt 2>&1
find gpurun_out/prof_fe -name "*.db" -delete
cat gpurun_out/prof_fe/timeline.txt; head -50 gpurun_out/prof_fe/trace.txt
