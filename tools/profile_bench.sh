# Developer aid (GPU box): rocprofv3 kernel trace + the two PMC passes of bench.py, summarised into gpurun_out/prof_r01b/*.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/prof_r01b
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01b/trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r01b/bench_trace.json 2> gpurun_out/prof_r01b/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_r01b/fetch -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_r01b/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_r01b/write -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_r01b/write.err
for d in trace fetch write; do python tools/rocpd_summary.py $(find gpurun_out/prof_r01b/$d -name "*.db") > gpurun_out/prof_r01b/$d.txt 2>&1; done
tail -1 gpurun_out/prof_r01b/bench_trace.json | head -c 400; echo
head -20 gpurun_out/prof_r01b/trace.txt; head -12 gpurun_out/prof_r01b/fetch.txt; head -12 gpurun_out/prof_r01b/write.txt
find gpurun_out/prof_r01b -name "*.db" -size +20M -delete
