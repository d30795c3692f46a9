#!/bin/bash
# GPU box: the round-6 evidence in one go -> gpurun_out/final_r06/   (python tools/collect_final_r06.py files it under profiles/)
# usage: bash tools/gp_final_r06.sh [quick]      quick: bench lines + kernel trace + PMC passes at S = 26 only
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final_r06
mkdir -p $O
timeout 600 python bench.py > $O/ecdsa.json 2> $O/ecdsa.err
for w in suite poseidon secp dag many; do timeout 900 python bench.py --workload $w --steps 5 --warmup 2 > $O/$w.json 2> $O/$w.err; done
timeout 900 python bench.py --S 104 --steps 5 --warmup 2 --cpu-sample-S 26 > $O/ecdsa_S104.json 2> $O/ecdsa_S104.err
bash tools/profile_r06.sh 26 > $O/profile_S26.log 2>&1
cp gpurun_out/prof_r06_S26/*.txt gpurun_out/prof_r06_S26/*.json $O/ 2>/dev/null
if [ "$1" != "quick" ]; then
timeout 1200 python bench.py --S 416 --steps 3 --warmup 1 --no-cpu-baseline --no-cold > $O/ecdsa_S416.json 2> $O/ecdsa_S416.err
bash tools/profile_r06.sh 104 > $O/profile_S104.log 2>&1
for f in trace fetch write sq; do cp gpurun_out/prof_r06_S104/$f.txt $O/S104_$f.txt 2>/dev/null; done
cp gpurun_out/prof_r06_S104/bench_under_rocprof.json $O/S104_bench_under_rocprof.json 2>/dev/null
timeout 600 python bench.py --job-lines on --no-cpu-baseline --no-cold > $O/ecdsa_with_job_lines.json 2> $O/ecdsa_with_job_lines.err
timeout 600 python tools/dag_side_ab.py 26 15 > $O/dag_side_ab.txt 2>&1
timeout 2400 python tests/tools/scale_variants.py > $O/scale_variants.txt 2>&1
timeout 900 python tests/tools/per_file_vs_oracle.py > $O/per_file_vs_oracle.txt 2>&1
timeout 600 python tests/tools/soak_determinism.py 26 60 > $O/soak_determinism.txt 2>&1
timeout 600 python tools/suite_stats.py 3 > $O/suite_per_file.txt 2>&1
for i in 1 2 3 4 5; do timeout 300 python tools/classify_time.py 104 2>/dev/null | tail -1; done > $O/classify_time.txt
timeout 300 python tools/classify_time.py 26 2>/dev/null | tail -1 >> $O/classify_time.txt
fi
for f in ecdsa suite poseidon secp dag many ecdsa_S104 ecdsa_S416; do tail -c 250 $O/$f.json 2>/dev/null; echo; done
tail -4 $O/scale_variants.txt 2>/dev/null | cut -c1-200; tail -2 $O/per_file_vs_oracle.txt 2>/dev/null
