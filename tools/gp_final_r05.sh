#!/bin/bash
# GPU box: the round-5 evidence in one go -> gpurun_out/final_r05/   (python tools/collect_final_r05.py files it under profiles/)
# usage: bash tools/gp_final_r05.sh [quick]      quick: bench lines + kernel trace + PMC passes at S = 26 only
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final_r05
mkdir -p $O
timeout 600 python bench.py > $O/ecdsa.json 2> $O/ecdsa.err
for w in suite poseidon secp dag many; do timeout 900 python bench.py --workload $w --steps 5 --warmup 2 > $O/$w.json 2> $O/$w.err; done
timeout 900 python bench.py --S 104 --steps 5 --warmup 2 --cpu-sample-S 26 > $O/ecdsa_S104.json 2> $O/ecdsa_S104.err
bash tools/profile_r05.sh 26 > $O/profile_S26.log 2>&1
cp gpurun_out/prof_r05_S26/*.txt gpurun_out/prof_r05_S26/*.json $O/ 2>/dev/null
if [ "$1" != "quick" ]; then
bash tools/profile_r05.sh 104 > $O/profile_S104.log 2>&1
for f in trace fetch write sq; do cp gpurun_out/prof_r05_S104/$f.txt $O/S104_$f.txt 2>/dev/null; done
cp gpurun_out/prof_r05_S104/bench_under_rocprof.json $O/S104_bench_under_rocprof.json 2>/dev/null
timeout 2400 python tests/tools/scale_variants.py > $O/scale_variants.txt 2>&1
timeout 900 python tests/tools/per_file_vs_oracle.py > $O/per_file_vs_oracle.txt 2>&1
bash tools/gp_roundlog.sh > /dev/null 2>&1
cp gpurun_out/roundlog_ecdsa_summary.txt $O/round_log_summary.txt 2>/dev/null
timeout 600 python tests/tools/soak_crew.py 60 > $O/soak_crew.txt 2>&1
timeout 600 python tests/tools/soak_determinism.py 26 60 > $O/soak_determinism.txt 2>&1
timeout 600 python tools/suite_stats.py 3 > $O/suite_per_file.txt 2>&1
for S in 26 104; do timeout 300 python tools/classify_time.py $S 2>/dev/null | tail -1; done > $O/classify_time.txt
fi
for f in ecdsa suite poseidon secp dag many ecdsa_S104; do tail -c 250 $O/$f.json; echo; done
tail -4 $O/scale_variants.txt 2>/dev/null | cut -c1-200; tail -2 $O/per_file_vs_oracle.txt 2>/dev/null
