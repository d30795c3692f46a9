#!/bin/bash
# GPU box: the round-3 evidence in one go -> gpurun_out/final_r03/
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final_r03
mkdir -p $O
bash tools/gp_bench_all.sh > $O/bench_all.log 2>&1
cp gpurun_out/bench_r03/*.json $O/ 2>/dev/null
bash tools/profile_bench.sh r03c > $O/profile.log 2>&1
cp gpurun_out/prof_r03c/*.txt gpurun_out/prof_r03c/*.json $O/ 2>/dev/null
timeout 2400 python tests/tools/scale_variants.py > $O/scale_variants.txt 2>&1
timeout 900 python tests/tools/per_file_vs_oracle.py > $O/per_file_vs_oracle.txt 2>&1
bash tools/gp_roundlog.sh > /dev/null 2>&1
cp gpurun_out/roundlog_ecdsa_summary.txt $O/round_log_summary.txt 2>/dev/null
tail -3 $O/bench_all.log; tail -5 $O/scale_variants.txt | cut -c1-300; tail -5 $O/per_file_vs_oracle.txt
