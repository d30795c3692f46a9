# developer aid (GPU box): one bench line per BASELINE configuration -> gpurun_out/bench_r03/*.json
O=gpurun_out/bench_r03
mkdir -p $O
timeout 600 python bench.py > $O/ecdsa.json 2> $O/ecdsa.err; tail -c 400 $O/ecdsa.json; echo
for w in suite poseidon secp dag; do timeout 900 python bench.py --workload $w --steps 5 --warmup 2 > $O/$w.json 2> $O/$w.err; tail -c 300 $O/$w.json; echo; tail -3 $O/$w.err; done
