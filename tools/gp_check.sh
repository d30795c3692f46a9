# developer aid (GPU box): parity on all fixture configs (default + forced 4 workgroups), ecdsa_like tests, bench, suite
timeout 300 python tests/tools/gpu_parity_debug.py > gpurun_out/par_a.log 2>&1; tail -1 gpurun_out/par_a.log
ECNE_FORCE_NWG=4 timeout 300 python tests/tools/gpu_parity_debug.py > gpurun_out/par_b.log 2>&1; tail -1 gpurun_out/par_b.log
timeout 300 python -m pytest tests/test_gpu_ecdsa_like.py -m gpu -x -q 2>&1 | tail -2
timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_now.json; python -c "
import json; d=json.load(open('gpurun_out/bench_now.json')); r=d['roofline']; print(d['ms_per_step'], r['phase_ms'], d['config']['verdict'], d['config']['pops']); print(r['queue_ms']); print(r['multi_ms'])"
timeout 200 python tests/tools/suite_bench.py 2>&1 | tail -1
