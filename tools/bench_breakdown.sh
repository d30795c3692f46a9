# Developer aid (GPU box): bench.py without the CPU leg, printing the in-kernel phase / queue / multi-round breakdown
timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_now.json; python -c "
import json; d=json.load(open('gpurun_out/bench_now.json')); r=d['roofline']; print(d['ms_per_step'], r['phase_ms'], d['config']['verdict'], d['config']['pops']); print(r['queue_ms']); print(r['multi_ms'])"
