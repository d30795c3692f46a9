# developer aid (GPU box): parity + timing after a schedule change
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ecdsa_like.py tests/test_fuzz.py tests/test_gpu_fastround.py tests/test_bigrows.py tests/test_gpu_chain.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python tests/tools/scale_variants.py 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['case'], d['kernel_ms'], 'ms', d['rounds'], 'rounds', d['fast_wave_rounds'], 'fast', d['rows_in_fast_rounds'], 'rows', d['fast_ms'], 'ms; multi', d['multi_workgroup_rounds'], d['multi_ms'])
    else: print(l.rstrip())"
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['latency_model'])"
