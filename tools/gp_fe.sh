# developer aid (GPU box): the whole -m gpu suite through each front-end, then file -> verdict timings
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/suite_default.log; tail -3 gpurun_out/suite_default.log
ECNE_FRONTEND=device ECNE_FULL_ORACLE=0 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/suite_device.log; tail -25 gpurun_out/suite_device.log
for fe in host device; do timeout 300 python tools/e2e_timing.py 26 1 $fe 2>&1 | tail -12; done
