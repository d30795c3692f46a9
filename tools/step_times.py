"""Developer aid (GPU box): wall time of every pass of a jobs.Runner workload (bench.py's job lists), to see outliers and drifts.
python tools/step_times.py suite|dag|many|secp|poseidon [passes]"""
import sys, time, os, argparse
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import torch, bench
from ecneproject_amd import jobs as J
w = sys.argv[1] if len(sys.argv) > 1 else "suite"
args = argparse.Namespace(workload=w, copies=8, S=26, stride=10, host_threads=1)
jl, label, data = bench.workload_jobs(w, args)
r = J.Runner(jl, 0, 1, 0, None)
st = torch.cuda.current_stream().cuda_stream
out = []
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    t = time.perf_counter(); res, ok = r.run(stream=st); torch.cuda.synchronize(); out.append(((time.perf_counter() - t) * 1e3, res[0].summary.device_ms))
print(w, " ".join("%.2f/%.2f" % x for x in out))
