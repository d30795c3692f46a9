"""Developer aid (GPU box): where the host side of `bench.py --workload many` goes -- cProfile over 20 passes of jobs.Runner.run on the 504 jobs."""
import cProfile, os, pstats, sys, time
import torch
torch.cuda.init()
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ecneproject_amd as E
from ecneproject_amd import jobs as J
import fixtures
import bench
E.warmup(0)
class A: copies = 8; S = 26; stride = 10
jobs, _, _ = bench.workload_jobs("many", A)
r = J.Runner(jobs, E=E)
for _ in range(3): r.run()
t = time.perf_counter()
for _ in range(20): res, ok = r.run()
dt = (time.perf_counter() - t) / 20
print("ms per pass %.3f, kernel ms (sum of launches' device_ms of the longest jobs) n/a, ok %s" % (dt * 1e3, ok))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): r.run()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
for label, st in (("torch loaded, stream None", None), ("torch loaded, torch's current stream", torch.cuda.current_stream().cuda_stream)):
    for _ in range(3): r.run(stream=st)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): r.run(stream=st)
    torch.cuda.synchronize(); print(label, "ms per pass %.3f" % ((time.perf_counter() - t) / 20 * 1e3), "stream handle", st)
import gc
gc.collect(); gc.disable()
keep = []
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): keep.append(r.run(stream=torch.cuda.current_stream().cuda_stream)[0])
torch.cuda.synchronize(); print("5 steps, results kept, gc off: ms per pass %.3f" % ((time.perf_counter() - t) / 5 * 1e3))
