"""Developer aid (GPU box): what the FIRST upload of a process costs, with and without torch in the process, with and without a
warm-up copy.   python tools/fe_first_upload.py [torch] [warm]"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
if "torch" in sys.argv:
    import torch
    torch.cuda.set_device(0)
    if "warm" in sys.argv:
        t = time.perf_counter(); x = torch.empty(64 << 20, dtype=torch.uint8).cuda(); torch.cuda.synchronize(); print("torch warm-up copy %.1f ms" % ((time.perf_counter() - t) * 1e3))
import ecneproject_amd as E, ecdsa_like
E.set_host_threads(0 if "threads" in sys.argv else 1)
p = ecdsa_like.cached(26, 10, directory="/tmp/ecne_bench_%d" % os.getuid())
for i in range(3):
    t = time.perf_counter(); f = E.R1CS(p); dt = (time.perf_counter() - t) * 1e3
    st = E.frontend_stats()
    print(sys.argv[1:], "load %d: %.1f ms wall, upload %.1f parse %.1f" % (i, dt, st["upload_ms"], st["parse_ms"]))
    del f
