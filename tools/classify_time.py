"""Developer aid (GPU box): k_classify_rows on ecdsa_like(S): HIP-event time over repeated calls and GB/s."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ecneproject_amd as E, fixtures, ecdsa_like
E.set_host_threads(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 26
s = E.System(E.R1CS(ecdsa_like.cached(S, 10))); s.abstract(E.R1CS(fixtures.path("secp256k1.r1cs")), "Secp256k1AddUnequal")
ts = []
for _ in range(6):
    shape, ms, nbytes = E.classify(s)
    ts.append(ms)
print("rows", len(shape), "bytes", nbytes, "ms per call", [round(t, 3) for t in ts], "best %.3f ms = %.0f GB/s" % (min(ts), nbytes / min(ts) / 1e6))
