"""include/ecne.h, ecne_result_digest -- restated in numpy on per-variable arrays (`flags`, `abz`, `nvalues`, `lb`, `ub`, `values` as the
engine's SolveResult and the oracle's OracleResult both carry them). Test infrastructure: tests/test_gpu_soak.py checks the device
digest against it on fetched states; tests/golden/make_scale_goldens.py digests the ORACLE's state of the million-row cases with it, so
that the `-m gpu` suite compares the engine's whole state there in seconds instead of minutes of oracle."""
import numpy as np

M64 = (1 << 64) - 1


def _mix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def numpy_digest(g):
    with np.errstate(over="ignore"):
        nv = len(g.flags)
        h = _mix(np.arange(1, nv + 1, dtype=np.uint64))
        nvl = g.nvalues.astype(np.uint64)
        h = _mix(h ^ (g.flags.astype(np.uint64) & np.uint64(3)))
        h = _mix(h ^ g.abz.astype(np.int32).view(np.uint32).astype(np.uint64))
        h = _mix(h ^ nvl)
        for k in range(4):
            h = _mix(h ^ g.lb[:, k])
        for k in range(4):
            h = _mix(h ^ g.ub[:, k])
        vals = g.values.reshape(nv, 8)
        for k in range(8):
            use = np.minimum(nvl, np.uint64(2)) * np.uint64(4) > np.uint64(k)
            h = np.where(use, _mix(h ^ vals[:, k]), h)
        s0 = int(np.sum(h, dtype=np.uint64))
        s1 = int(np.sum(_mix(h ^ np.uint64(0xA5A5A5A5A5A5A5A5)), dtype=np.uint64))
    return (s0 & M64, s1 & M64)
