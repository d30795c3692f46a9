"""ctypes binding of the CPU parity oracle (oracle/libecne_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (ecneproject_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

ORDER_JULIA, ORDER_ASCENDING, ORDER_RANDOM = 0, 1, 2

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617

STATUS_NAMES = {0: "OK", -1: "EFORMAT", -2: "EBOUNDS", -3: "EDIVZERO", -4: "EUNDEF_DSU",
                -5: "EKEY", -6: "EDETSIZE", -7: "EIO", -10: "ECAPACITY", -12: "EWATCHDOG"}


class Summary(C.Structure):
    _fields_ = [("status", C.c_int32), ("verdict", C.c_int32),
                ("n_vars", C.c_int64), ("n_rows_main", C.c_int64), ("n_rows_reduced", C.c_int64),
                ("n_specials", C.c_int64),
                ("unique_nontrivial", C.c_int64), ("n_nontrivial", C.c_int64),
                ("unique_targets", C.c_int64), ("n_targets", C.c_int64),
                ("successful_steps", C.c_int64), ("outer_iterations", C.c_int64),
                ("pops", C.c_int64), ("num_unique", C.c_int64),
                ("rule_hits", C.c_int64 * 16),
                ("alg_bytes_pops", C.c_int64), ("alg_bytes_sweep", C.c_int64),
                ("nnz_reduced", C.c_int64), ("n_bad_rows", C.c_int64),
                ("t_read", C.c_double), ("t_abstract", C.c_double), ("t_solve", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ORACLE_DIR, "libecne_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_run.restype = C.c_void_p
        L.orc_run.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                              C.c_int, C.c_int, C.c_uint64, C.c_int]
        L.orc_run_io.restype = C.c_void_p
        L.orc_run_io.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int,
                                 C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.orc_get_summary.argtypes = [C.c_void_p, C.POINTER(Summary)]
        L.orc_get_states.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_get_bad_rows.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_special_count.restype = C.c_int64
        L.orc_special_count.argtypes = [C.c_void_p]
        L.orc_special_get.restype = C.c_int64
        L.orc_special_get.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_int64, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.orc_read_info.restype = C.c_int
        L.orc_read_info.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
                                    C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]
        L.orc_julia_order.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.orc_fp_op.restype = C.c_int
        L.orc_fp_op.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def limbs_to_int(a):
    """(..., 4) uint64 little-endian limbs -> python ints (object array or scalar)."""
    a = np.asarray(a, dtype=np.uint64)
    if a.ndim == 1:
        return sum(int(a[i]) << (64 * i) for i in range(4))
    return [limbs_to_int(r) for r in a]


def int_to_limbs(x):
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


class OracleResult:
    """Everything the sequential restatement knows after a run."""

    def __init__(self, summary, flags, lb, ub, abz, nvalues, values, bad_rows, specials):
        self.summary = summary
        self.status = summary.status
        self.verdict = bool(summary.verdict)
        self.flags = flags            # uint8 per variable (index v-1): bit0 unique, bit1 is_known
        self.lb, self.ub = lb, ub     # (nVars, 4) uint64
        self.abz = abz                # int64
        self.nvalues, self.values = nvalues, values   # uint8, (nVars, 2, 4) uint64
        self.bad_rows = bad_rows      # 1-based rows holding a non-unique variable
        self.specials = specials      # [(name, inputs, outputs)]

    @property
    def unique(self):
        return (self.flags & 1).astype(bool)

    @property
    def is_known(self):
        return ((self.flags >> 1) & 1).astype(bool)

    def counts(self):
        s = self.summary
        return (s.unique_nontrivial, s.n_nontrivial, s.unique_targets, s.n_targets)

    def alg_bytes(self):
        """B_alg of SURVEY.md §8(d): pops term + (3*outer_iterations + 1) sweeps."""
        s = self.summary
        return s.alg_bytes_pops + (3 * s.outer_iterations + 1) * s.alg_bytes_sweep


def run(main_path, trusted=(), names=(), secp_solve=False, policy=ORDER_JULIA, seed=0,
        shuffle_queue=False, want_states=True, known_variables=None, target_variables=None):
    """known_variables / target_variables: SolveConstraintsSymbolic's own arguments (:583-592) instead of what readR1CS returns"""
    L = lib()
    n = len(trusted)
    tp = (C.c_char_p * max(n, 1))(*[os.fsencode(t) for t in trusted])
    tn = (C.c_char_p * max(n, 1))(*[s.encode() for s in names])
    if known_variables is not None or target_variables is not None:
        assert policy == ORDER_JULIA and not shuffle_queue
        kn = None if known_variables is None else np.asarray(list(known_variables), np.int64)
        tg = None if target_variables is None else np.asarray(list(target_variables), np.int64)
        h = L.orc_run_io(os.fsencode(main_path), n, tp, tn, int(secp_solve),
                         None if kn is None else max(kn.ctypes.data, 8), 0 if kn is None else len(kn),
                         None if tg is None else max(tg.ctypes.data, 8), 0 if tg is None else len(tg))
    else:
        h = L.orc_run(os.fsencode(main_path), n, tp, tn, int(secp_solve), int(policy), int(seed),
                      int(shuffle_queue))
    try:
        s = Summary()
        L.orc_get_summary(h, C.byref(s))
        nv = max(int(s.n_vars), 0)
        flags = np.zeros(nv, np.uint8)
        lb = np.zeros((nv, 4), np.uint64)
        ub = np.zeros((nv, 4), np.uint64)
        abz = np.zeros(nv, np.int64)
        nvalues = np.zeros(nv, np.uint8)
        values = np.zeros((nv, 2, 4), np.uint64)
        bad = np.zeros(int(s.n_bad_rows), np.int64)
        specials = []
        if s.status == 0 and want_states:
            L.orc_get_states(h, flags.ctypes.data, lb.ctypes.data, ub.ctypes.data, abz.ctypes.data,
                             nvalues.ctypes.data, values.ctypes.data)
            L.orc_get_bad_rows(h, bad.ctypes.data)
        for i in range(L.orc_special_count(h)):
            name = C.create_string_buffer(256)
            ins = np.zeros(4096, np.int64)
            outs = np.zeros(4096, np.int64)
            nout = C.c_int64()
            nin = L.orc_special_get(h, i, name, 256, ins.ctypes.data, 4096, outs.ctypes.data, 4096,
                                    C.byref(nout))
            specials.append((name.value.decode(), ins[:nin].tolist(), outs[:nout.value].tolist()))
        return OracleResult(s, flags, lb, ub, abz, nvalues, values, bad, specials)
    finally:
        L.orc_free(h)


def read_info(path):
    L = lib()
    info = np.zeros(8, np.int64)
    kn = np.zeros(1 << 16, np.int64)
    out = np.zeros(1 << 16, np.int64)
    nk, no = C.c_int64(), C.c_int64()
    nnz = np.zeros(3, np.int64)
    st = L.orc_read_info(os.fsencode(path), info.ctypes.data, kn.ctypes.data, kn.size, C.byref(nk),
                         out.ctypes.data, out.size, C.byref(no), nnz.ctypes.data)
    if st != 0:
        return st, None
    keys = ["nWires", "nPubOut", "nPubIn", "nPrvIn", "nLabels", "nConstraints", "nVars", "fieldSize"]
    d = dict(zip(keys, info.tolist()))
    d["knowns"] = kn[:min(nk.value, kn.size)].tolist()
    d["n_knowns"] = nk.value
    d["outputs"] = out[:min(no.value, out.size)].tolist()
    d["n_outputs"] = no.value
    d["nnz"] = nnz.tolist()
    return 0, d


def julia_order(keys, mode=0):
    L = lib()
    k = np.asarray(keys, np.int64)
    out = np.zeros(len(set(keys)), np.int64)
    L.orc_julia_order(k.ctypes.data, len(k), mode, out.ctypes.data)
    return out.tolist()


def fp_op(op, a, b=0):
    L = lib()
    x, y = int_to_limbs(a), int_to_limbs(b)
    out = np.zeros(4, np.uint64)
    st = L.orc_fp_op(op, x.ctypes.data, y.ctypes.data, out.ctypes.data)
    return st, limbs_to_int(out)
