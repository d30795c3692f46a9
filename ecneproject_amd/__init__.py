"""ecneproject_amd — host-side mirror of Ecne's solver interface over libecne_hip (MI355X / gfx950).

The reference (franklynwang/EcneProject) is Julia; its seam for the hot path is three functions
(src/ParseR1CS.jl:50, src/R1CSConstraintSolver.jl:502 and :583).  Julia is not available in the
build image, so this Python module plays the role the Julia shim `julia/EcneHIP.jl` plays for a
Julia user: the same function names, argument meaning and error behaviour, on top of the C ABI of
include/ecne.h.  All propagation work happens in the HIP kernels; nothing here evaluates a rule.

    readR1CS(filename)                         -> (equations, known_vars, output_vars, nVars)
    SolveConstraintsSymbolic(constraints, special_constraints, known_variables, debug=False,
                             target_variables=[], num_variables=-1, input_sym="default.sym",
                             secp_solve=False) -> bool
    solveWithTrustedFunctions(input_r1cs, input_r1cs_name, trusted_r1cs=[], trusted_r1cs_names=[],
                              debug=False, printRes=True, abstractionOnly=False, input_sym="",
                              secp_solve=False) -> bool
"""
import ctypes as C
import os
import time

import numpy as np

from . import _lib
from ._lib import Info, Opts, Summary, SystemInfo

__all__ = ["readR1CS", "SolveConstraintsSymbolic", "solveWithTrustedFunctions", "solve_batch",
           "R1CS", "System", "SolveResult", "EcneError", "BoundsError", "DivideError", "UndefVarError",
           "device_count", "classify", "set_host_threads", "set_frontend", "frontend_stats", "set_split"]

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


# ---- the reference's exception types, as raised through the status codes of the C ABI
class EcneError(Exception):
    status = 0


class BoundsError(EcneError, IndexError):        # Julia BoundsError (:916, :762, :785)
    status = -2


class DivideError(EcneError, ZeroDivisionError):  # Julia DivideError (:919-920, :1467)
    status = -3


class UndefVarError(EcneError, NameError):       # Julia UndefVarError: dsu (:762), msg (:556-559)
    status = -4


class FormatError(EcneError, AssertionError):    # @assert in readR1CS (ParseR1CS.jl:58,62,69)
    status = -1


class AbstractionKeyError(EcneError, KeyError):  # KeyError (:381-382)
    status = -5


class DetSizeError(EcneError):
    status = -6


class NoDeviceError(EcneError, RuntimeError):
    status = -8


class DeviceBusyError(EcneError, TimeoutError):
    status = -11


class CapacityError(EcneError, MemoryError):
    """a device table overflowed or an allocation failed (ECNE_ECAPACITY)"""
    status = -10


class NoConvergenceError(EcneError, RuntimeError):
    """the propagation queue never drains on this input: the reference would not terminate (include/ecne.h ECNE_ENOCONVERGE)"""
    status = -12


_EXC = {-1: FormatError, -2: BoundsError, -3: DivideError, -4: UndefVarError, -5: AbstractionKeyError,
        -6: DetSizeError, -7: OSError, -8: NoDeviceError, -9: ValueError, -10: CapacityError, -11: DeviceBusyError, -12: NoConvergenceError}


def _check(st, what=""):
    if st == 0:
        return
    msg = _lib.lib().ecne_strerror(st).decode()
    raise _EXC.get(st, EcneError)("%s%s (ecne_status %d)" % (what + ": " if what else "", msg, st))


def device_count():
    return _lib.lib().ecne_device_count()


def warmup(device=0):
    """Pays a cold process's one-off costs up front (the HIP runtime's copy path, the library's code objects, the device's scratch memory
    for the solve kernels) instead of inside the first readR1CS / solve; returns the milliseconds it took. Optional."""
    ms = C.c_double()
    _check(_lib.lib().ecne_warmup(int(device), C.byref(ms)), "ecne_warmup")
    return float(ms.value)


def set_host_threads(n=0):
    """Opt in to host worker threads for parsing, abstraction and the flat-array layout (n <= 0: the cores present,
    at most 32). The library works on the calling thread unless asked. Returns the count now in effect."""
    return int(_lib.lib().ecne_set_host_threads(int(n)))


FRONTEND_HOST, FRONTEND_DEVICE, FRONTEND_AUTO = 0, 1, 2


def set_frontend(mode=-1):
    """Which front-end turns files into the solver's arrays: FRONTEND_HOST, FRONTEND_DEVICE (parse, abstraction and layout as
    kernels on the current HIP device), FRONTEND_AUTO (device from 100 000 constraints on; the default). mode < 0 only reads."""
    return int(_lib.lib().ecne_set_frontend(int(mode)))


def set_split(mode):
    """One file, several independent parts (include/ecne.h: ecne_set_split): 0 never, 1 from the second solve of a system on when the
    first took long enough -- and before the first solve of a file of many medium groups, counted on the device -- (the default), 2 at the
    first solve."""
    _check(_lib.lib().ecne_set_split(int(mode)))


def frontend_stats():
    """timing of the calling thread's last trip through the front-end (include/ecne.h: ecne_frontend_stats)"""
    a = (C.c_double * 16)()
    _check(_lib.lib().ecne_frontend_stats(a))
    k = ["parse_device", "upload_ms", "offsets_ms", "fill_ms", "parse_ms", "file_bytes", "abstract_device", "prep_ms", "fingerprint_ms",
         "scan_ms", "verify_ms", "compact_ms", "candidates", "matched", "layout_device", "layout_ms"]
    return dict(zip(k, list(a)))


class R1CS:
    """A parsed .r1cs file (the `equations` object of readR1CS)."""

    def __init__(self, path):
        self.path = os.fspath(path)
        self._h = C.c_void_p()
        _check(_lib.lib().ecne_r1cs_load(os.fsencode(self.path), C.byref(self._h)), self.path)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:          # (module globals are gone at interpreter shutdown)
            _lib.lib().ecne_r1cs_free(h)
            self._h = None

    @property
    def info(self):
        i = Info()
        _check(_lib.lib().ecne_r1cs_info(self._h, C.byref(i)))
        return i

    def __len__(self):
        return int(self.info.n_constraints)

    def io(self):
        kn, tg = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)()
        nk, nt = C.c_size_t(), C.c_size_t()
        _check(_lib.lib().ecne_r1cs_io(self._h, C.byref(kn), C.byref(nk), C.byref(tg), C.byref(nt)))
        return [kn[i] for i in range(nk.value)], [tg[i] for i in range(nt.value)]

    def csr(self, part):
        """(rowptr, col, coeff[nnz,4]) of part 0/1/2 in file order; copies."""
        rp, col, cf = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint64)()
        _check(_lib.lib().ecne_r1cs_csr(self._h, part, C.byref(rp), C.byref(col), C.byref(cf)))
        n = len(self)
        rowptr = np.ctypeslib.as_array(rp, (n + 1,)).copy()
        nnz = int(rowptr[-1])
        if nnz == 0:      # a part without a single non-zero term: the library may hand out null arrays
            return rowptr, np.zeros(0, np.uint32), np.zeros((0, 4), np.uint64)
        c = np.ctypeslib.as_array(col, (nnz,)).copy()
        v = np.ctypeslib.as_array(cf, (nnz * 4,)).reshape(nnz, 4).copy()
        return rowptr, c, v


class System:
    """Rows + special constraints + I/O lists: what SolveConstraintsSymbolic is called with."""

    def __init__(self, r1cs):
        self._h = C.c_void_p()
        self.main = r1cs
        _check(_lib.lib().ecne_system_from_r1cs(r1cs._h, C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:
            _lib.lib().ecne_system_free(h)
            self._h = None

    def abstract(self, trusted, name):
        """abstraction(name, rows, knowns_t, eqs_t, outs_t) (:237-395) applied in place."""
        _check(_lib.lib().ecne_abstract(self._h, trusted._h, name.encode()), "abstraction(%s)" % name)

    @staticmethod
    def last_abstract_stats():
        """of the calling thread's last abstract(): dict(device, fingerprint_ms, scan_ms, bytes, upload_ms, candidates)"""
        a = (C.c_double * 6)()
        _check(_lib.lib().ecne_abstract_stats(a))
        return dict(device=bool(a[0]), fingerprint_ms=a[1], scan_ms=a[2], bytes=a[3], upload_ms=a[4], candidates=int(a[5]))

    @property
    def info(self):
        i = SystemInfo()
        _check(_lib.lib().ecne_system_info_get(self._h, C.byref(i)))
        return i

    def split_info(self):
        """(parts the next solve runs as -- 0: as one system --, groups of rows found, plan ms, a plan has been looked for): ecne_set_split"""
        a = (C.c_double * 4)()
        _check(_lib.lib().ecne_system_split_info(self._h, a))
        return int(a[0]), int(a[1]), float(a[2]), bool(a[3])

    def io(self):
        """(known_variables, target_variables) the solve will be asked with"""
        kn, tg = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)()
        nk, nt = C.c_size_t(), C.c_size_t()
        _check(_lib.lib().ecne_system_io(self._h, C.byref(kn), C.byref(nk), C.byref(tg), C.byref(nt)))
        return [kn[i] for i in range(nk.value)], [tg[i] for i in range(nt.value)]

    def set_io(self, known_variables, target_variables):
        kn = (C.c_int64 * max(len(known_variables), 1))(*known_variables)
        tg = (C.c_int64 * max(len(target_variables), 1))(*target_variables)
        _check(_lib.lib().ecne_system_set_io(self._h, kn, len(known_variables), tg, len(target_variables)), "set_io")

    def set_secp_solve(self, flag):
        """this system's kwarg secp_solve inside a batch launch (True / False; None = what solve_batch is called with)"""
        _check(_lib.lib().ecne_system_set_secp_solve(self._h, -1 if flag is None else int(bool(flag))))

    def set_specials(self, special_constraints):
        """special_constraints: [(name, inputs, outputs), ...] as abstraction builds them (:388)"""
        L = _lib.lib()
        _check(L.ecne_system_clear_specials(self._h))
        for name, ins, outs in special_constraints:
            a = (C.c_int64 * max(len(ins), 1))(*ins)
            b = (C.c_int64 * max(len(outs), 1))(*outs)
            _check(L.ecne_system_add_special(self._h, name.encode(), a, len(ins), b, len(outs)), "add_special")

    def specials(self):
        out = []
        for k in range(int(self.info.n_specials)):
            name = C.c_char_p()
            ins, outs = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)()
            ni, no = C.c_size_t(), C.c_size_t()
            _check(_lib.lib().ecne_system_special(self._h, k, C.byref(name), C.byref(ins), C.byref(ni),
                                                  C.byref(outs), C.byref(no)))
            out.append((name.value.decode(), [ins[i] for i in range(ni.value)], [outs[i] for i in range(no.value)]))
        return out

    def __len__(self):
        return int(self.info.n_rows)

    def dict_rows(self, part):
        """(rowptr, var, coeff[n,4]) of part 0/1/2 in the reference's DICTIONARY order (zeros and placeholders included); copies."""
        rp, var, cf, n = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint64)(), C.c_uint64()
        _check(_lib.lib().ecne_system_dict_rows(self._h, part, C.byref(rp), C.byref(var), C.byref(cf), C.byref(n)))
        rowptr = np.ctypeslib.as_array(rp, (n.value + 1,)).copy()
        m = int(rowptr[-1])
        if m == 0:
            return rowptr, np.zeros(0, np.uint32), np.zeros((0, 4), np.uint64)
        return rowptr, np.ctypeslib.as_array(var, (m,)).copy(), np.ctypeslib.as_array(cf, (m * 4,)).reshape(m, 4).copy()

    def static_array(self, which, device=0):
        """test hook: bytes of static array `which` of the device image (ecne_debug_static_array)"""
        p, n = C.c_void_p(), C.c_size_t()
        _check(_lib.lib().ecne_debug_static_array(self._h, device, which, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value) if n.value else b""

    def rows(self, part):
        """(rowptr, col, coeff[nnz,4]) of part 0/1/2 in the reference's nonzeroKeys order; copies."""
        rp, col, cf = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint64)()
        _check(_lib.lib().ecne_system_rows(self._h, part, C.byref(rp), C.byref(col), C.byref(cf)))
        n = len(self)
        rowptr = np.ctypeslib.as_array(rp, (n + 1,)).copy()
        nnz = int(rowptr[-1])
        if nnz == 0:      # a part without a single non-zero term: the library may hand out null arrays
            return rowptr, np.zeros(0, np.uint32), np.zeros((0, 4), np.uint64)
        c = np.ctypeslib.as_array(col, (nnz,)).copy()
        v = np.ctypeslib.as_array(cf, (nnz * 4,)).reshape(nnz, 4).copy()
        return rowptr, c, v


class SolveResult:
    """Outcome of one solve. With fetch_states=False only the summary (verdict + counts) is read
    back; the per-variable arrays stay in HBM and the attributes below are None."""

    flags = lb = ub = abz = nvalues = values = bad_rows = None

    def __init__(self, handle, fetch_states=True, summary=None):
        L = _lib.lib()
        if summary is not None and not fetch_states:      # (solve_batch: the summaries of the whole batch came in one call, the handles are freed in one)
            self.summary, self.status, self.function_good, self.digest = summary, int(summary.status), bool(summary.function_good), None
            return
        try:
            self._read(L, handle, fetch_states)
        finally:
            L.ecne_result_free(handle)      # whatever happened above: the handle keeps its result object alive

    def _read(self, L, handle, fetch_states):
        s = Summary()
        _check(L.ecne_result_summary(handle, C.byref(s)))
        self.summary = s
        self.status = int(s.status)
        self.function_good = bool(s.function_good)
        self.digest = None
        if fetch_states in ("digest", "both"):      # the device-side digest of the per-variable state (instead of / next to the state itself)
            d = (C.c_uint64 * 2)()
            _check(L.ecne_result_digest(handle, d))
            self.digest = (int(d[0]), int(d[1]))
            if fetch_states == "digest":
                return
        if not fetch_states:
            return
        nv = int(s.n_vars)
        fl, lb, ub = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        abz, nvs, vals = C.POINTER(C.c_int32)(), C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint64)()
        _check(L.ecne_result_states(handle, C.byref(fl), C.byref(lb), C.byref(ub), C.byref(abz), C.byref(nvs),
                                    C.byref(vals)))

        def arr(p, shape):
            if nv == 0:
                return np.zeros(shape, dtype=np.ctypeslib.as_array(p, (1,)).dtype if p else np.uint8)
            return np.ctypeslib.as_array(p, shape).copy()
        self.flags = arr(fl, (nv,))
        self.lb = arr(lb, (nv, 4))
        self.ub = arr(ub, (nv, 4))
        self.abz = arr(abz, (nv,))
        self.nvalues = arr(nvs, (nv,))
        self.values = arr(vals, (nv, 2, 4))
        rows, n = C.POINTER(C.c_int64)(), C.c_size_t()
        _check(L.ecne_result_bad_rows(handle, C.byref(rows), C.byref(n)))
        self.bad_rows = np.array([rows[i] for i in range(n.value)], dtype=np.int64)

    @property
    def unique(self):
        return (self.flags & 1).astype(bool)

    @property
    def is_known(self):
        return ((self.flags >> 1) & 1).astype(bool)

    def counts(self):
        s = self.summary
        return (s.unique_nontrivial, s.n_nontrivial, s.unique_targets, s.n_targets)

    def raise_for_status(self):
        _check(self.status, "SolveConstraintsSymbolic")


def _opts(device=0, secp_solve=False, queue_mode=0, stream=None, force_nwg=0):
    o = Opts()
    o.device = int(device)
    o.secp_solve = int(bool(secp_solve))
    o.debug = int(force_nwg)
    o.queue_mode = int(queue_mode)
    o.stream = stream
    return o


def solve_batch(systems, secp_solve=False, device=0, queue_mode=0, stream=None, fetch_states=True, force_nwg=0):
    """Run n independent systems in one launch (one workgroup each). Returns SolveResult list."""
    L = _lib.lib()
    n = len(systems)
    hs = (C.c_void_p * n)(*[s._h for s in systems])
    outs = (C.c_void_p * n)()
    o = _opts(device, secp_solve, queue_mode, stream, force_nwg)
    _check(L.ecne_solve_batch(hs, n, C.byref(o), outs), "ecne_solve_batch")
    if not fetch_states and n > 1:
        # only the summaries: one call for the whole batch, one to free the handles (504 small systems: two FFI calls per result were
        # more time than the GPU spent on the batch)
        sums = (Summary * n)()
        try:
            _check(L.ecne_result_summaries(outs, n, sums), "ecne_result_summaries")
        finally:
            L.ecne_results_free(outs, n)
        return [SolveResult(None, False, sums[i]) for i in range(n)]
    results, err = [], None
    for i in range(n):      # (every handle is read or at least freed: an exception on one result must not leak the others)
        try:
            results.append(SolveResult(C.c_void_p(outs[i]), fetch_states))
        except Exception as e:      # noqa: BLE001
            err = err or e
    if err is not None:
        raise err
    return results


def classify(system, device=0):
    """k_classify_rows on one system: (shape words, kernel ms, bytes streamed)."""
    L = _lib.lib()
    n = len(system)
    shape = np.zeros(max(n, 1), np.uint32)
    ms, by = C.c_double(), C.c_uint64()
    o = _opts(device)
    _check(L.ecne_classify(system._h, C.byref(o), shape.ctypes.data, C.byref(ms), C.byref(by)), "ecne_classify")
    return shape[:n], ms.value, int(by.value)


# ------------------------------------------------------------------ the reference's interface
def readR1CS(filename):
    """ParseR1CS.jl:50-124 -> (equations, known_vars, output_vars, nVars)."""
    f = R1CS(filename)
    kn, out = f.io()
    return f, kn, out, int(f.info.n_vars)


# state of the last solve, for callers that want more than the Bool
last_result = None


def SolveConstraintsSymbolic(constraints, special_constraints=None, known_variables=None, debug=False,
                             target_variables=None, num_variables=-1, input_sym="default.sym",
                             secp_solve=False, device=0):
    """R1CSConstraintSolver.jl:583-1646.  `constraints` is a System (the rows after abstraction) or an R1CS.
    `special_constraints`, `known_variables` and `target_variables` are the reference's arguments: when given they
    REPLACE what the handle carries (the file's lists, the specials abstraction produced) before the solve, as a caller
    who edits them expects; None keeps the handle's. `num_variables` has to be the handle's nVars (or -1): the state
    arrays are sized by it (:681) and it cannot be changed after parsing."""
    global last_result
    t_begin = time.time()
    system = constraints if isinstance(constraints, System) else System(constraints)
    if num_variables not in (-1, None) and int(num_variables) != int(system.info.n_vars):
        raise ValueError("num_variables=%d differs from the system's %d" % (num_variables, system.info.n_vars))
    kn0, tg0 = system.io()
    if known_variables is not None or target_variables is not None:
        kn = list(known_variables) if known_variables is not None else kn0
        tg = list(target_variables) if target_variables is not None else tg0
        if kn != kn0 or tg != tg0:
            system.set_io(kn, tg)
    if special_constraints is not None:
        sp = [(str(c[0]), [int(x) for x in c[1]], [int(x) for x in c[2]]) for c in special_constraints]
        if sp != system.specials():
            system.set_specials(sp)
    system.info                                              # the flat arrays (the reference's per-solve set-up, :593-703)
    print("setup solver %d milliseconds" % int((time.time() - t_begin) * 1000))   # :704 (always printed)
    res = solve_batch([system], secp_solve=secp_solve, device=device)[0]
    last_result = res
    res.raise_for_status()
    s = res.summary
    from . import report
    # what the reference always prints (:1565-1571, :1586-1592), the state dump of debug=true between the two (:1573-1577),
    # and its report (:1599-1643)
    print("Solved for %d variables out of %d total variables" % (s.unique_nontrivial, s.n_nontrivial))
    if debug:
        print(report.debug_states(system, res), end="")
    print("Solved for %d target variables out of %d total target variables" % (s.unique_targets, s.n_targets))
    sym = report.read_sym(input_sym) if input_sym and os.path.exists(input_sym) else None
    if input_sym and sym is None:
        raise FileNotFoundError(input_sym)                  # CSV.File(input_sym) (:1603) on a missing file -- "default.sym" included
    print(report.render(system, res, sym), end="")
    return res.function_good


def solveWithTrustedFunctions(input_r1cs, input_r1cs_name, trusted_r1cs=(), trusted_r1cs_names=(),
                              debug=False, printRes=True, abstractionOnly=False, input_sym="",
                              secp_solve=False, device=0):
    """R1CSConstraintSolver.jl:502-581."""
    a = time.time()
    assert len(trusted_r1cs) == len(trusted_r1cs_names)                      # :514
    main, _knowns, _outs, _nv = readR1CS(input_r1cs)                           # :515
    function_list = []
    for path, name in zip(trusted_r1cs, trusted_r1cs_names):                   # :517-523
        function_list.append((name, R1CS(path)))
    if debug:
        print("file read")
    function_list.sort(key=lambda x: -len(x[1]))                               # :527 (stable)
    system = System(main)
    for name, f in function_list:                                              # :531-544
        if printRes:
            print("called abstraction")
        system.abstract(f, name)
    if abstractionOnly:                                                        # :546-549: println(specials)
        print("Any[" + ", ".join('("%s", %s, %s)' % (n, list(i), list(o)) for n, i, o in system.specials()) + "]")
        return True
    print("time to prep inputs %d milliseconds" % int((time.time() - a) * 1000))   # :551
    result = SolveConstraintsSymbolic(system, None, None, debug, None, -1, input_sym if input_sym else "", secp_solve, device=device)
    if result:
        if function_list:
            if printRes:
                # :556-559 is a tuple assignment that reads `msg` before binding it: UndefVarError
                raise UndefVarError("msg not defined (reference :556-559)")
            return True
        if printRes:
            print("R1CS function " + input_r1cs_name + " has sound constraints (No trusted functions needed!)")
        return True
    if printRes:
        print("R1CS function " + input_r1cs_name + " has potentially unsound constraints")
    return False
