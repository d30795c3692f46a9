"""Text report of a solve, as the reference prints it when `input_sym` is given
(/root/reference/src/R1CSConstraintSolver.jl:1599-1645: "Bad Constraints" with printEquation
:431-456 and printState :397-419).  Host-side rendering only: the row numbers
come from ecne_result_bad_rows, the per-variable state from ecne_result_states, the term order
from ecne_system_rows (the reference's nonzeroKeys order)."""
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_FIX = 21888242871839275222246405745257275088548363400416034343698204186575808495517   # fix_number :421-428


def _int(limbs):
    return sum(int(limbs[i]) << (64 * i) for i in range(4))


def fix_number(x):
    return x - P if x > _FIX else x


def read_sym(path):
    """4-column CSV, 4th column = signal name, line k <-> variable k + 1 (:1603-1607)."""
    names = []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line:
                names.append(line.split(",", 3)[3])
    return names


def equation_text(system, row, names, _cache={}):
    """printEquation (:431-456) for 1-based row `row`."""
    key = id(system)
    if key not in _cache:
        _cache.clear()
        _cache[key] = [system.rows(p) for p in range(3)]
    parts = []
    for rp, col, cf in _cache[key]:
        a, b = int(rp[row - 1]), int(rp[row])
        if a == b:
            parts.append("0")
            continue
        terms = []
        for k in range(a, b):
            v = int(col[k])
            name = names[v - 2] if v > 1 else "1"          # fix_signal(key - 1)
            terms.append("%d * %s" % (fix_number(_int(cf[k])), name))
        parts.append("(" + " + ".join(terms) + ")")
    return "%s * %s = %s" % tuple(parts)


def state_text(result, v):
    """printState (:397-419) for 1-based variable v."""
    lb, ub = _int(result.lb[v - 1]), _int(result.ub[v - 1])
    out = ["Uniquely Determined: " + ("true" if result.flags[v - 1] & 1 else "false")]
    out.append("Bounds: None" if (lb == 0 and ub == P - 1) else "Bounds: [%d, %d]" % (lb, ub))
    n = int(result.nvalues[v - 1])
    if n:
        # println of a Vector{BigInt} carries the element type: "BigInt[0, 1]"
        out.append("All possible values: BigInt" + str(sorted(_int(result.values[v - 1][i]) for i in range(n))))
    out.append("")
    return out


def _order(system, row):
    """variables in the order the reference's report walks them (ecne_system_report_order)"""
    import ctypes as C
    from . import _lib
    ptr, n = C.POINTER(C.c_int64)(), C.c_size_t()
    st = _lib.lib().ecne_system_report_order(system._h, int(row), C.byref(ptr), C.byref(n))
    if st != 0:
        raise ValueError("ecne_system_report_order: status %d" % st)
    return [int(ptr[i]) for i in range(n.value)]


def bad_constraints_report(system, result, sym_path):
    """The lines between "------ Bad Constraints ------" and "------ All Variables ------" (:1609-1632)."""
    names = read_sym(sym_path) if isinstance(sym_path, str) else sym_path
    lines = []
    for row in result.bad_rows.tolist():
        lines.append("constraint #%d" % row)
        lines.append(equation_text(system, row, names))
        for v in _order(system, row):                # for j in getVariables(constraints[i]) (:1625)
            if v == 1:
                continue
            lines.append(names[v - 2])
            lines.extend(state_text(result, v))
    return lines


def all_variables_report(system, result, names):
    """The lines after "------ All Variables ------" (:1633-1643): every non-trivial variable but the constant wire, in the
    iteration order of the reference's Set."""
    lines = []
    for v in _order(system, 0):
        if v == 1:
            continue
        lines.append(names[v - 2])
        lines.extend(state_text(result, v))
    return lines


def debug_states(system, result):
    """debug=true: printState of every non-trivial variable, in the iteration order of the reference's Set, between the two
    "Solved for" lines (:1573-1577; the constant wire included, no signal names)."""
    lines = []
    for v in _order(system, 0):
        lines.extend(state_text(result, v))
    return "".join(l + "\n" for l in lines)


def render(system, result, names=None):
    """Everything SolveConstraintsSymbolic prints after its two count lines (:1599-1643). `names` = read_sym(input_sym), or
    None when input_sym == "" (then only the header is printed, as in the reference)."""
    out = ["------ Bad Constraints ------", ""]
    if names is not None:
        out += bad_constraints_report(system, result, names)
        out += ["------ All Variables ------", ""]
        out += all_variables_report(system, result, names)
    return "\n".join(out) + "\n"
