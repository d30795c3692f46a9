"""ref2 -- an independent SECOND reading of the reference's solver path, in plain sequential Python.

Test infrastructure (N-version check of oracle/ecne_oracle.cpp): written from the text of
/root/reference/src/R1CSConstraintSolver.jl (:26-56, :135-201, :205-395, :502-581, :583-1597) and
/root/reference/src/ParseR1CS.jl (:50-124) WITHOUT consulting the C++ oracle, statement for statement, with Python integers
for the field and with the reference's own data structures modelled as objects:

  JDict   Julia 1.7 `Dict` (base/dict.jl: open addressing, linear probing, 16 slots to start with, growth x4 / x2 when more than
          2/3 full or when a probe sequence reaches max(16, sz >> 6), rehash in slot order, `sizehint!`, iteration in slot order)
          -- validated against the 16 620 known-answer vectors of the reference's own dumps (tests/test_ref2.py)
  JSet    `Set{Any}` = a JDict of keys; `Set(itr)` goes through union! and therefore sizehint!
  DefaultDict with insert-on-read (`d[k]` on a missing key stores the default: DataStructures' get!)
  R1CSEquation with three DefaultDicts that really ARE rebuilt when checkBinary flips a row (:1003-1009): the flipped dictionary is
          filled in the old one's iteration order, which is not always the old one's slot layout
  VariableState objects with reference semantics (state_1 = variable_states[key_1] aliases, :1095) and make_values /
          make_bounds returning NEW objects whose abz is reset to -1 (:148-159)

tests/test_ref2.py compares verdict, printed counts, successful_steps / num_unique and the whole per-variable state with the
oracle. Any disagreement is a finding about the reference text.
"""
import itertools
import struct

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
M64 = (1 << 64) - 1


class BoundsError(Exception):
    status = -2


class DivideError(Exception):
    status = -3


class UndefVarError(Exception):
    status = -4


class JlKeyError(Exception):
    status = -5


class Watchdog(Exception):
    status = -12


# ------------------------------------------------------------------------------------------------ Julia 1.7 Dict / Set
def hash_64_64(n):
    a = n & M64
    a = (~a + (a << 21)) & M64
    a = a ^ (a >> 24)
    a = (a + (a << 3) + (a << 8)) & M64
    a = a ^ (a >> 14)
    a = (a + (a << 2) + (a << 4)) & M64
    a = a ^ (a >> 28)
    a = (a + (a << 31)) & M64
    return a


def _tablesz(x):
    n = 16
    while n < x:
        n <<= 1
    return n


class JDict:
    """base/dict.jl of Julia 1.7 for Int64 keys (hash(x::Int64) = hash_64_64(x)); no deletions happen on this path"""
    __slots__ = ("slots", "ks", "vs", "count", "maxprobe")

    def __init__(self):
        self.slots = [0] * 16
        self.ks = [None] * 16
        self.vs = [None] * 16
        self.count = 0
        self.maxprobe = 0

    def __len__(self):
        return self.count

    def _keyindex(self, key):                       # ht_keyindex: -1 when absent
        sz = len(self.ks)
        index = hash_64_64(key) & (sz - 1)
        it = 0
        while True:
            if not self.slots[index]:
                return -1
            if self.ks[index] == key:
                return index
            index = (index + 1) & (sz - 1)
            it += 1
            if it > self.maxprobe:
                return -1

    def _keyindex2(self, key):                      # ht_keyindex2!: index >= 0 found, else -(slot + 1) to insert at
        while True:
            sz = len(self.ks)
            index = hash_64_64(key) & (sz - 1)
            it = 0
            done = False
            while True:
                if not self.slots[index]:
                    return -(index + 1)
                if self.ks[index] == key:
                    return index
                index = (index + 1) & (sz - 1)
                it += 1
                if it > self.maxprobe:
                    break
            maxallowed = max(16, sz >> 6)
            while it < maxallowed:
                if not self.slots[index]:
                    self.maxprobe = it
                    return -(index + 1)
                index = (index + 1) & (sz - 1)
                it += 1
            self._rehash(sz * 2 if self.count > 64000 else sz * 4)
            if done:
                break

    def _rehash(self, newsz):
        newsz = _tablesz(newsz)
        olds, oldk, oldv = self.slots, self.ks, self.vs
        slots, ks, vs = [0] * newsz, [None] * newsz, [None] * newsz
        maxprobe = 0
        for i in range(len(oldk)):
            if olds[i]:
                k = oldk[i]
                index0 = index = hash_64_64(k) & (newsz - 1)
                while slots[index]:
                    index = (index + 1) & (newsz - 1)
                probe = (index - index0) & (newsz - 1)
                if probe > maxprobe:
                    maxprobe = probe
                slots[index] = 1
                ks[index] = k
                vs[index] = oldv[i]
        self.slots, self.ks, self.vs, self.maxprobe = slots, ks, vs, maxprobe

    def _setindex(self, v, key, index):             # _setindex!
        self.slots[index] = 1
        self.ks[index] = key
        self.vs[index] = v
        self.count += 1
        sz = len(self.ks)
        if self.count * 3 > sz * 2:
            self._rehash(self.count * 2 if self.count > 64000 else self.count * 4)

    def __setitem__(self, key, v):
        index = self._keyindex2(key)
        if index >= 0:
            self.ks[index] = key
            self.vs[index] = v
        else:
            self._setindex(v, key, -index - 1)

    def __contains__(self, key):
        return self._keyindex(key) >= 0

    def get(self, key, default=None):
        i = self._keyindex(key)
        return default if i < 0 else self.vs[i]

    def get_or_insert(self, key, make_default):     # get!(h, key, default)
        index = self._keyindex2(key)
        if index >= 0:
            return self.vs[index]
        v = make_default()
        self._setindex(v, key, -index - 1)
        return v

    def sizehint(self, newsz):                      # sizehint! (never shrinks)
        oldsz = len(self.slots)
        newsz = -(-3 * newsz // 2)
        if newsz <= oldsz:
            return
        self._rehash(newsz)

    def items(self):
        return [(self.ks[i], self.vs[i]) for i in range(len(self.ks)) if self.slots[i]]

    def keys(self):
        return [self.ks[i] for i in range(len(self.ks)) if self.slots[i]]

    def values(self):
        return [self.vs[i] for i in range(len(self.ks)) if self.slots[i]]


class JSet:
    """Set{Any}: Set() then push!, or Set(itr) = union!(Set(), itr) which reserves room first (abstractset.jl)"""
    __slots__ = ("d",)

    def __init__(self, itr=None):
        self.d = JDict()
        if itr is not None:
            itr = list(itr)
            self.d.sizehint(len(self.d) + len(itr))
            for x in itr:
                self.push(x)

    def push(self, x):
        self.d[x] = None

    def __contains__(self, x):
        return x in self.d

    def __iter__(self):
        return iter(self.d.keys())

    def __len__(self):
        return len(self.d)


class FDict(JDict):
    """DefaultDict{Int64, GFElem}(F(0)): reading a missing key stores F(0) (insert-on-read)"""
    __slots__ = ()

    def __getitem__(self, key):
        return self.get_or_insert(key, lambda: 0)


class Equation:
    __slots__ = ("a", "b", "c")

    def __init__(self, a, b, c):
        self.a, self.b, self.c = a, b, c


def dict_eq(l, r):
    """== of two AbstractDicts: same length and every pair of l is in r"""
    if len(l) != len(r):
        return False
    for k, v in l.items():
        i = r._keyindex(k)
        if i < 0 or r.vs[i] != v:
            return False
    return True


# ------------------------------------------------------------------------------------------------ field helpers
def finv(x):
    return pow(x, P - 2, P)


def divexact(a, b):
    if b % P == 0:
        raise DivideError()
    return a * finv(b) % P


def neg(a):
    return (-a) % P


# ------------------------------------------------------------------------------------------------ readR1CS (ParseR1CS.jl:50-124)
class FormatError(Exception):
    status = -1


def read_r1cs(path):
    with open(path, "rb") as f:
        arr = f.read()

    def four(i):                                     # 1-based index like the reference's arr[cur_idx:cur_idx+3]
        if i - 1 + 4 > len(arr):
            raise FormatError()
        return struct.unpack_from("<I", arr, i - 1)[0]

    def arrint(i, n):
        if i - 1 + n > len(arr):
            raise FormatError()
        return int.from_bytes(arr[i - 1:i - 1 + n], "little")
    cur = 5
    if four(cur) != 1:
        raise FormatError()
    cur = 9
    sections = four(cur)
    cur += 4
    if sections != 3:
        raise FormatError()
    starts = [0, 0, 0]
    for _ in range(sections):
        s = four(cur)
        if not 1 <= s <= 3:
            raise FormatError()
        starts[s - 1] = cur
        cur += 4
        cur += arrint(cur, 8) + 8
    sec1 = starts[0] + 12
    fs = four(sec1)
    sec1 += 4 + fs
    nwires = four(sec1)
    pub_out = four(sec1 + 4)
    pub_in = four(sec1 + 8)
    prv_in = four(sec1 + 12)
    sec1 += 16
    arrint(sec1, 8)
    sec1 += 8
    ncons = four(sec1)
    sec2 = starts[1] + 12
    eqs = []
    for _ in range(ncons):
        parts = []
        for _p in range(3):
            n = four(sec2)
            sec2 += 4
            d = FDict()
            for _k in range(n):
                idx = four(sec2)
                sec2 += 4
                coeff = arrint(sec2, 32)
                sec2 += 32
                d[idx + 1] = coeff % P
            if n == 0:
                d[1] = 0
            parts.append(d)
        eqs.append(Equation(*parts))
    knowns = [1] + list(range(2 + pub_out, 1 + pub_out + pub_in + prv_in + 1))
    outs = list(range(2, 1 + pub_out + 1))
    return eqs, knowns, outs, nwires + 1


# ------------------------------------------------------------------------------------------------ nonzeroKeys / getVariables (:26-56)
def nonzero_keys(lin):
    s = JSet()
    for k, v in lin.items():
        if v != 0:
            s.push(k)
    return s


def get_variables(eq):
    s = JSet()
    for part in (eq.a, eq.b, eq.c):
        for k, v in part.items():
            if v != 0:
                s.push(k)
    return s


# ------------------------------------------------------------------------------------------------ abstraction (:205-395)
def _check_nonzero_values(m1, m2):
    from collections import Counter
    x1, x2 = Counter(m1.values()), Counter(m2.values())
    for e in x1:
        if e != 0 and x1[e] != x2[e]:
            return False
    for e in x2:
        if e != 0 and x1[e] != x2[e]:
            return False
    return True


def _hash_list(e):
    l = sorted(e.a.values()) + sorted(e.b.values()) + sorted(e.c.values())
    return tuple(x for x in l if x != 0)            # (the reference hashes this list; equal lists <=> equal hashes up to collisions)


def abstraction(name, constraints, known_inputs, sub, known_outputs):
    hc = [_hash_list(x) for x in constraints]
    hs = [_hash_list(x) for x in sub]
    cands = []
    for i in range(1, len(constraints) - len(sub) + 2):
        ok = True
        for j in range(1, len(sub)):
            if hc[i + j - 2] != hs[j - 1]:
                ok = False
                break
        if ok:
            cands.append(i)
    matches = []
    amo = JDict()                                    # DefaultDict{Int64, Vector{Tuple}}(Vector)
    counter = 1
    for j in range(len(sub)):
        for eq in (sub[j].a, sub[j].b, sub[j].c):
            for k, v in eq.items():
                if v != 0:
                    amo.get_or_insert(k, list).append((counter, v))
            counter += 1
    for i in cands:
        works = True
        amc = JDict()
        app = 0

        def add_equation(e1, e2, app):
            if not _check_nonzero_values(e1, e2):
                return False
            for k, v in e1.items():
                if v != 0:
                    amc.get_or_insert(k, list).append((app, v))
            return True
        for j in range(1, len(sub) + 1):
            for part in ("a", "b", "c"):
                app += 1
                if not add_equation(getattr(constraints[i + j - 2], part), getattr(sub[j - 1], part), app):
                    works = False
                    break
            if not works:
                break
        if not works:
            continue
        l1 = sorted(amc.items(), key=lambda x: x[1])        # stable, by the list of (counter, value) tuples
        l2 = sorted(amo.items(), key=lambda x: x[1])
        if len(l1) != len(l2):
            continue
        if any(l1[x][1] != l2[x][1] for x in range(len(l1))):
            continue
        matches.append((i, {l2[x][0]: l1[x][0] for x in range(len(l1))}))
    red, specials = [], []
    cur = 1
    i = 1
    while i <= len(constraints):
        if cur > len(matches) or i != matches[cur - 1][0]:
            red.append(constraints[i - 1])
            i += 1
        else:
            m = matches[cur - 1][1]
            try:
                specials.append((name, [m[x] for x in known_inputs if x != 1], [m[x] for x in known_outputs]))
            except KeyError:
                raise JlKeyError()
            i += len(sub)
            cur += 1
    return specials, red


# ------------------------------------------------------------------------------------------------ VariableState (:135-201)
class VS:
    __slots__ = ("index", "is_known", "unique", "values", "lb", "ub", "abz")

    def __init__(self, index, is_known=False, unique=False, values=None, lb=0, ub=P - 1, abz=-1):
        self.index, self.is_known, self.unique = index, is_known, unique
        self.values = [] if values is None else values
        self.lb, self.ub = lb, ub
        self.abz = -1                                # the 8-argument constructor ignores its abz argument (:148-159)


def make_values(a, new_values):
    return VS(a.index, True, a.unique, new_values, a.lb, a.ub, a.abz)


def make_bounds(a, lb, ub):
    return VS(a.index, True, a.unique, a.values, lb, ub, a.abz)


class IntDisjointSet:
    def __init__(self, n):
        self.parents = list(range(1, n + 1))
        self.ranks = [0] * n

    def push(self):
        self.parents.append(len(self.parents) + 1)
        self.ranks.append(0)
        return len(self.parents)

    def find_root(self, x):
        if not 1 <= x <= len(self.parents):
            raise BoundsError()
        p = self.parents[x - 1]
        if self.parents[p - 1] != p:
            p = self.find_root(p)
            self.parents[x - 1] = p
        return p

    def union(self, x, y):
        px, py = self.find_root(x), self.find_root(y)
        if px == py:
            return
        rx, ry = self.ranks[px - 1], self.ranks[py - 1]
        if rx < ry:
            px, py = py, px
        elif rx == ry:
            self.ranks[px - 1] += 1
        self.parents[py - 1] = px

    def in_same_set(self, x, y):
        return self.find_root(x) == self.find_root(y)


class Result:
    pass


_FLIP_THRESHOLD = 20888242871839275222246405745257275088548364400416034343698204186575808495616   # the literal of :1247 (NOT p - 1)


def parity(perm):
    """Combinatorics.parity: 0 for an even, 1 for an odd permutation"""
    inv = 0
    for i in range(len(perm)):
        for j in range(i + 1, len(perm)):
            inv += perm[i] > perm[j]
    return inv & 1


# ------------------------------------------------------------------------------------------------ SolveConstraintsSymbolic (:583-1597)
def solve(constraints, special_constraints, known_variables, target_variables, num_variables, secp_solve=False, pop_cap=None):
    R = Result()
    nC = len(constraints)

    def vs_get(i):
        if not 1 <= i <= num_variables:
            raise BoundsError()
        return variable_states[i - 1]

    def vs_set(i, v):
        if not 1 <= i <= num_variables:
            raise BoundsError()
        variable_states[i - 1] = v
    known_set = JSet(known_variables)
    num_unknowns = [sum(1 for v in get_variables(x) if v not in known_set) for x in constraints]
    in_queue = [False] * nC
    equation_solved = [False] * nC
    special_solved = [False] * len(special_constraints)
    l = []
    for eq in constraints:
        for j in get_variables(eq):
            l.append(j)
    for sp in special_constraints:
        l.extend(sp[1])
        l.extend(sp[2])
    l.extend(target_variables)
    all_nontrivial = JSet(l)
    from collections import deque
    q = deque()
    for i in range(1, nC + 1):
        if num_unknowns[i - 1] <= 1:
            q.append(i)
            in_queue[i - 1] = True
    v2i = {}
    for i in range(1, nC + 1):
        for j in get_variables(constraints[i - 1]):
            v2i.setdefault(j, []).append(i)
    dsu = None
    if secp_solve:
        dsu = IntDisjointSet(num_variables)
        const_vals = {}
        for eq in constraints:
            if len(nonzero_keys(eq.a)) == 0 and len(nonzero_keys(eq.b)) == 0:
                if len(eq.c) == 2:
                    if sorted(eq.c.values()) == sorted([1, P - 1]):
                        ll = list(nonzero_keys(eq.c))
                        dsu.union(ll[0], ll[1])
                    else:
                        ll = []
                        constant_val = False
                        for i in nonzero_keys(eq.c):
                            ll.append(i)
                            if i == 1:
                                constant_val = True
                        if len(ll) < 1:
                            raise BoundsError()
                        non_one = ll[0]
                        if ll[0] == 1:
                            if len(ll) < 2:
                                raise BoundsError()
                            non_one = ll[1]
                        if not constant_val:
                            continue
                        value = divexact(eq.c[1], neg(eq.c[non_one]))
                        if value not in const_vals:
                            const_vals[value] = dsu.push()
                        dsu.union(non_one, const_vals[value])
    variable_states = [VS(i) for i in range(1, num_variables + 1)]
    for i in known_variables:
        st = vs_get(i)
        if i == 1:
            st.values = [1]
        st.unique = True
        st.is_known = True
    successful_steps = 0
    prev_successful_steps = -1
    nzk_a = [nonzero_keys(c.a) for c in constraints]
    nzk_b = [nonzero_keys(c.b) for c in constraints]
    nzk_c = [nonzero_keys(c.c) for c in constraints]
    num_unique = 0
    pops = 0
    outer = 0
    if pop_cap is None:
        pop_cap = 4096 + 64 * sum(len(a) + len(b) + len(c) for a, b, c in zip(nzk_a, nzk_b, nzk_c))

    def requeue(var):
        for r in v2i.get(var, ()):
            if not in_queue[r - 1]:
                q.append(r)
                in_queue[r - 1] = True

    while True:
        if prev_successful_steps == successful_steps:
            break
        prev_successful_steps = successful_steps
        outer += 1
        # ---- P1 (:718-747)
        for i in range(len(special_constraints)):
            if not special_solved[i]:
                solved = True
                for j in special_constraints[i][1]:
                    if not vs_get(j).unique:
                        solved = False
                        break
                if not solved:
                    continue
                special_solved[i] = True
                successful_steps += 1
                for j in special_constraints[i][2]:
                    st = vs_get(j)
                    if st.unique:
                        continue
                    st.unique = True
                    st.is_known = True
                    requeue(j)
        # ---- P2 (:750-800)
        for i in range(len(special_constraints)):
            if special_constraints[i][0] != "BigMultModP":
                continue
            for j in range(len(special_constraints)):
                if special_constraints[j][0] != "BigLessThan":
                    continue
                ci, cj = special_constraints[i], special_constraints[j]
                same_set = True
                for k in range(1, 7):
                    if dsu is None:
                        raise UndefVarError()
                    if k + 3 > len(ci[1]) or k > len(cj[1]):
                        raise BoundsError()
                    if not dsu.in_same_set(ci[1][k + 2], cj[1][k - 1]):
                        same_set = False
                if same_set:
                    if len(cj[2]) < 1:
                        raise BoundsError()
                    if vs_get(cj[2][0]).values == [1]:
                        for idx in range(len(ci[2])):
                            vs_get(ci[2][idx])
                        for idx in (1, 2, 3, 7, 8, 9):
                            if idx > len(ci[1]):
                                raise BoundsError()
                            vs_get(ci[1][idx - 1])
                if len(cj[1]) < 3:
                    raise BoundsError()
                for jj in cj[1][0:3]:
                    st = vs_get(jj)
                    if st.unique:
                        continue
                    st.unique = True
                    st.is_known = True
                    requeue(jj)
        # ---- the queue (:805-1349)
        while len(q) >= 1:
            lead = q.popleft()
            pops += 1
            if pops > pop_cap:
                raise Watchdog()
            in_queue[lead - 1] = False
            if equation_solved[lead - 1]:
                continue
            eq = constraints[lead - 1]
            ka, kb, kc = nzk_a[lead - 1], nzk_b[lead - 1], nzk_c[lead - 1]

            # R1 check_unique (:827-873)
            def check_unique():
                nonlocal num_unique, successful_steps
                for i in kb:
                    if not vs_get(i).unique:
                        return False
                for i in ka:
                    if not vs_get(i).unique:
                        return False
                non_unique = -1
                for i in kc:
                    if not vs_get(i).unique:
                        if non_unique == -1:
                            non_unique = i
                        else:
                            return False
                if non_unique == -1:
                    return False
                st = vs_get(non_unique)
                st.unique = True
                num_unique += 1
                st.is_known = True
                successful_steps += 1
                requeue(non_unique)
                return True
            check_unique()

            # R2 check_quadratic (:875-942)
            def check_quadratic():
                nonlocal successful_steps
                if len(kc) >= 1:
                    return False
                unknown = -1
                for i in get_variables(eq):
                    if not vs_get(i).is_known:
                        if unknown == -1:
                            unknown = i
                        else:
                            return False
                slope_a = icpt_a = 0
                for i in ka:
                    if i == unknown:
                        slope_a = eq.a[i]
                    elif i == 1:
                        icpt_a = eq.a[i]
                    else:
                        return False
                slope_b = icpt_b = 0
                for i in kb:
                    if i == unknown:
                        slope_b = eq.b[i]
                    elif i == 1:
                        icpt_b = eq.b[i]
                    else:
                        return False
                old = vs_get(unknown)                 # variable_states[-1]: BoundsError comes before the divisions
                new = make_values(old, [divexact(neg(icpt_a), slope_a), divexact(neg(icpt_b), slope_b)])
                vs_set(unknown, new)
                if new.values == [0, 1] or new.values == [1, 0]:
                    vs_set(unknown, make_bounds(vs_get(unknown), 0, 1))
                requeue(unknown)
                equation_solved[lead - 1] = True
                successful_steps += 1
                return True
            check_quadratic()
            if len(ka) >= 1 or len(kb) >= 1:
                continue

            # R3 check_linear (:949-988)
            def check_linear():
                nonlocal successful_steps, num_unique
                non_one = [i for i in kc if i != 1]
                if len(non_one) != 1:
                    return False
                x = non_one[0]
                true_value = divexact(neg(eq.c[1]), eq.c[x])          # eq.c[1] inserts {1 => 0} when key 1 is absent
                st = vs_get(x)
                new_info = False
                if st.values != [true_value]:
                    st.values = [true_value]
                    successful_steps += 1
                    new_info = True
                st.lb = true_value
                st.ub = true_value
                if not st.unique:
                    st.unique = True
                    num_unique += 1
                    new_info = True
                st.is_known = True
                if new_info:
                    requeue(x)
            check_linear()

            # R4 checkBinary (:991-1076)
            def check_binary():
                nonlocal successful_steps, num_unique, eq
                ln = len(kc)
                if ln == 0:
                    return False
                t1 = sorted([1] + [neg(pow(2, i, P)) for i in range(0, ln - 1)])
                t2 = sorted([P - 1] + [pow(2, i, P) for i in range(0, ln - 1)])
                if sorted(eq.c.values()) == t2:
                    flipped = FDict()
                    for k, v in eq.c.items():
                        flipped[k] = neg(v)
                    constraints[lead - 1] = Equation(eq.a, eq.b, flipped)
                    eq = constraints[lead - 1]
                if sorted(eq.c.values()) != t1:
                    return False
                new_key = -1
                for i in kc:
                    if eq.c[i] == 1:
                        new_key = i
                    else:
                        st = vs_get(i)
                        if st.lb != 0 or st.ub != 1:
                            return False
                progress = False
                sk = vs_get(new_key)
                bound = (pow(2, ln - 1, P) - 1) % P
                if not (sk.lb == 0 and sk.ub == bound):
                    if sk.ub > (1 << (ln - 1)) - 1:
                        sk.lb = 0
                        sk.ub = bound
                        sk.is_known = True
                        progress = True
                        successful_steps += 1
                        requeue(new_key)
                if vs_get(new_key).unique:
                    for i in kc:
                        if i != new_key:
                            st = vs_get(i)
                            if not st.unique:
                                st.unique = True
                                num_unique += 1
                                st.is_known = True
                                progress = True
                                successful_steps += 1
                                requeue(i)
                return progress
            check_binary()

            # R5 checkpropagateBounds (:1078-1146)
            def check_propagate():
                nonlocal successful_steps, num_unique
                if len(kc) >= 3:
                    return False
                if sorted(eq.c.values()) != sorted([1, P - 1]):
                    return False
                x = eq.c.keys()
                key_1, key_2 = x[0], x[1]
                s1, s2 = vs_get(key_1), vs_get(key_2)
                changed = []
                if s2.ub != s1.ub or s2.lb != s1.lb or s2.unique != s1.unique:
                    if s2.unique != s1.unique:
                        # `variable_states[key_1] != make_unique(state_1)` compares two different mutable objects: always true
                        vs_get(key_1).is_known = True
                        vs_get(key_1).unique = True
                        changed.append(key_1)
                        num_unique += 1
                        vs_get(key_1).is_known = True          # (:1107-1108 write key_1 again)
                        vs_get(key_1).unique = True
                        num_unique += 1
                        changed.append(key_2)
                    mnub = min(s1.ub, s2.ub)
                    mxlb = max(s1.lb, s2.lb)
                    if s1.ub > mnub or s1.lb < mxlb:
                        t = vs_get(key_1)
                        t.is_known = True
                        t.lb = mxlb % P
                        t.ub = mnub % P
                        changed.append(key_1)
                    if s2.ub > mnub or s2.lb < mxlb:
                        t = vs_get(key_2)
                        t.is_known = True
                        t.lb = mxlb % P
                        t.ub = mnub % P
                        changed.append(key_2)
                    cs = JSet(changed)
                    successful_steps += len(cs)
                    for j in cs:
                        requeue(j)
                    return True
                return False
            check_propagate()

            # R6 checkOnePropagateBounds (:1148-1232)
            def check_one_propagate():
                nonlocal successful_steps, num_unique
                if len(kc) >= 4:
                    return False
                if sorted(eq.c.values()) != sorted([1, P - 1, P - 1]):
                    return False
                for k, v in eq.c.items():
                    if v == 1 and k != 1:
                        return False
                key_1 = key_2 = -1
                for k, v in eq.c.items():
                    if v == P - 1:
                        if key_1 == -1:
                            key_1 = k
                        else:
                            key_2 = k
                s1, s2 = vs_get(key_1), vs_get(key_2)
                changed = []
                if s2.ub != s1.ub or s2.lb != s1.lb or s2.unique != s1.unique:
                    if s2.unique != s1.unique:
                        s1.is_known = True
                        s1.unique = True
                        changed.append(key_1)
                        num_unique += 1
                        s2.is_known = True
                        s2.unique = True
                        num_unique += 1
                        changed.append(key_2)
                    mnub = min(s1.ub, s2.ub)
                    mxlb = max(s1.lb, s2.lb)
                    if mnub != 1 or mxlb != 0:
                        return False
                    if s1.ub > mnub or s1.lb < mxlb:
                        s1.is_known = True
                        s1.lb = mxlb
                        s1.ub = mnub
                        s1.values = [mnub, mxlb]
                        changed.append(key_1)
                    if s2.ub > mnub or s2.lb < mxlb:
                        s2.is_known = True
                        s2.lb = mxlb
                        s2.ub = mnub
                        s2.values = [mnub, mxlb]
                        changed.append(key_2)
                    cs = JSet(changed)
                    successful_steps += len(cs)
                    for j in cs:
                        requeue(j)
                    return True
                return False
            check_one_propagate()

            # R7 checkModularArithmetic (:1235-1298)
            def check_modular():
                nonlocal successful_steps, num_unique
                unknown = [a for a in kc if not vs_get(a).unique]
                if len(unknown) == 0:
                    return False

                def flip(x):
                    return x - P if x > _FLIP_THRESHOLD else x
                states = [vs_get(k) for k in unknown]
                coeffs = [abs(flip(eq.c[k])) for k in unknown]
                for s in states:
                    if not s.is_known:
                        return False
                r = sorted(range(len(coeffs)), key=lambda i: coeffs[i])       # sortperm: stable
                for i in range(len(r) - 1):
                    lo, hi = coeffs[r[i]], coeffs[r[i + 1]]
                    if lo == 0:
                        raise DivideError()
                    if hi % lo != 0 or hi // lo <= states[r[i]].ub - states[r[i]].lb:
                        return False
                if coeffs[r[-1]] * (states[r[-1]].ub + 1) > P:
                    return False
                successful_steps += len(unknown)
                for j in unknown:
                    st = vs_get(j)
                    st.unique = True
                    num_unique += 1
                    st.is_known = True
                    requeue(j)
                return True
            check_modular()

            # R8 checkAllButOneZeroGroup (:1304-1348)
            def check_abz():
                nonlocal successful_steps, num_unique
                abz_index = -1
                abzs = []
                for i in kc:
                    st = vs_get(i)
                    if st.unique:
                        continue
                    if st.abz != -1:
                        if abz_index == -1:
                            abz_index = st.abz
                            abzs.append(i)
                        elif st.abz != abz_index:
                            return False
                        else:
                            abzs.append(i)
                    else:
                        return False
                if len(abzs) == 0:
                    return False
                for i in abzs:
                    st = vs_get(i)
                    if st.unique:
                        continue
                    st.unique = True
                    num_unique += 1
                    successful_steps += 1
                    st.is_known = True
                    requeue(i)
            check_abz()
        # ---- P3 (:1357-1417)
        lin_freq = {}
        for i in range(1, nC + 1):
            c = constraints[i - 1]
            all_vars = get_variables(c)
            unknown_vars = []
            linear_eq = True
            for j in all_vars:
                if not vs_get(j).unique:
                    if j in nzk_a[i - 1] and j in nzk_b[i - 1]:
                        linear_eq = False
                        break
                    unknown_vars.append(j)
            if not linear_eq:
                continue
            c_linear = True
            for j in all_vars:
                if not vs_get(j).unique:
                    if j in nzk_a[i - 1] or j in nzk_b[i - 1] or j not in nzk_c[i - 1]:
                        c_linear = False
            if not c_linear:
                continue
            unknown_vars = sorted(unknown_vars)
            key = tuple(unknown_vars)
            rows = lin_freq.setdefault(key, [])
            rows.append([c.c[k] for k in unknown_vars])
            if len(rows) == len(unknown_vars):
                k = len(unknown_vars)
                res = 0
                for perm in itertools.permutations(range(k)):
                    term = 1
                    for j in range(k):
                        term = term * rows[j][perm[j]] % P
                    res = (res + parity(perm) * term) % P
                if res != 0 or (k == 1 and rows[0][0] != 0):
                    successful_steps += k
                    for nv in unknown_vars:
                        st = vs_get(nv)
                        st.unique = True
                        st.is_known = True
                        requeue(nv)
        # ---- P4 (:1425-1483)
        for i in range(1, nC + 1):
            if len(nzk_c[i - 1]) != 0:
                continue
            for j in nzk_a[i - 1]:
                if not vs_get(j).unique:
                    break
            if len(nzk_b[i - 1]) > 1:
                continue
            b_val = 0
            unique_b = True
            for j in nzk_b[i - 1]:
                if not vs_get(j).unique:
                    unique_b = False
                    b_val = j
            if unique_b:
                continue
            if len(nzk_a[i - 1]) > 2:
                continue
            slope = icpt = 0
            slope_index = 0
            for j in nzk_a[i - 1]:
                if j == 1:
                    icpt = constraints[i - 1].a[j]
                else:
                    slope = constraints[i - 1].a[j]
                    slope_index = j
            divexact(neg(icpt), slope)                 # root; `root in bad_values[slope_index]` is always false (the sets are empty)
            st = vs_get(b_val)
            if st.abz == -1:
                successful_steps += 1
            else:
                continue
            st.abz = slope_index
            st.is_known = True
            requeue(b_val)
        # ---- P5 (:1492-1550)
        for i in range(1, nC):
            if len(nzk_c[i]) != 0:
                continue
            if len(nzk_b[i]) != 1:
                continue
            if len(nzk_c[i - 1]) != 2:
                continue
            a_unique = True
            for j in nzk_a[i - 1]:
                if not vs_get(j).unique:
                    a_unique = False
                    break
            if not a_unique:
                continue
            if not dict_eq(constraints[i - 1].a, constraints[i].a):
                continue
            is_not_one = False
            var_key = 0
            for j in nzk_b[i]:
                if j != 1:
                    is_not_one = True
                    var_key = j
            if not is_not_one:
                continue
            bad_key = False
            for j in nzk_c[i - 1]:
                if j != 1 and j != var_key:
                    bad_key = True
            if bad_key:
                continue
            st = vs_get(var_key)
            if not st.unique:
                st.is_known = True
                st.unique = True
                successful_steps += 1
                equation_solved[i - 1] = True
                equation_solved[i] = True
                requeue(var_key)
    # ---- verdict (:1558-1597)
    R.unique_nontrivial = sum(1 for i in range(1, num_variables + 1) if variable_states[i - 1].unique and i in all_nontrivial)
    R.n_nontrivial = len(all_nontrivial)
    target_unique = 0
    for i in target_variables:
        if vs_get(i).unique:
            target_unique += 1
    R.unique_targets, R.n_targets = target_unique, len(target_variables)
    R.verdict = target_unique == len(target_variables)
    R.successful_steps, R.num_unique, R.pops, R.outer_iterations = successful_steps, num_unique, pops, outer
    R.states = variable_states
    R.status = 0
    return R


def run(main_path, trusted=(), names=(), secp_solve=False):
    """solveWithTrustedFunctions (:502-581) up to the Bool; returns a Result (status < 0: the exception the reference raises)"""
    R = Result()
    R.specials = []
    try:
        eqs, knowns, outs, nv = read_r1cs(main_path)
        fl = []
        for pth, nm in zip(trusted, names):
            e, k, o, _ = read_r1cs(pth)
            fl.append((nm, e, k, o))
        fl.sort(key=lambda x: -len(x[1]))             # stable (:527)
        specials = []
        red = eqs
        for nm, e, k, o in fl:
            new, red = abstraction(nm, red, k, e, o)
            specials.extend(new)
        R.specials = specials
        R.n_rows_reduced = len(red)
        out = solve(red, specials, knowns, outs, nv, secp_solve)
        out.specials, out.n_rows_reduced = specials, len(red)
        return out
    except (BoundsError, DivideError, UndefVarError, JlKeyError, Watchdog, FormatError) as e:
        R.status = e.status
        return R
