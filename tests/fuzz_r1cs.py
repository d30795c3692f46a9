"""Seeded random R1CS systems biased towards the shapes Ecne's rules look for (bit checks, x == y,
1 = x + y, binary decompositions in both orientations, single-variable rows, mixed-radix sums,
decoder / isZero pairs, small linear systems, products) plus degenerate rows (constants only,
explicit zero coefficients, duplicate wire ids, coefficients >= p).  Used by the oracle invariants
test and by the GPU parity fuzz."""
import random

import r1cs_py

P = r1cs_py.P


def make(seed, n_rows=None, n_vars=None, allow_errors=True):
    """Rows are built around a random witness so that the system is satisfiable, like a real circuit
    (contradictory single-variable or equality rows make the REFERENCE loop forever: two R3 rows with
    different values for one variable re-set each other, :966-985). A small fraction of seeds adds
    degenerate rows the reference raises on, or one contradiction (watchdog)."""
    rng = random.Random(seed)
    n_vars = n_vars or rng.randint(4, 48)
    n_rows = n_rows or rng.randint(1, 70)
    n_out = rng.randint(0, min(3, n_vars - 2))
    n_in = rng.randint(0, min(4, n_vars - 1 - n_out))
    nbits = rng.randint(1, max(1, n_vars // 2))
    bits = set(rng.sample(range(2, n_vars + 1), min(nbits, n_vars - 1)))
    w = {1: 1}
    for v in range(2, n_vars + 1):
        w[v] = rng.randint(0, 1) if v in bits else rng.choice([0, 1, 2, 3, 7, 255, 1 << 20, (1 << 86) - 5, P - 2])
    V = lambda: rng.randint(2, n_vars)                                       # noqa: E731
    B = lambda: rng.choice(sorted(bits))                                     # noqa: E731
    small = [1, -1, 2, -2, 3, 5, -7, 1 << 8, -(1 << 20), (1 << 86) + 3, P - 5, 1 << 250, -(1 << 251)]
    C = lambda: rng.choice(small)                                            # noqa: E731
    rows = []
    ev = lambda terms: sum(c * w[v] for v, c in terms) % P                   # noqa: E731

    def lin_fixed(terms):
        """0 = terms + k*one with k chosen so that the witness satisfies it"""
        k = (-ev(terms)) % P
        t = list(terms)
        if k:
            t.append((1, k))
        rng.shuffle(t)
        return ([], [], t)
    weird = allow_errors and rng.random() < 0.12
    for _ in range(n_rows):
        kind = rng.random()
        if kind < 0.14:      # bit check
            b = B()
            rows.append(([(b, 1), (1, -1)], [(b, 1)], []))
        elif kind < 0.24:    # x == y between equal-valued variables (either orientation)
            x = V()
            same = [v for v in range(1, n_vars + 1) if v != x and w[v] == w[x]]
            if same:
                y = rng.choice(same)
                s = rng.choice([1, -1])
                rows.append(([], [], [(x, s), (y, -s)]))
        elif kind < 0.30:    # 1 = x + y on complementary bits
            x = B()
            comp = [v for v in bits if v != x and w[v] == 1 - w[x]]
            if comp:
                rows.append(([], [], [(1, 1), (x, -1), (rng.choice(comp), -1)]))
        elif kind < 0.42:    # binary decomposition of a fresh-valued sum variable, T or T2 orientation
            k = rng.randint(1, 6)
            bs = rng.sample(sorted(bits), min(k, len(bits)))
            tot = sum(w[b] << i for i, b in enumerate(bs))
            cands = [v for v in range(2, n_vars + 1) if v not in bits and v not in bs]
            if cands:
                sv = rng.choice(cands)
                if w[sv] != tot and not any(sv in [vv for part in r for vv, _c in part] for r in rows):
                    w[sv] = tot
                if w[sv] == tot:
                    s = rng.choice([1, -1])
                    t = [(sv, s)] + [(b, -s * (1 << i)) for i, b in enumerate(bs)]
                    rng.shuffle(t)
                    rows.append(([], [], t))
        elif kind < 0.50:    # single-variable row fixing x to its witness value
            x = V()
            rows.append(lin_fixed([(x, C())]))
        elif kind < 0.58:    # mixed radix sum over bounded digits
            k = rng.randint(2, 4)
            vs = rng.sample(range(2, n_vars + 1), min(k, n_vars - 1))
            radix, t = 1, []
            for v in vs:
                t.append((v, radix))
                radix *= rng.choice([2, 3, 4])
            rows.append(lin_fixed(t))
        elif kind < 0.66:    # decoder row / isZero pair
            x, y, z = V(), V(), V()
            if rng.random() < 0.5:
                kk = rng.randint(0, 3)
                if (w[x] - kk) % P == 0 or w[y] == 0:
                    rows.append(([(x, 1)] + ([(1, -kk)] if kk else []), [(y, 1)], []))
            elif y != 1 and (w[x] * w[y]) % P == 0:
                rows.append(([(x, 1)], [(z, 1)], [(1, 1), (y, -1)]))
                rows.append(([(x, 1)], [(y, 1)], []))
        elif kind < 0.76:    # small linear system rows over the same unknowns
            k = rng.randint(2, 3)
            vs = rng.sample(range(2, n_vars + 1), min(k, n_vars - 1))
            for _r in range(rng.randint(1, k + 1)):
                rows.append(lin_fixed([(v, C()) for v in vs]))
        elif kind < 0.92:    # product a*b = c + const
            a = [(V(), C()) for _ in range(rng.randint(1, 2))]
            b = [(V(), C()) for _ in range(rng.randint(1, 2))]
            c = [(V(), C()) for _ in range(rng.randint(0, 2))]
            k = (ev(a) * ev(b) - ev(c)) % P
            if k:
                c.append((1, k))
            rows.append((a, b, c))
        elif kind < 0.96:    # general sum
            vs = rng.sample(range(2, n_vars + 1), min(rng.randint(2, 6), n_vars - 1))
            rows.append(lin_fixed([(v, C()) for v in vs]))
        elif kind < 0.98:    # explicit zero coefficient / duplicate wire id (last wins)
            x, y = V(), V()
            c2 = rng.choice([-1, 2, 5])
            k = (-(c2 * w[x])) % P
            rows.append(([], [], [(x, 1), (y, 0), (x, c2)] + ([(1, k)] if k else [])))
        else:                # coefficient written un-reduced (>= p):  x - w[x] = 0 with both terms + p
            x = V()
            rows.append(([], [], [(x, P + 1), (1, P + (-w[x]) % P)]))
        if weird and rng.random() < 0.08:
            r = rng.random()
            if r < 0.35:
                rows.append(([(1, 2)], [(1, 3)], []))                 # constants only: BoundsError (:916)
            elif r < 0.7:
                rows.append(([(1, 1)], [(V(), 1)], []))               # slope missing from A: DivideError (:919)
            else:
                x = V()
                rows.append(([], [], [(x, 1), (1, (-(w[x] + 1)) % P)]))   # contradicts the witness
    return dict(n_wires=n_vars - 1, n_out=n_out, n_pub=0, n_prv=n_in, rows=rows, witness=w)


def make_wide(seed, scale=1):
    """A larger random system (100-400 variables, 150-700 small rows from make(); `scale` multiplies both and
    the number and length of the long rows) with LONG rows woven in --
    plain sums of 65-300 terms, decoder groups closed by a long sum (R8), mixed-radix sums over bit
    variables (R7), long binary decompositions (R4 shape: popped alone) -- over a mix of the base system's
    variables and fresh ones, all consistent with one witness. Exercises the engine's long-row paths
    (rows riding along in rounds, wavefront rounds, multi-workgroup rounds) in random surroundings."""
    rng = random.Random(1000003 * seed + 17)
    base = make(seed + 5000, n_rows=scale * rng.randint(150, 700), n_vars=scale * rng.randint(100, 400), allow_errors=False)
    w = dict(base["witness"])
    rows = list(base["rows"])
    nv = base["n_wires"] + 1

    def fresh(val):
        nonlocal nv
        nv += 1
        w[nv] = val % P
        return nv

    def fix(v):      # v = its witness value (R3 makes it unique)
        rows.append(([], [], [(v, 1), (1, (-w[v]) % P)] if w[v] else [(v, 1)]))

    def bit():
        b = fresh(rng.randint(0, 1))
        rows.append(([(b, 1), (1, -1)], [(b, 1)], []))
        return b

    old = list(range(2, base["n_wires"] + 2))
    for _ in range(scale * rng.randint(2, 6)):
        kind = rng.random()
        n = rng.randint(65, 300 if scale == 1 else 1100)
        if kind < 0.3:       # plain long sum, sometimes every addend pinned (R1 fires), sometimes not
            xs = [rng.choice(old) if rng.random() < 0.5 else fresh(rng.choice([0, 1, 5, 1 << 40])) for _ in range(n)]
            xs = list(dict.fromkeys(xs))
            cs = [rng.choice([1, 2, 3, -1, 7, 1 << 30]) for _ in xs]
            y = fresh(sum(c * w[x] for c, x in zip(cs, xs)))
            rows.append(([], [], [(y, -1)] + list(zip(xs, cs))))
            if rng.random() < 0.6:
                for x in xs:
                    if rng.random() < 0.9:
                        fix(x)
        elif kind < 0.55:    # decoder: out_i * (inp - i) = 0, sum out_i = s
            t = rng.randrange(n)
            inp = fresh(t)
            outs = [fresh(1 if i == t else 0) for i in range(n)]
            sv = fresh(1)
            for i, o in enumerate(outs):
                rows.append(([(inp, 1)] + ([(1, -i)] if i else []), [(o, 1)], []))
            rows.append(([], [], [(sv, -1)] + [(o, 1) for o in outs]))
            if rng.random() < 0.7:
                fix(sv)
            if rng.random() < 0.5:
                fix(inp)
        elif kind < 0.8:     # mixed radix over bits, not the binary-decomposition pattern (coefficient 5 on y)
            n = min(n, 250)
            bs = [bit() for _ in range(n)]
            tot = sum(w[b] << i for i, b in enumerate(bs))
            y = fresh(tot * pow(5, -1, P))
            rows.append(([], [], [(y, 5)] + [(b, -(1 << i)) for i, b in enumerate(bs)]))
            if rng.random() < 0.7:
                fix(y)
        else:                # long binary decomposition (either orientation)
            n = min(n, 200)
            bs = [bit() for _ in range(n)]
            y = fresh(sum(w[b] << i for i, b in enumerate(bs)))
            sgn = rng.choice([1, -1])
            rows.append(([], [], [(y, sgn)] + [(b, -sgn * (1 << i)) for i, b in enumerate(bs)]))
            if rng.random() < 0.7:
                fix(y)
    order = list(range(len(rows)))
    if rng.random() < 0.5:
        rng.shuffle(order)        # long rows early / late / in between
    return dict(n_wires=nv - 1, n_out=base["n_out"], n_pub=0, n_prv=base["n_prv"], rows=[rows[i] for i in order], witness=w)


def make_decomp(seed):
    """Systems made of LONG BINARY DECOMPOSITIONS (the R4 shape, :991-1076, l = 16..120 terms, both orientations -- the reference negates a
    row of the second one at its first visit, :1001-1011 --, the pivot anywhere in the row) in the states the engine's shortcuts tell apart
    (fastrow.hip.hpp: long_r4_idle, exec_long_r4, long_r4_done, the watched pair): bits with and without their bit check (bounds [0,1] or
    not), the lowest bit pinned / unbounded / plain, the pivot pinned at once, pinned late (behind a chain of equalities and products: the
    row is re-queued while pivot and bits are not unique), never, or shared by a SECOND decomposition of another length (the pivot's bounds
    then come cut already, or get cut twice); rows in any order. Everything consistent with one witness."""
    rng = random.Random(7919 * seed + 3)
    w = {1: 1}
    rows = []
    nv = 1

    def fresh(val):
        nonlocal nv
        nv += 1
        w[nv] = val % P
        return nv

    def fix(v):
        rows.append(([], [], [(v, 1), (1, (-w[v]) % P)] if w[v] else [(v, 1)]))

    def bitcheck(b):
        rows.append(([(b, 1), (1, -1)], [(b, 1)], []))

    def late(v, depth):
        """v becomes unique `depth` pops later: v = u_1, u_1 = u_2, ..., u_depth pinned (x == y rows, or products with a pinned 1)"""
        cur = v
        for _ in range(depth):
            u = fresh(w[cur])
            if rng.random() < 0.6:
                rows.append(([], [], [(cur, 1), (u, -1)]))
            else:
                one = fresh(1)
                fix(one)
                rows.append(([(u, 1)], [(one, 1)], [(cur, 1)]))
            cur = u
        fix(cur)

    pivots = []
    for _ in range(rng.randint(2, 7)):
        l = rng.choice([rng.randint(16, 40), rng.randint(41, 90), rng.randint(91, 120), rng.randint(3, 15)])
        nb = l - 1
        p_check = rng.choice([1.0, 1.0, 1.0, 1.0, 0.97, 0.8, 0.0])
        bs = [fresh(rng.randint(0, 1)) for _ in range(nb)]
        for i, b in enumerate(bs):
            if rng.random() < p_check:
                bitcheck(b)
        # (a pinned bit has bounds [v, v], not [0, 1]: R4 stops there for good, :1020-1029 -- one decomposition in five gets one)
        k0 = rng.random()
        if k0 < 0.08:
            fix(bs[0])
        elif k0 < 0.2:
            fix(rng.choice(bs))
        if pivots and rng.random() < 0.3:
            y = rng.choice(pivots)      # a second decomposition of a pivot: its value has to fit
            val = w[y]
            if val >> nb:
                y = fresh(sum(w[b] << i for i, b in enumerate(bs)))
            else:
                for i, b in enumerate(bs):
                    w[b] = (val >> i) & 1
        else:
            y = fresh(sum(w[b] << i for i, b in enumerate(bs)))
        if y not in pivots:
            pivots.append(y)
            k = rng.random()
            if k < 0.25:
                fix(y)
            elif k < 0.7:
                late(y, rng.randint(1, 6))
            elif k < 0.8:
                z = fresh(w[y])                  # an equal variable that is never pinned: bounds travel, nothing becomes unique
                rows.append(([], [], [(y, 1), (z, -1)]))
        sgn = rng.choice([1, -1])
        terms = [(y, sgn)] + [(b, -sgn * (1 << i)) for i, b in enumerate(bs)]
        if rng.random() < 0.7:
            rng.shuffle(terms)
        rows.append(([], [], terms))
    order = list(range(len(rows)))
    if rng.random() < 0.6:
        rng.shuffle(order)
    return dict(n_wires=nv - 1, n_out=0, n_pub=0, n_prv=0, rows=[rows[i] for i in order], witness=w)


def make_oob(seed):
    """A random system in which some rows name variable ids ABOVE num_variables (nWires + 1 .. nWires + 5): the reference sizes
    `variable_states` by num_variables (:681) and raises BoundsError at the first rule that READS such a state -- lazily, since
    `variable_to_indices` is a DefaultDict (:628). Which read comes first depends on the order a rule walks the row in and on its
    early exits, so the cases are built around those: ids in A, B or C of rows that are popped at once / only later / never, behind
    or in front of a variable that is not unique, next to rows that raise DivideError, with a zero coefficient (never read), in
    isZero pairs, and rows whose P3 visit ends at a non-unique variable of A n B before it reaches the id (no error at all)."""
    rng = random.Random(424243 * seed + 7)
    base = make(seed + 70000, allow_errors=rng.random() < 0.35)
    rows = list(base["rows"])
    nv = base["n_wires"] + 1
    w = dict(base["witness"])
    W = lambda: nv + rng.randint(1, 5)                                       # noqa: E731
    V = lambda: rng.randint(2, nv)                                           # noqa: E731
    for k in range(1, 6):
        w[nv + k] = rng.choice([0, 1, 5])
    mode = rng.random()
    if mode < 0.45 and rows:        # an existing row names such an id instead of one of its variables
        for _ in range(rng.randint(1, 3)):
            i = rng.randrange(len(rows))
            parts = [list(p) for p in rows[i]]
            cand = [p for p in range(3) if parts[p]]
            if not cand:
                continue
            p = rng.choice(cand)
            e = rng.randrange(len(parts[p]))
            parts[p][e] = (W(), parts[p][e][1])
            rows[i] = tuple(parts)
    if mode >= 0.3:
        quiet = rng.random() < 0.45     # only rows that may never be read
        for _ in range(rng.randint(1, 4)):
            j, k, x, ww = V(), V(), V(), W()
            kind = rng.random()
            if quiet:
                kind = rng.choice([0.05, 0.5])
            if kind < 0.25:      # j in A and B: P3's walk ends at j while j is not unique (:1366); R1 stops at B's j
                new = [([(j, 1)], [(j, 1)], [(ww, 1)])] if rng.random() < 0.5 else [([(j, 1), (ww, 3)], [(j, 1)], [(x, 1)])]
            elif kind < 0.4:     # the id behind other factors
                new = [([(j, 1), (ww, 1)], [(k, 1)], [(x, 1)])]
            elif kind < 0.55:    # zero coefficient: never read
                new = [([], [], [(x, 1), (ww, 0), (1, (-w[x]) % P)])]
            elif kind < 0.7:     # C empty: R2 walks getVariables until the second variable that is not known
                new = [([(ww, 1)], [(j, 1)], [])] if rng.random() < 0.5 else [([(j, 1)], [(j, 1), (ww, 2)], [])]
            elif kind < 0.85:    # linear rows
                new = [([], [], [(ww, 1), (x, P - 1)])] if rng.random() < 0.5 else [([], [], [(x, 1), (j, 2), (ww, 4), (k, 8)])]
            else:                # isZero pair with the id as y, or in the shared A
                a = [(j, 1), (ww, 1)] if rng.random() < 0.5 else [(j, 1)]
                y = ww if len(a) == 1 or rng.random() < 0.3 else x
                new = [(a, [(k, 1)], [(1, 1), (y, P - 1)]), (a, [(y, 1)], [])]
            pos = rng.randint(0, len(rows))
            rows[pos:pos] = new
    return dict(n_wires=base["n_wires"], n_out=base["n_out"], n_pub=base["n_pub"], n_prv=base["n_prv"], rows=rows, witness=w)


def make_oob_p4(seed):
    """Systems around P4's two state reads on rows that name an id above num_variables (src/R1CSConstraintSolver.jl:1430-1436 `unique_a`
    walks nzk_a in ITS Set order while unique -- before the `length(nzk_b[i]) > 1` test --, :1443 reads B's only key). P3's walk
    (:1364-1372, `getVariables` order over A u B u C) comes first and raises unless a non-unique variable of A n B stands in front of the
    id in THAT order; the two orders differ when the tables differ in size, so B carries 0..20 further entries (inputs, mostly). Around
    them: rows that make the A n B variable unique at once / in a later outer iteration / never, rows that divide by zero in P4 (:1467)
    below and above, bit checks on the same variables (R2 walks getVariables too), isZero pairs sharing the A. Returns raw rows."""
    import ref2
    rng = random.Random(9176 * seed + 11)
    npub = rng.randint(8, 26)
    nprv = rng.randint(0, 3)
    nout = rng.randint(1, 2)
    nint = rng.randint(4, 14)
    nw = nout + npub + nprv + nint             # wires 0..nw-1, nVars = nw + 1 (variable ids 1..nw+1; the last id is internal too)
    nv = nw + 1
    outs = list(range(2, 2 + nout))
    ins = list(range(2 + nout, 2 + nout + npub + nprv))
    internal = list(range(2 + nout + npub + nprv, nv + 1))
    Wid = lambda: nv + rng.randint(1, 90)                                    # noqa: E731
    coef = lambda: rng.choice([1, 1, 1, 2, 5, P - 1, 7])                     # noqa: E731
    rows = []
    # filler: products of inputs into internal variables (R1 at the first pop), chains, a constant row
    for x in rng.sample(internal, rng.randint(0, min(4, len(internal)))):
        a, b = rng.choice(ins), rng.choice(ins)
        rows.append(([(a, 1)], [(b, 1)], [(x, 1)]))
    if rng.random() < 0.4:
        x, y = rng.choice(internal), rng.choice(internal)
        rows.append(([], [], [(x, 1), (y, P - 1)]))
    for _ in range(rng.choice([1, 1, 1, 2, 3])):
        v = rng.choice(internal + outs)        # the variable of A n B (unique or not, depending on the filler)
        w = Wid()
        kind = rng.random()
        a = [(v, coef()), (w, coef())]
        if kind < 0.15:
            a.append((rng.choice(ins), 1))
        elif kind < 0.25:
            a.append((1, coef()))
        elif kind < 0.32:
            a.append((rng.choice(internal), 1))
        elif kind < 0.36:
            a = [(w, coef())]                  # the id alone in A
        rng.shuffle(a)
        nb = rng.choice([0, 1, 2, 5, 8, 9] + list(range(10, 21)) * 2)
        b = [(x, 1) for x in rng.sample(ins, min(nb, len(ins)))]
        r = rng.random()
        if r < 0.8:
            b.insert(rng.randint(0, len(b)), (v, coef()))
        elif r < 0.88:
            b = [(Wid(), 1)] if rng.random() < 0.5 else [(w, 1)]            # B's only key is such an id (:1443)
        elif r < 0.94:
            b.insert(rng.randint(0, len(b)), (rng.choice(internal), 1))
        if rng.random() < 0.1:
            b.append((w, 0))                   # explicit zero: not a key
        c = [] if rng.random() < 0.9 else [(rng.choice(ins), 1)]
        if rng.random() < 0.85 and any(x == w for x, _c in a) and any(x == v for x, _c in a) and any(x == v for x, _c in b):
            # steer towards the case only P4 decides: v in front of the id in getVariables order (P3 ends at v), the id in front of v
            # in nzk_a's own order. The orders are Julia's (second reading's Set model); try other ids until they come out that way.
            for _try in range(60):
                fa, fb = ref2.FDict(), ref2.FDict()
                for x, cf in a:
                    fa[x] = cf % P
                for x, cf in b:
                    fb[x] = cf % P
                gv = list(ref2.get_variables(ref2.Equation(fa, fb, ref2.FDict())))
                na = list(ref2.nonzero_keys(fa))
                if w in gv and v in gv and gv.index(v) < gv.index(w) and na.index(w) < na.index(v):
                    break
                w2 = Wid()
                a = [(w2 if x == w else x, cf) for x, cf in a]
                w = w2
        new = [(a, b, c)]
        if rng.random() < 0.12:                # an isZero-shaped pair on the same A (P5 :1503 walks it as well)
            y = rng.choice(internal)
            new = [(a, [(rng.choice(ins), 1)], [(1, 1), (y, P - 1)]), (a, [(y, 1)], [])]
        pos = rng.randint(0, len(rows))
        rows[pos:pos] = new
    for _ in range(rng.choice([0] * 10 + [1, 2])):         # P4 rows that divide by zero (no slope variable in A), somewhere
        x = rng.choice(internal + outs)
        a = [] if rng.random() < 0.3 else [(1, coef())]
        rows.insert(rng.randint(0, len(rows)), (a, [(x, 1)], []))
    for _ in range(rng.randint(0, 2)):         # ordinary P4 rows / bit checks
        x = rng.choice(internal)
        if rng.random() < 0.5:
            rows.insert(rng.randint(0, len(rows)), ([(x, 1), (1, P - 1)], [(x, 1)], []))
        else:
            rows.insert(rng.randint(0, len(rows)), ([(rng.choice(internal), 1), (1, 3)], [(x, 1)], []))
    if rng.random() < 0.3:                     # makes a variable unique only in the second outer iteration (P3 single-row group -> R1)
        x, y = rng.choice(internal), rng.choice(internal)
        rows.append(([(rng.choice(ins), 1)], [(x, 1)], [(y, 1)]))
    return dict(n_wires=nw, n_out=nout, n_pub=npub, n_prv=nprv, rows=rows)


def write(path, spec):
    rows = []
    for A, B, C in spec["rows"]:
        rows.append(([(v, c % (1 << 256) if c >= 0 else c % P) for v, c in A],
                     [(v, c % (1 << 256) if c >= 0 else c % P) for v, c in B],
                     [(v, c % (1 << 256) if c >= 0 else c % P) for v, c in C]))
    write_raw(path, spec["n_wires"], spec["n_out"], spec["n_pub"], spec["n_prv"], rows)


def write_raw(path, nwires, nout, npub, nprv, rows, section_order=(2, 1, 3)):
    """like r1cs_py.write but keeps coefficients un-reduced (< 2^256) and lets the caller pick the
    section order"""
    import struct
    body2 = bytearray()
    for parts in rows:
        for terms in parts:
            body2 += struct.pack("<I", len(terms))
            for v, c in terms:
                body2 += struct.pack("<I", v - 1) + int(c).to_bytes(32, "little")
    body1 = struct.pack("<I", 32) + P.to_bytes(32, "little") + struct.pack("<IIII", nwires, nout, npub, nprv)
    body1 += struct.pack("<QI", nwires, len(rows))
    body3 = b"".join(struct.pack("<Q", i) for i in range(nwires))
    bodies = {1: body1, 2: bytes(body2), 3: body3}
    out = b"r1cs" + struct.pack("<II", 1, 3)
    for t in section_order:
        out += struct.pack("<IQ", t, len(bodies[t])) + bodies[t]
    with open(path, "wb") as f:
        f.write(out)
