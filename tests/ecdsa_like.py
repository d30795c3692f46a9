"""Deterministic synthetic stand-in for the reference's north-star input `ecdsa.r1cs`.

`ecdsa.r1cs` (= ECDSAPrivToPub(86, 3) compiled with circom --O0, /root/reference/
Circom_Functions/ecdsa.circom:14-128) is not in the reference tree (README.md:59, .gitignore:1)
and circom is not available, so BASELINE.json's config 5 runs on `ecdsa_like(S)`: the same
template structure written out row by row in the shapes circom 2.0 --O0 emits (decoded from the
reference's own fixtures):

  * main-level wiring rows `0 = rhs - lhs` for every `<==`, constant rows `0 = c*one - x` for the
    2 x S x 3 x 2^stride table inputs, the three select rows per limb (ecdsa.circom:116-118)
  * Num2Bits(86) x 3, Bits2Num(stride) x S, Multiplexer(3, 2^stride) x 2S (Decoder +
    EscalarProduct x 3), IsZero x S, OR x S
  * S-1 verbatim copies of all 15 935 rows of secp256k1.r1cs (Secp256k1AddUnequal(86,3)) in file
    order with wires renumbered by a constant offset, so that abstraction() finds them
    (src/R1CSConstraintSolver.jl:259-351)

S = 26, stride = 10 reproduces the shape of the real circuit (~1.09 M rows). Table constants are
fixed 86-bit values derived from a counter (any non-zero constants exercise the same rules).
No RNG. The component order follows circom's: a template's own constraints first, then its
sub-components in creation order.
"""
import os
import struct

import numpy as np

import r1cs_py

P = r1cs_py.P


class _Builder:
    def __init__(self):
        self.nvars = 1            # variable 1 = constant one
        self.counts = []          # per row-part term counts (3 per row)
        self.wires = []           # flat 1-based variable ids
        self.coefs = []           # flat python ints (mod P)

    def alloc(self, n=1):
        first = self.nvars + 1
        self.nvars += n
        return first

    def row(self, A, B, C):
        for part in (A, B, C):
            self.counts.append(len(part))
            for v, c in part:
                self.wires.append(v)
                self.coefs.append(c % P)

    def wire(self, lhs, rhs):
        """`lhs <== rhs` between two signals: 0 = rhs - lhs"""
        self.row([], [], [(lhs, -1), (rhs, 1)])

    def n_rows(self):
        return len(self.counts) // 3


def _num2bits(b, n):
    out = b.alloc(n)
    inp = b.alloc(1)

    def emit():
        for i in range(n):
            b.row([(out + i, 1), (1, -1)], [(out + i, 1)], [])
        b.row([], [], [(out + i, -(1 << i)) for i in range(n)] + [(inp, 1)])
    return dict(out=out, inp=inp, emit=emit)


def _bits2num(b, n):
    out = b.alloc(1)
    inp = b.alloc(n)

    def emit():
        b.row([], [], [(inp + i, 1 << i) for i in range(n)] + [(out, -1)])
    return dict(out=out, inp=inp, emit=emit)


def _iszero(b):
    out = b.alloc(1)
    inp = b.alloc(1)
    inv = b.alloc(1)

    def emit():
        b.row([(inp, 1)], [(inv, 1)], [(out, -1), (1, 1)])
        b.row([(inp, 1)], [(out, 1)], [])
    return dict(out=out, inp=inp, emit=emit)


def _or(b):
    out = b.alloc(1)
    a = b.alloc(1)
    bb = b.alloc(1)

    def emit():
        b.row([(a, 1)], [(bb, 1)], [(a, 1), (out, -1), (bb, 1)])
    return dict(out=out, a=a, b=bb, emit=emit)


def _multiplexer(b, w_in, n_in):
    out = b.alloc(w_in)
    inp = b.alloc(n_in * w_in)       # inp[k][j] at inp + k*w_in + j
    sel = b.alloc(1)
    # Decoder(n_in): out[n_in], success, inp
    d_out = b.alloc(n_in)
    d_success = b.alloc(1)
    d_inp = b.alloc(1)
    eps = []
    for _j in range(w_in):            # EscalarProduct(n_in): out, in1[n], in2[n], aux[n]
        e_out = b.alloc(1)
        e_in1 = b.alloc(n_in)
        e_in2 = b.alloc(n_in)
        e_aux = b.alloc(n_in)
        eps.append((e_out, e_in1, e_in2, e_aux))

    def emit():
        b.wire(d_inp, sel)
        for j in range(w_in):
            e_out, e_in1, e_in2, _e_aux = eps[j]
            for k in range(n_in):
                b.wire(e_in1 + k, inp + k * w_in + j)
                b.wire(e_in2 + k, d_out + k)
            b.wire(out + j, e_out)
        b.row([], [], [(1, 1), (d_success, -1)])
        # Decoder
        for k in range(n_in):
            A = [(d_inp, 1)] + ([(1, -k)] if k else [])
            b.row(A, [(d_out + k, 1)], [])
        b.row([], [], [(d_success, -1)] + [(d_out + k, 1) for k in range(n_in)])
        b.row([(d_success, 1), (1, -1)], [(d_success, 1)], [])
        # EscalarProducts
        for j in range(w_in):
            e_out, e_in1, e_in2, e_aux = eps[j]
            for k in range(n_in):
                b.row([(e_in1 + k, -1)], [(e_in2 + k, 1)], [(e_aux + k, -1)])
            b.row([], [], [(e_out, -1)] + [(e_aux + k, 1) for k in range(n_in)])
    return dict(out=out, inp=inp, sel=sel, emit=emit)


def _table_const(i, l, j, idx, seed=0):
    """seed 0 = the committed workload (golden digests); another seed = another table (bench.py --gpus N: one independent job per rank)"""
    x = (i * 7919 + l * 104729 + j * 1299709 + idx * 15485863 + 0x1234567 + seed * 0x5DEECE66D) * 0x9E3779B97F4A7C15
    x = (x ^ (x >> 31)) % (1 << 86)
    return x | 1


def generate(path, S=26, stride=10, nbits=86, k=3, adder_rel="secp256k1.r1cs", adder_path=None, seed=0):
    """Writes ecdsa_like(S) to `path`; returns a dict with its sizes."""
    import fixtures
    adder_path = adder_path or fixtures.path(adder_rel)
    a_hdr, a_rows = r1cs_py.parse_file(adder_path)
    a_nvars = a_hdr["nWires"] + 1                 # adder variables 1..a_nvars, var 1 = one
    assert a_hdr["nPubOut"] == 2 * k and a_hdr["nPrvIn"] + a_hdr["nPubIn"] == 4 * k

    b = _Builder()
    n_in = 1 << stride
    pub = b.alloc(2 * k)                         # pubkey[l][i] at pub + l*k + i     (vars 2..7)
    priv = b.alloc(k)                            # privkey[i]                         (vars 8..10)
    partial = b.alloc(S * 2 * k)                 # partial[i][l][idx]
    inter1 = b.alloc(max(S - 1, 0) * 2 * k)
    inter2 = b.alloc(max(S - 1, 0) * 2 * k)

    def P_(i, l, idx):
        return partial + (i * 2 + l) * k + idx

    def I1(i, l, idx):
        return inter1 + (i * 2 + l) * k + idx

    def I2(i, l, idx):
        return inter2 + (i * 2 + l) * k + idx

    n2b = [_num2bits(b, nbits) for _ in range(k)]
    sels = [_bits2num(b, stride) for _ in range(S)]
    mux = [[_multiplexer(b, k, n_in) for _l in range(2)] for _i in range(S)]
    isz = [_iszero(b) for _ in range(S)]
    hpn = [_or(b) for _ in range(S)]
    adders = []                                  # variable offset of each adder block
    for _ in range(S - 1):
        first = b.alloc(a_nvars - 1)             # adder var v (>= 2) -> first + (v - 2)
        adders.append(first - 2)

    # ---- main template's own constraints, in source order (ecdsa.circom:19-127)
    for i in range(k):
        b.wire(n2b[i]["inp"], priv + i)
    for i in range(S):
        for j in range(stride):
            bit = i * stride + j
            if bit // nbits < k:
                b.wire(sels[i]["inp"] + j, n2b[bit // nbits]["out"] + bit % nbits)
            else:
                b.row([], [], [(sels[i]["inp"] + j, -1)])          # `<== 0`
    for i in range(S):
        for l in range(2):
            m = mux[i][l]
            b.wire(m["sel"], sels[i]["out"])
            for idx in range(k):
                for j in range(n_in):
                    b.row([], [], [(1, _table_const(i, l, j, idx, seed)), (m["inp"] + j * k + idx, -1)])
    for i in range(S):
        b.wire(isz[i]["inp"], sels[i]["out"])
    b.row([], [], [(hpn[0]["a"], -1)])
    b.row([], [], [(1, 1), (isz[0]["out"], -1), (hpn[0]["b"], -1)])
    for i in range(1, S):
        b.wire(hpn[i]["a"], hpn[i - 1]["out"])
        b.row([], [], [(1, 1), (isz[i]["out"], -1), (hpn[i]["b"], -1)])
    for idx in range(k):
        for l in range(2):
            b.wire(P_(0, l, idx), mux[0][l]["out"] + idx)
    for i in range(1, S):
        off = adders[i - 1]
        a_out = lambda l, idx: off + 2 + l * k + idx                  # noqa: E731  out[l][idx] = vars 2..7
        a_a = lambda l, idx: off + 2 + 2 * k + l * k + idx            # noqa: E731  a[l][idx]   = vars 8..13
        a_b = lambda l, idx: off + 2 + 4 * k + l * k + idx            # noqa: E731  b[l][idx]   = vars 14..19
        for idx in range(k):
            for l in range(2):
                b.wire(a_a(l, idx), P_(i - 1, l, idx))
                b.wire(a_b(l, idx), mux[i][l]["out"] + idx)
        for idx in range(k):
            for l in range(2):
                z, h = isz[i]["out"], hpn[i - 1]["out"]
                mo = mux[i][l]["out"] + idx
                b.row([(z, -1)], [(P_(i - 1, l, idx), 1), (a_out(l, idx), -1)],
                      [(a_out(l, idx), 1), (I1(i - 1, l, idx), -1)])
                b.row([(z, 1)], [(mo, 1)], [(mo, 1), (I2(i - 1, l, idx), -1)])
                b.row([(h, -1)], [(I1(i - 1, l, idx), 1), (I2(i - 1, l, idx), -1)],
                      [(I2(i - 1, l, idx), 1), (P_(i, l, idx), -1)])
    for i in range(k):
        for l in range(2):
            b.wire(pub + l * k + i, P_(S - 1, l, i))
    # ---- sub-components in creation order
    for c in n2b:
        c["emit"]()
    for c in sels:
        c["emit"]()
    n_before_mux = b.n_rows()
    for i in range(S):
        for l in range(2):
            mux[i][l]["emit"]()
    for c in isz:
        c["emit"]()
    for c in hpn:
        c["emit"]()

    # ---- serialise: builder rows, then the adder blocks (vectorised renumbering of a template)
    counts = np.asarray(b.counts, dtype=np.uint32)
    wires = np.asarray(b.wires, dtype=np.uint32)
    nterms = len(wires)
    coef_bytes = np.zeros((nterms, 32), dtype=np.uint8)
    # most coefficients are tiny or p - tiny: convert through python ints once
    cb = b"".join(int(c).to_bytes(32, "little") for c in b.coefs)
    coef_bytes[:] = np.frombuffer(cb, dtype=np.uint8).reshape(nterms, 32)

    def serialise(counts, wires0, coef_bytes):
        """counts: per part; wires0: 0-based wire ids; -> bytes of the constraint section body"""
        nparts, nt = len(counts), len(wires0)
        total = 4 * nparts + 36 * nt
        out = np.zeros(total, dtype=np.uint8)
        starts = np.zeros(nparts, dtype=np.int64)        # byte offset of each part's count word
        terms_before = np.concatenate(([0], np.cumsum(counts)[:-1])).astype(np.int64)
        starts[:] = 4 * np.arange(nparts, dtype=np.int64) + 36 * terms_before
        cview = counts.astype("<u4").view(np.uint8).reshape(nparts, 4)
        for bidx in range(4):
            out[starts + bidx] = cview[:, bidx]
        part_of_term = np.repeat(np.arange(nparts, dtype=np.int64), counts)
        idx_in_part = np.arange(nt, dtype=np.int64) - terms_before[part_of_term]
        tpos = starts[part_of_term] + 4 + 36 * idx_in_part
        wview = wires0.astype("<u4").view(np.uint8).reshape(nt, 4)
        for bidx in range(4):
            out[tpos + bidx] = wview[:, bidx]
        for bidx in range(32):
            out[tpos + 4 + bidx] = coef_bytes[:, bidx]
        return out.tobytes(), tpos

    body_main, _ = serialise(counts, wires - 1, coef_bytes)
    # adder template
    a_counts, a_w, a_c = [], [], []
    for parts in a_rows:
        for terms in parts:
            a_counts.append(len(terms))
            for v, c in terms:
                a_w.append(v)
                a_c.append(c)
    a_counts = np.asarray(a_counts, dtype=np.uint32)
    a_w = np.asarray(a_w, dtype=np.int64)
    a_cb = np.frombuffer(b"".join(int(c).to_bytes(32, "little") for c in a_c), dtype=np.uint8).reshape(len(a_c), 32)
    tmpl, tpos = serialise(a_counts, np.where(a_w == 1, 0, a_w - 1).astype(np.uint32), a_cb)
    tmpl = np.frombuffer(tmpl, dtype=np.uint8)
    blocks = []
    for off in adders:
        blk = tmpl.copy()
        w = np.where(a_w == 1, 0, a_w + off - 1).astype("<u4").view(np.uint8).reshape(len(a_w), 4)
        for bidx in range(4):
            blk[tpos + bidx] = w[:, bidx]
        blocks.append(blk.tobytes())
    body2 = body_main + b"".join(blocks)
    n_rows = b.n_rows() + len(adders) * len(a_rows)
    n_wires = b.nvars - 1                         # circom 2.0.x convention: max wire id == header nWires
    body1 = struct.pack("<I", 32) + P.to_bytes(32, "little") + struct.pack("<IIII", n_wires, 2 * k, 0, k)
    body1 += struct.pack("<QI", n_wires, n_rows)
    body3 = np.arange(n_wires, dtype="<u8").tobytes()
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(b"r1cs" + struct.pack("<II", 1, 3))
        f.write(struct.pack("<IQ", 2, len(body2)))
        f.write(body2)
        f.write(struct.pack("<IQ", 1, len(body1)))
        f.write(body1)
        f.write(struct.pack("<IQ", 3, len(body3)))
        f.write(body3)
    os.replace(tmp, path)
    return dict(S=S, stride=stride, n_rows=n_rows, n_vars=b.nvars, n_main_rows=n_before_mux,
                n_adders=len(adders), adder_rows=len(a_rows), bytes=os.path.getsize(path))


def cached(S=26, stride=10, directory=None, seed=0):
    """Path of ecdsa_like(S, stride), generated on first use under the fixture scratch directory."""
    import fixtures
    directory = directory or fixtures._CACHE
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, "ecdsa_like_S%d_s%d%s.r1cs" % (S, stride, "" if seed == 0 else "_seed%d" % seed))
    if not os.path.exists(path):
        generate(path, S=S, stride=stride, seed=seed)
    return path


if __name__ == "__main__":
    import sys
    import time
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    t = time.time()
    print(generate("/tmp/ecdsa_like_S%d_s%d.r1cs" % (S, stride), S, stride), "%.1fs" % (time.time() - t))
