"""N renumbered copies of one fixture circuit, concatenated into one .r1cs: a million-row input that is NOT shaped like
ecdsa_like -- many independent medium-length dependency chains side by side, no long sums, no trusted functions.
(tests/tools/scale_variants.py: is the engine's schedule policy fitted to one generator?)

Variable layout of the result (ParseR1CS.jl:123: knowns = [1] ++ inputs, targets = outputs, both contiguous id ranges):
    1 | outputs of copy 0, 1, ... | inputs of copy 0, 1, ... | the other signals of copy 0, 1, ...
"""
import os
import struct

import numpy as np

import r1cs_py

P = r1cs_py.P


def generate(path, rel, copies):
    import fixtures
    hdr, rows = r1cs_py.parse_file(fixtures.path(rel))
    nw, nout = hdr["nWires"], hdr["nPubOut"]
    nin = hdr["nPubIn"] + hdr["nPrvIn"]
    nint = nw - nout - nin                     # variables nout + nin + 2 .. nw + 1
    counts, w, c = [], [], []
    for parts in rows:
        for terms in parts:
            counts.append(len(terms))
            for v, co in terms:
                w.append(v)
                c.append(co)
    counts = np.asarray(counts, dtype=np.uint32)
    w = np.asarray(w, dtype=np.int64)
    cb = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in c), dtype=np.uint8).reshape(len(c), 32)
    nparts, nt = len(counts), len(w)
    terms_before = np.concatenate(([0], np.cumsum(counts)[:-1])).astype(np.int64)
    starts = 4 * np.arange(nparts, dtype=np.int64) + 36 * terms_before
    tmpl = np.zeros(4 * nparts + 36 * nt, dtype=np.uint8)
    cview = counts.astype("<u4").view(np.uint8).reshape(nparts, 4)
    for b in range(4):
        tmpl[starts + b] = cview[:, b]
    part_of_term = np.repeat(np.arange(nparts, dtype=np.int64), counts)
    tpos = starts[part_of_term] + 4 + 36 * (np.arange(nt, dtype=np.int64) - terms_before[part_of_term])
    for b in range(32):
        tmpl[tpos + 4 + b] = cb[:, b]
    is_one, is_out, is_in = w == 1, (w >= 2) & (w <= 1 + nout), (w >= 2 + nout) & (w <= 1 + nout + nin)
    blocks = []
    for k in range(copies):
        nv = np.where(is_one, 1,
             np.where(is_out, w + k * nout,
             np.where(is_in, (w - nout) + copies * nout + k * nin,
                      (w - nout - nin) + copies * (nout + nin) + k * nint)))
        blk = tmpl.copy()
        wv = (nv - 1).astype("<u4").view(np.uint8).reshape(nt, 4)
        for b in range(4):
            blk[tpos + b] = wv[:, b]
        blocks.append(blk.tobytes())
    body2 = b"".join(blocks)
    n_wires = copies * nw
    n_rows = copies * len(rows)
    body1 = struct.pack("<I", 32) + P.to_bytes(32, "little") + struct.pack("<IIII", n_wires, copies * nout, copies * hdr["nPubIn"], copies * hdr["nPrvIn"])
    body1 += struct.pack("<QI", n_wires, n_rows)
    body3 = np.arange(n_wires, dtype="<u8").tobytes()
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(b"r1cs" + struct.pack("<II", 1, 3))
        f.write(struct.pack("<IQ", 2, len(body2)) + body2)
        f.write(struct.pack("<IQ", 1, len(body1)) + body1)
        f.write(struct.pack("<IQ", 3, len(body3)) + body3)
    os.replace(tmp, path)
    return dict(copies=copies, n_rows=n_rows, n_vars=n_wires + 1, bytes=os.path.getsize(path))


def cached(rel, copies, directory=None):
    import fixtures
    directory = directory or fixtures._CACHE
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, "copies_%d_%s" % (copies, os.path.basename(rel)))
    if not os.path.exists(path):
        generate(path, rel, copies)
    return path


def generate_mixed(path, rels, interleave=False, extra_rows=()):
    """Several fixture circuits (a name may repeat) written into ONE file with disjoint variables (only the constant wire is
    shared), their rows one circuit after the other or, interleave=True, dealt out round-robin; extra_rows are appended as they
    are (rows over the constant wire only, say). Pure Python (r1cs_py): for small files. Variable layout as in generate()."""
    import fixtures
    pieces = []
    for rel in rels:
        hdr, rows = r1cs_py.parse_file(fixtures.path(rel))
        pieces.append((hdr["nWires"], hdr["nPubOut"], hdr["nPubIn"], hdr["nPrvIn"], rows))
    tot_out = sum(p[1] for p in pieces)
    tot_in = sum(p[2] + p[3] for p in pieces)
    ob = ib = tb = 0
    mapped = []
    for nw, nout, npub, nprv, rows in pieces:
        nin = npub + nprv

        def m(w, nout=nout, nin=nin, ob=ob, ib=ib, tb=tb):
            if w == 1:
                return 1
            if w <= 1 + nout:
                return w + ob
            if w <= 1 + nout + nin:
                return (w - nout) + tot_out + ib
            return (w - nout - nin) + tot_out + tot_in + tb
        mapped.append([[[(m(v), c) for v, c in part] for part in row] for row in rows])
        ob += nout
        ib += nin
        tb += nw - nout - nin
    out_rows = []
    if interleave:
        k = 0
        while any(k < len(r) for r in mapped):
            for r in mapped:
                if k < len(r):
                    out_rows.append(r[k])
            k += 1
    else:
        for r in mapped:
            out_rows.extend(r)
    out_rows.extend(extra_rows)
    n_wires = sum(p[0] for p in pieces)
    r1cs_py.write(path, n_wires, tot_out, sum(p[2] for p in pieces), sum(p[3] for p in pieces), out_rows)
    return dict(n_rows=len(out_rows), n_vars=n_wires + 1)
