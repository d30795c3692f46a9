"""Tiny pure-Python .r1cs reader used only by tests (fixture statistics, Julia-order KAT,
synthetic-file round trips).  Follows the iden3 v1 layout the same way the reference reads it
(/root/reference/src/ParseR1CS.jl:50-124): sections in any order, 32-byte coefficients."""
import struct

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def parse(data):
    assert struct.unpack_from("<I", data, 4)[0] == 1
    nsec = struct.unpack_from("<I", data, 8)[0]
    cur = 12
    starts = {}
    for _ in range(nsec):
        t, sz = struct.unpack_from("<IQ", data, cur)
        starts[t] = cur + 12
        cur += 12 + sz
    s1 = starts[1]
    fs = struct.unpack_from("<I", data, s1)[0]
    s1 += 4
    prime = int.from_bytes(data[s1:s1 + fs], "little")
    s1 += fs
    nwires, nout, npub, nprv = struct.unpack_from("<IIII", data, s1)
    s1 += 16
    nlabels, ncons = struct.unpack_from("<QI", data, s1)
    s2 = starts[2]
    rows = []
    for _ in range(ncons):
        parts = []
        for _p in range(3):
            n = struct.unpack_from("<I", data, s2)[0]
            s2 += 4
            terms = []
            for _k in range(n):
                w = struct.unpack_from("<I", data, s2)[0]
                s2 += 4
                c = int.from_bytes(data[s2:s2 + 32], "little") % P
                s2 += 32
                terms.append((w + 1, c))
            parts.append(terms)
        rows.append(parts)
    hdr = dict(fieldSize=fs, prime=prime, nWires=nwires, nPubOut=nout, nPubIn=npub, nPrvIn=nprv,
               nLabels=nlabels, nConstraints=ncons)
    return hdr, rows


def parse_file(path):
    import lzma
    with open(path, "rb") as f:
        data = f.read()
    if path.endswith(".xz"):
        data = lzma.decompress(data)
    return parse(data)


def write(path, nwires, nout, npub, nprv, rows, nlabels=None):
    """rows: list of (A, B, C) with each part a list of (var_id_1based, coeff int).  Section order
    [2, 1, 3] like every circom 2.0 file in the reference tree."""
    nlabels = nwires if nlabels is None else nlabels
    body2 = bytearray()
    for parts in rows:
        for terms in parts:
            body2 += struct.pack("<I", len(terms))
            for v, c in terms:
                body2 += struct.pack("<I", v - 1) + (c % P).to_bytes(32, "little")
    body1 = struct.pack("<I", 32) + P.to_bytes(32, "little") + struct.pack("<IIII", nwires, nout, npub, nprv)
    body1 += struct.pack("<QI", nlabels, len(rows))
    body3 = b"".join(struct.pack("<Q", i) for i in range(nlabels))
    out = b"r1cs" + struct.pack("<II", 1, 3)
    out += struct.pack("<IQ", 2, len(body2)) + bytes(body2)
    out += struct.pack("<IQ", 1, len(body1)) + body1
    out += struct.pack("<IQ", 3, len(body3)) + body3
    with open(path, "wb") as f:
        f.write(out)
