"""Synthetic systems whose interesting row is LONG (> 64 terms): the engine pops such rows alone and
runs R1 / R7 / R8 on the whole workgroup (exec_big_row_wg), a different code path from the
lane-per-row executor the fixtures mostly exercise.  Each case names the rule the long row must hit
(index into rule_hits: 0 = R1, 6 = R7, 7 = R8) and whether it must fire.

Variable ids are 1-based: 1 = constant one, then outputs, public inputs, private inputs, internals
(/root/reference/src/ParseR1CS.jl:117-123)."""
import r1cs_py

P = r1cs_py.P


def _binary_row(x, other_root=1):
    # x * (x - other_root) = 0  -> R2 gives x the candidate values {0, other_root}
    return ([(x, 1)], [(x, 1), (1, (-other_root) % P)], [])


def r7_chain(n=100, broken_at=None):
    """5*y = sum 2^i x_i, x_i binary via R2, y a public input.  R7 makes every x_i unique; with
    broken_at=j, x_j is {0,2}-valued (no [0,1] bounds) and the chain breaks at the link after it."""
    y = 2
    xs = list(range(3, 3 + n))
    rows = [_binary_row(x, 2 if i == broken_at else 1) for i, x in enumerate(xs)]
    rows.append(([], [], [(y, 5)] + [(x, (-(1 << i)) % P) for i, x in enumerate(xs)]))
    return dict(nwires=2 + n, nout=0, npub=1, nprv=0, rows=rows)


def r1_sum(n=80):
    """y = sum x_i with every x_i a public input: R1 on a plain long sum."""
    y = 2
    xs = list(range(3, 3 + n))
    rows = [([], [], [(y, (-1) % P)] + [(x, 1 + i) for i, x in enumerate(xs)])]
    return dict(nwires=2 + n, nout=1, npub=n, nprv=0, rows=rows)


def r1_long_a(n=70):
    """z = (sum x_i) * 1: a long row WITH A/B terms (only R1 / R2 apply to it)."""
    z = 2
    xs = list(range(3, 3 + n))
    rows = [([(x, 3 + i) for i, x in enumerate(xs)], [(1, 1)], [(z, 1)])]
    return dict(nwires=2 + n, nout=1, npub=n, nprv=0, rows=rows)


def r8_decoder(n=100, with_success_input=True):
    """circomlib Decoder(n): out_i * (inp - i) = 0 (P4 tags every out_i with inp), sum out_i = success."""
    success, inp = 2, 3
    outs = list(range(4, 4 + n))
    rows = [([(inp, 1), (1, (-i) % P)], [(o, 1)], []) for i, o in enumerate(outs)]
    rows.append(([], [], [(success, (-1) % P)] + [(o, 1) for o in outs]))
    npub = 2 if with_success_input else 1
    if not with_success_input:
        success, inp = 3, 2      # success becomes an internal signal: the group is not closed
        rows = [([(inp, 1), (1, (-i) % P)], [(o, 1)], []) for i, o in enumerate(outs)]
        rows.append(([], [], [(success, (-1) % P)] + [(o, 1) for o in outs]))
    return dict(nwires=3 + n, nout=0, npub=npub, nprv=0, rows=rows)


def hub_fanout(n_rows=8200, n_hubs=10):
    """n_hubs variables that each occur in ALL n_rows rows and all become unique in the same round: their
    REQUEUE lists (n_hubs * n_rows candidates) exceed the engine's candidate buffer, which must then replay
    the events sequentially (resolve_pushes / queue_round_multi fallback)."""
    ins = list(range(2 + n_rows, 2 + n_rows + n_hubs + n_rows))      # public inputs: n_hubs sources, n_rows addends
    zs = list(range(2, 2 + n_rows))                                    # outputs z_j
    src, ws = ins[:n_hubs], ins[n_hubs:]
    first_internal = 2 + n_rows + n_hubs + n_rows
    hubs = list(range(first_internal, first_internal + n_hubs))
    rows = [([], [], [(h, 1), (sv, (-1) % P)]) for h, sv in zip(hubs, src)]          # h_i == source_i
    for j in range(n_rows):
        rows.append(([], [], [(zs[j], (-1) % P), (ws[j], 1)] + [(h, 2 + i) for i, h in enumerate(hubs)]))
    return dict(nwires=first_internal + n_hubs - 1, nout=n_rows, npub=n_hubs + n_rows, nprv=0, rows=rows)


CASES = {
    # name: (spec, rule index, must fire)
    "r7_chain_100": (r7_chain(100), 6, True),
    "r7_chain_250": (r7_chain(250), 6, True),
    "r7_broken_first": (r7_chain(100, broken_at=0), 6, False),
    "r7_broken_mid": (r7_chain(100, broken_at=57), 6, False),
    "r7_broken_last": (r7_chain(100, broken_at=99), 6, False),
    "r1_sum_80": (r1_sum(80), 0, True),
    "r1_sum_600": (r1_sum(600), 0, True),
    "r1_long_a_70": (r1_long_a(70), 0, True),
    "r8_decoder_100": (r8_decoder(100), 7, True),
    "r8_decoder_700": (r8_decoder(700), 7, True),
    "r8_decoder_open": (r8_decoder(100, with_success_input=False), 7, False),
    "hub_fanout": (hub_fanout(), 0, True),
}


def write(path, spec):
    r1cs_py.write(path, spec["nwires"], spec["nout"], spec["npub"], spec["nprv"], spec["rows"])
