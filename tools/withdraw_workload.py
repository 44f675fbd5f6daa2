"""bench.py's workload: the reference's withdraw circuit, laid out row by row (measurement infrastructure, not product and
not oracle: nothing here is imported by zkt-plonk_amd/, and nothing here imports oracle/).

What the reference does to arrive at the prover's inputs: `ProveWithdraw` (bin/src/main.rs:190-293)
assembles a `WithdrawCircuit<F, u64, G, H, INPUTS, HEIGHT>` from the wallet (notes, Merkle paths), `ZKTPlonk::prove`
(plonk-core/src/plonk.rs:94-111) runs `circuit.synthesize` on a proving composer (circuits/src/withdraw.rs:57-150) and
hands the composer to `proof_system::prove`.  This module produces the same three things for a synthetic wallet:

  * the CIRCUIT (selector rows, wire -> variable indices, public-input rows) in CLOSED FORM: every gate's selectors are
    written down directly from the algebra of the gate and the constants of the hasher instead of being folded step by
    step through LTVariable transforms (tests/test_withdraw_workload.py checks the result row for row against the
    oracle's operational restatement of the composer);
  * the host part of the WITNESS: the ~150 variables per note that are not outputs of Poseidon gates (note data, path
    bits and siblings, selects, sums), which need the hash VALUES (native Poseidon, as main.rs:248-271 computes them);
  * the list of hash calls (first variable, input variables) whose vars_per_hash variables each -- 99.9 % of the
    witness -- the device produces (zkt_plonk_amd.PoseidonGadget -> k_poseidon_gadget).

Gate counts: per hash P = Rf (3W + W^2) + Rp (3 + W^2) (x4: 1288, x5: 1888); per note (3 + H) P + 7 H + 4; global
2 P + 130 + INPUTS.  bin/Cargo.toml's default features (height-48, notes-3, x4) give 200 793 gates -> n = 2^18, its largest
(height-64, notes-4, x5) 511 702 -> 2^19; INPUTS = 8, HEIGHT = 64, x5 gives 1 019 498 -> n = 2^20 (BASELINE.json configs[3]);
INPUTS = 1, HEIGHT = 7, x4 gives 15 640 -> 2^14 (configs[0]).  x3 cannot hash a leaf (three inputs, WIDTH 3: FullBuffer).
"""
import json
import os
import random

import numpy as np

ZERO = 0xFFFFFFFF          # ZKT_VARIABLE_ZERO
K1, K2 = 7, 13             # plonk-core/src/permutation/constants.rs:13-20
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# (width, INPUTS, HEIGHT) per log2 of the domain: the shipped feature sets where they exist
SHAPES = {14: (4, 1, 7), 18: (4, 3, 48), 19: (5, 4, 64), 20: (5, 8, 64), 22: (5, 32, 64)}


class Hasher:
    """PoseidonConstants as the gadget and the native hasher read them."""

    def __init__(self, p, width, half_full, partial, rc, mds, domain_tag=None):
        self.p, self.width, self.half_full, self.partial = p, width, half_full, partial
        self.rc, self.mds = list(rc), [list(r) for r in mds]
        self.tag = ((1 << (width - 1)) - 1) % p if domain_tag is None else domain_tag
        self.per_hash = 2 * half_full * (3 * width + width * width) + partial * (3 + width * width)
        self.hash_var_offset = self.per_hash - 1 - (width - 2) * width

    def native(self, inputs):
        p, W = self.p, self.width
        if len(inputs) > W - 1:
            raise ValueError("Poseidon Error: FullBuffer")
        st = [self.tag] + [x % p for x in inputs] + [0] * (W - 1 - len(inputs))
        k = 0
        for r in range(2 * self.half_full + self.partial):
            full = r < self.half_full or r >= self.half_full + self.partial
            for i in range(W):
                st[i] = (st[i] + self.rc[k + i]) % p
                if full or i == 0:
                    st[i] = pow(st[i], 5, p)
            k += W
            st = [sum(self.mds[i][j] * st[i] for i in range(W)) % p for j in range(W)]
        return st[1]


def reference_hasher(p, width):
    """Bn254x{3,4,5} (gadgets/src/poseidon): the committed field elements of tests/golden/poseidon_bn254.npz."""
    arr = np.load(os.path.join(GOLDEN, "poseidon_bn254.npz"))
    with open(os.path.join(GOLDEN, "poseidon_bn254.json")) as f:
        meta = json.load(f)["x%d" % width]
    to_int = lambda a: [sum(int(v) << (64 * i) for i, v in enumerate(row)) for row in a]
    rc, mds = to_int(arr["rc_x%d" % width]), to_int(arr["mds_x%d" % width])
    return Hasher(p, width, meta["full_rounds"] // 2, meta["partial_rounds"], rc, [mds[i * width:(i + 1) * width] for i in range(width)],
                  meta["domain_tag"])


def synthetic_hasher(p, width, full_rounds=8, partial_rounds=56, seed=0x9051D0):
    """For fields the reference ships no table for (BLS12-381: PoseidonConstants::generate() at run time): same round
    numbers as Bn254x5, pseudo-random constants."""
    rnd = random.Random(seed + width)
    rc = [rnd.randrange(p) for _ in range((full_rounds + partial_rounds) * width)]
    mds = [[rnd.randrange(1, p) for _ in range(width)] for _ in range(width)]
    return Hasher(p, width, full_rounds // 2, partial_rounds, rc, mds)


class MerkleTree:
    """The wallet's sparse tree (gadgets/src/merkle_tree.rs:57-111)."""

    def __init__(self, hs, height):
        self.hs, self.height, self.tree, self.nodes, self.next_index, self.root = hs, height, {}, [], 0, 0
        h = 0
        for _ in range(height):
            self.nodes.append(h)
            h = hs.native([h, h])

    def add_leaf(self, h):
        index = self.next_index
        self.next_index += 1
        for layer in range(self.height):
            idx = index >> layer
            self.tree[(layer, idx)] = h
            sib = self.tree.get((layer, idx ^ 1), self.nodes[layer])
            h = self.hs.native([sib, h]) if idx & 1 else self.hs.native([h, sib])
        self.root = h
        return index

    def path(self, index):
        return [self.tree.get((layer, (index >> layer) ^ 1), self.nodes[layer]) for layer in range(self.height)]


def make_instance(hs, inputs, height, seed, ident_set=None):
    """A wallet state and a withdrawal from it: `inputs` notes among 2 * inputs + 3 deposited ones."""
    rnd = random.Random(seed)
    p = hs.p
    if ident_set is None:
        ident_set = [random.Random(0x1D5E7 + k).randrange(1, p) for k in range(7)]
    total = min(2 * inputs + 3, 1 << height)
    tree = MerkleTree(hs, height)
    notes = []
    for k in range(total):
        secret, ident, amount = rnd.randrange(1, p), rnd.choice(ident_set), rnd.randrange(1, 1 << 40)
        idx = tree.add_leaf(hs.native([ident, amount, hs.native([secret])]))
        notes.append((secret, ident, amount, idx))
    spent = rnd.sample(notes, inputs)
    amount_in = sum(nt[2] for nt in spent)
    return dict(secrets=[nt[0] for nt in spent], identifiers=[nt[1] for nt in spent], amounts=[nt[2] for nt in spent],
                poes=[(nt[3], tree.path(nt[3])) for nt in spent], root=tree.root, new_secret=rnd.randrange(1, p),
                new_identifier=rnd.choice(ident_set), withdraw_amount=rnd.randrange(1, amount_in), ident_set=list(ident_set))


class Layout:
    """Rows of the circuit + the host-known variable values (0 where the device writes)."""

    def __init__(self, p):
        self.p = p
        self.q = {k: [] for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "q_lookup")}
        self.w = ([], [], [])
        self.values = []
        self.pi = {}
        self.hash_calls = []           # (first variable, input variables)

    def var(self, value):
        self.values.append(value % self.p)
        return len(self.values) - 1

    def row(self, l, r, o, qm=0, ql=0, qr=0, qo=0, qc=0, lk=0, pi=None):
        q = self.q
        if pi is not None:
            self.pi[len(q["q_m"])] = pi % self.p
        q["q_m"].append(qm); q["q_l"].append(ql); q["q_r"].append(qr); q["q_o"].append(qo); q["q_c"].append(qc)
        q["q_lookup"].append(lk)
        self.w[0].append(l); self.w[1].append(r); self.w[2].append(o)

    @property
    def n_gates(self):
        return len(self.q["q_m"])


def _gadget_template(hs, n_inputs):
    """One hash's rows relative to its first variable: references are ('t', k) = the k-th variable of this hash,
    ('i', k) = its k-th input, or ZERO.  The selectors depend on the constants only, so the template is built once per
    arity and stamped per call."""
    p, W = hs.p, hs.width
    m1 = p - 1
    rows = []
    st = [(ZERO, hs.tag)] + [(("i", k), 0) for k in range(n_inputs)] + [(ZERO, 0)] * (W - 1 - n_inputs)
    t = 0
    k = 0
    for r in range(2 * hs.half_full + hs.partial):
        full = r < hs.half_full or r >= hs.half_full + hs.partial
        st = [(v, (o + hs.rc[k + i]) % p) for i, (v, o) in enumerate(st)]     # add_constant: offsets only
        k += W
        for i in range(W if full else 1):
            v, o = st[i]
            # (v + o)^2 = v v + o v + o v + o^2 ; x^4 = x^2 x^2 ; x^5 = x^4 (v + o) = x^4 v + o x^4
            rows.append((v, v, ("t", t), 1, o, o, m1, o * o % p))
            rows.append((("t", t), ("t", t), ("t", t + 1), 1, 0, 0, m1, 0))
            rows.append((("t", t + 1), v, ("t", t + 2), 1, o, 0, m1, 0))
            st[i] = (("t", t + 2), 0)
            t += 3
        nx = []
        for j in range(W):
            acc = ZERO
            for i in range(W):
                v, o = st[i]
                m = hs.mds[i][j]
                rows.append((acc, v, ("t", t), 0, 1, m, m1, o * m % p))        # acc + m (v + o) = new
                acc = ("t", t)
                t += 1
            nx.append((acc, 0))
        st = nx
    assert t == hs.per_hash and st[1][0] == ("t", hs.hash_var_offset)
    return rows


def _hash(L, hs, templates, in_vars):
    tpl = templates.get(len(in_vars))
    if tpl is None:
        tpl = templates[len(in_vars)] = _gadget_template(hs, len(in_vars))
    base = len(L.values)
    L.values.extend([0] * hs.per_hash)
    L.hash_calls.append((base, tuple(in_vars)))
    res = lambda ref: ref if ref == ZERO else (base + ref[1] if ref[0] == "t" else in_vars[ref[1]])
    q, w = L.q, L.w
    for (l, r, o, qm, ql, qr, qo, qc) in tpl:
        q["q_m"].append(qm); q["q_l"].append(ql); q["q_r"].append(qr); q["q_o"].append(qo); q["q_c"].append(qc)
        q["q_lookup"].append(0)
        w[0].append(res(l)); w[1].append(res(r)); w[2].append(base + o[1])
    return base + hs.hash_var_offset


def layout(hs, inst):
    """WithdrawCircuit::synthesize (circuits/src/withdraw.rs:57-150) as rows."""
    p = hs.p
    m1 = p - 1
    L = Layout(p)
    T = {}
    val = lambda v: L.values[v]
    amount_out = sum(inst["amounts"]) - inst["withdraw_amount"]
    assert amount_out >= 0
    amount_vars = [L.var(a) for a in inst["amounts"]]
    ident_vars = [L.var(i) for i in inst["identifiers"]]
    root_var = L.var(inst["root"])
    L.row(ZERO, ZERO, root_var, qo=m1, pi=inst["root"])                                   # set_variable_public
    for amount_var, ident_var, secret, (leaf_index, path) in zip(amount_vars, ident_vars, inst["secrets"], inst["poes"]):
        secret_var = L.var(secret)
        commitment = hs.native([secret])
        c_var = _hash(L, hs, T, [secret_var])
        inv = pow(secret, -1, p)
        inv_var = L.var(inv)                                                              # div_gate(1, secret): wires (secret, z, Zero)
        L.row(secret_var, inv_var, ZERO, qm=1, qo=m1, qc=m1)                              #   secret z - (0 + 1) = 0
        nullifier = hs.native([inv])
        n_var = _hash(L, hs, T, [inv_var])
        L.row(ZERO, ZERO, n_var, qo=m1, pi=nullifier)
        leaf = hs.native([val(ident_var), val(amount_var), commitment])
        cur_var, cur = _hash(L, hs, T, [ident_var, amount_var, c_var]), leaf
        bits = []
        for layer in range(len(path)):                                                    # PoECircuit::synthesize
            b = L.var((leaf_index >> layer) & 1)
            L.row(b, b, b, qm=1, qo=m1)                                                   # boolean_gate
            bits.append(b)
        sib_vars = [L.var(node) for node in path]
        for layer, (b, s_var) in enumerate(zip(bits, sib_vars)):                          # merkle_proof
            bit, node = val(b), val(s_var)
            pair = []
            for a_var, a_val, o_var, o_val in ((s_var, node, cur_var, cur), (cur_var, cur, s_var, node)):
                x, y = L.var(bit * a_val), L.var((1 - bit) * o_val)                       # conditional_select(bit, a, o)
                z = L.var(val(x) + val(y))
                L.row(b, a_var, x, qm=1, qo=m1)
                L.row(b, o_var, y, qm=m1, qr=1, qo=m1)
                L.row(x, y, z, ql=1, qr=1, qo=m1)
                pair.append(z)
            cur = hs.native([val(pair[0]), val(pair[1])])
            cur_var = _hash(L, hs, T, pair)
        L.row(cur_var, root_var, ZERO, ql=1, qr=m1)                                       # equal_constrain(root, pub_root)
        lk = L.var(val(ident_var))
        L.row(ident_var, ZERO, lk, ql=1, qo=m1, lk=1)                                     # lookup_constrain(identifier)
    vs = []
    for k in range(64):
        b = L.var((amount_out >> k) & 1)
        L.row(b, b, b, qm=1, qo=m1)
        vs.append(b)
    mult = 2
    while len(vs) > 1:                                                                    # bits_le_constrain
        nxt = []
        for k in range(0, len(vs), 2):
            nv = L.var(val(vs[k]) + val(vs[k + 1]) * mult)
            L.row(vs[k], vs[k + 1], nv, ql=1, qr=mult % p, qo=m1)
            nxt.append(nv)
        vs, mult = nxt, mult * mult % (1 << 64)
    out_var = vs[0]
    right = ZERO
    for a in amount_vars[1:]:
        nv = L.var((0 if right == ZERO else val(right)) + val(a))
        L.row(right, a, nv, ql=1, qr=1, qo=m1)
        right = nv
    L.row(amount_vars[0], right, out_var, ql=m1, qr=m1, qo=1, pi=inst["withdraw_amount"])
    ns_var, ni_var = L.var(inst["new_secret"]), L.var(inst["new_identifier"])
    nc = hs.native([inst["new_secret"]])
    nc_var = _hash(L, hs, T, [ns_var])
    new_leaf = hs.native([inst["new_identifier"], amount_out, nc])
    nl_var = _hash(L, hs, T, [ni_var, out_var, nc_var])
    L.row(ZERO, ZERO, ni_var, qo=m1, pi=inst["new_identifier"])
    L.row(ZERO, ZERO, nl_var, qo=m1, pi=new_leaf)
    return L


def sigma_columns(L, n):
    """permutation/mod.rs:104-137: the wires of one variable form a cycle in insertion order.  Returns three (n,) int64
    arrays of encoded targets col * n + row (padding rows map to themselves)."""
    g = L.n_gates
    w = np.stack([np.asarray(c, dtype=np.int64) for c in L.w], axis=1).reshape(-1)         # gate-major: L, R, O
    enc = np.tile(np.arange(3, dtype=np.int64) * n, g) + np.repeat(np.arange(g, dtype=np.int64), 3)
    order = np.argsort(w, kind="stable")
    ws, es = w[order], enc[order]
    nxt = np.roll(es, -1)
    first = np.r_[True, ws[1:] != ws[:-1]]
    last = np.r_[first[1:], True]
    starts = es[first]
    group = np.cumsum(first) - 1
    nxt[last] = starts[group[last]]
    sig = np.arange(3 * n, dtype=np.int64)
    sig[es] = nxt
    return sig[:n], sig[n:2 * n], sig[2 * n:]


def setup_vectors(L, log_n, gen, table_size=1024):
    """The ten evaluation vectors proof_system::setup transforms (setup.rs:62-90), as Python-int lists of length n."""
    p = L.p
    n = 1 << log_n
    assert L.n_gates <= n and table_size < n
    pad = [0] * (n - L.n_gates)
    sel = {k: v + pad for k, v in L.q.items()}
    w = pow(gen, (p - 1) >> log_n, p)
    roots = [1] * n
    for i in range(1, n):
        roots[i] = roots[i - 1] * w % p
    ks = (1, K1, K2)
    kroots = [roots, [K1 * x % p for x in roots], [K2 * x % p for x in roots]]
    for name, col in zip(("sigma1", "sigma2", "sigma3"), sigma_columns(L, n)):
        c, r = np.divmod(col, n)
        c, r = c.tolist(), r.tolist()
        sel[name] = [kroots[ci][ri] for ci, ri in zip(c, r)]
    sel["q_table"] = [0] * table_size + [1] * (n - table_size)
    return sel
