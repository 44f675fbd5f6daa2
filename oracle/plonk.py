"""CPU restatement of the reference's PLONK+Plookup prover hot path
(oracle; test infrastructure only -- never imported by the product).

Every function cites the reference file:line it follows (paths relative to
/root/reference/plonk-core/src).  Field elements are canonical Python ints,
points are ``None`` / ``(x, y)``; polynomials are coefficient lists with trailing zeros
stripped exactly where ``DensePolynomial::from_coefficients_vec`` strips them.

The heavy transforms can be swapped for the C restatement (oracle/coracle.py) through the
``Backend`` object; the default backend is pure Python.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

from . import curve as C
from .curve import Point
from .fields import Curve, K1, K2
from .ntt import Domain, poly_eval, trim
from .transcript import MerlinTranscript

ZERO_VAR = -1  # Variable::Zero (constraint_system/variable.rs:10-15)

# blinder draw order of proof_system/prove.rs:125-127,170-171,225,244,296
BLINDER_LAYOUT = (("a", 2), ("b", 2), ("c", 2), ("h1", 3), ("h2", 2), ("z1", 3), ("z2", 3), ("q", 2))
NUM_BLINDERS = sum(k for _, k in BLINDER_LAYOUT)  # 19


class Backend:
    """Transforms + MSM used by the oracle prover (pure Python by default)."""

    def __init__(self, cv: Curve):
        self.cv = cv
        self._domains: Dict[int, Domain] = {}

    def domain(self, n: int) -> Domain:
        if n not in self._domains:
            self._domains[n] = Domain(self.cv.fr, n)
        return self._domains[n]

    def ifft(self, n, evals):
        return self.domain(n).ifft(evals)

    def fft(self, n, coeffs):
        return self.domain(n).fft(coeffs)

    def coset_fft(self, n, coeffs):
        return self.domain(n).coset_fft(coeffs)

    def coset_ifft(self, n, evals):
        return self.domain(n).coset_ifft(evals)

    def msm(self, bases: Sequence[Point], scalars: Sequence[int]) -> Point:
        return C.msm_pippenger(self.cv, bases, scalars)


# ---------------------------------------------------------------------------------------
# Constraint system (the subset of constraint_system/* needed to build test circuits)
# ---------------------------------------------------------------------------------------
class ConstraintSystem:
    """Setup + Proving composer in one object (constraint_system/composer.rs:121-304,
    mod.rs:56-135).  Variables are indices into ``values``; ZERO_VAR is Variable::Zero."""

    def __init__(self, cv: Curve, table: Sequence[int], table_size: int):
        self.cv = cv
        self.p = cv.fr.p
        self.table_size = table_size
        # lookup/table.rs:19 IndexSet: insertion order, duplicates dropped
        seen, tbl = set(), []
        for t in table:
            t %= self.p
            if t not in seen:
                seen.add(t)
                tbl.append(t)
        self.table = tbl
        self.values: List[int] = []
        self.q_m: List[int] = []
        self.q_l: List[int] = []
        self.q_r: List[int] = []
        self.q_o: List[int] = []
        self.q_c: List[int] = []
        self.q_lookup: List[int] = []
        self.w_l: List[int] = []
        self.w_r: List[int] = []
        self.w_o: List[int] = []
        self.pi: Dict[int, int] = {}  # BTreeMap position -> value (pi.rs:52-105)

    # -- variables ----------------------------------------------------------------------
    def assign_variable(self, value: int) -> int:
        self.values.append(value % self.p)
        return len(self.values) - 1

    def value_of(self, var: int) -> int:
        return 0 if var == ZERO_VAR else self.values[var]

    @property
    def n_gates(self) -> int:
        return len(self.q_m)

    def circuit_bound(self) -> int:
        # constraint_system/mod.rs:96-103
        total = max(self.n_gates, self.table_size)
        return 1 << (total - 1).bit_length() if total > 1 else 1

    # -- gates --------------------------------------------------------------------------
    def arith_constrain(self, w_l, w_r, w_o, q_m=0, q_l=0, q_r=0, q_o=0, q_c=0, q_lookup=0,
                        pi: Optional[int] = None):
        """composer.rs:170-199 gate_constrain + 274-293 input_wires."""
        p = self.p
        if pi is not None:
            self.pi[self.n_gates] = pi % p
        self.q_m.append(q_m % p)
        self.q_l.append(q_l % p)
        self.q_r.append(q_r % p)
        self.q_o.append(q_o % p)
        self.q_c.append(q_c % p)
        self.q_lookup.append(q_lookup % p)
        self.w_l.append(w_l)
        self.w_r.append(w_r)
        self.w_o.append(w_o)

    def add_gate(self, x: int, y: int) -> int:  # constraint_system/arithmetic.rs:15-43
        z = self.assign_variable(self.value_of(x) + self.value_of(y))
        self.arith_constrain(x, y, z, q_l=1, q_r=1, q_o=-1)
        return z

    def mul_gate(self, x: int, y: int) -> int:  # arithmetic.rs:77-104
        z = self.assign_variable(self.value_of(x) * self.value_of(y))
        self.arith_constrain(x, y, z, q_m=1, q_o=-1)
        return z

    def boolean_gate(self, x: int) -> int:  # boolean.rs:26-34
        self.arith_constrain(x, x, x, q_m=1, q_o=-1)
        return x

    def conditional_select(self, bit: int, a: int, b: int) -> int:  # mod.rs:318-373
        bv = self.value_of(bit)
        assert bv in (0, 1)
        xv = bv * self.value_of(a) % self.p
        yv = (1 - bv) * self.value_of(b) % self.p
        x = self.assign_variable(xv)
        y = self.assign_variable(yv)
        z = self.assign_variable(xv + yv)
        self.arith_constrain(bit, a, x, q_m=1, q_o=-1)
        self.arith_constrain(bit, b, y, q_m=-1, q_r=1, q_o=-1)
        self.arith_constrain(x, y, z, q_l=1, q_r=1, q_o=-1)
        return z

    def set_variable_public(self, x: int):  # mod.rs:245-270
        self.arith_constrain(ZERO_VAR, ZERO_VAR, x, q_o=-1, pi=self.value_of(x))

    def lookup_constrain(self, x: int):  # mod.rs:140-160
        w_o = self.assign_variable(self.value_of(x))
        self.arith_constrain(x, ZERO_VAR, w_o, q_l=1, q_o=-1, q_lookup=1)

    # -- derived data ---------------------------------------------------------------------
    def wire_evals(self, n: int):
        """prove.rs:39-55 pad_to + wire_evals."""
        pad = n - self.n_gates
        a = [self.value_of(v) for v in self.w_l] + [0] * pad
        b = [self.value_of(v) for v in self.w_r] + [0] * pad
        c = [self.value_of(v) for v in self.w_o] + [0] * pad
        return a, b, c

    def sigma_mappings(self, n: int):
        """permutation/mod.rs:104-137 compute_sigma_permutations: wires of one variable form
        a cycle in insertion order (Left, Right, Output per gate)."""
        sig = [[(0, i) for i in range(n)], [(1, i) for i in range(n)], [(2, i) for i in range(n)]]
        var_wires: Dict[int, List[Tuple[int, int]]] = {}
        for g in range(self.n_gates):
            for col, var in ((0, self.w_l[g]), (1, self.w_r[g]), (2, self.w_o[g])):
                var_wires.setdefault(var, []).append((col, g))
        for wires in var_wires.values():
            for k, (col, g) in enumerate(wires):
                sig[col][g] = wires[(k + 1) % len(wires)]
        return sig

    def check_satisfied(self) -> bool:
        """constraint_system/helper.rs check_gate, all rows."""
        p = self.p
        tbl = set(self.table) | {0}
        for g in range(self.n_gates):
            a, b, c = self.value_of(self.w_l[g]), self.value_of(self.w_r[g]), self.value_of(self.w_o[g])
            v = (self.q_m[g] * a * b + self.q_l[g] * a + self.q_r[g] * b + self.q_o[g] * c
                 + self.q_c[g] + self.pi.get(g, 0)) % p
            if v != 0:
                return False
            if self.q_lookup[g] and c not in tbl:
                return False
        return True


def test_circuit(cv: Curve, a=2, b=3, d=10, e=True, size=100) -> ConstraintSystem:
    """plonk.rs:144-180 TestCircuit (a + b = c, d = a*c public, select public, c in table)
    with the table of plonk.rs:205 ({1, 2, 5})."""
    cs = ConstraintSystem(cv, [1, 2, 5], size)
    va = cs.assign_variable(a)
    vb = cs.assign_variable(b)
    vc = cs.add_gate(va, vb)
    cs.arith_constrain(va, vc, ZERO_VAR, q_m=-1, pi=d)
    ve = cs.boolean_gate(cs.assign_variable(1 if e else 0))
    vf = cs.conditional_select(ve, va, vb)
    cs.set_variable_public(vf)
    cs.lookup_constrain(vc)
    return cs


def synthetic_circuit(cv: Curve, n_gates: int, table_size: int, seed: int = 1,
                      n_public: int = 7, lookup_every: int = 16, value_seed: Optional[int] = None) -> ConstraintSystem:
    """Withdraw-shaped synthetic trace: random satisfying add/mul/linear
    gates chained through copy constraints, a lookup row every ``lookup_every`` gates,
    ``n_public`` public inputs.  Witness synthesis is out of scope; only the row count and the
    constraint mix matter to the prover.  With ``value_seed`` the witness (free values, table contents, looked-up
    entries, hence the public inputs) is drawn from its own generator: circuits of one ``seed`` then share their
    structure (selectors, copy constraints) and differ in everything a prover is handed per proof."""
    import random
    rnd = random.Random(seed)
    vrnd = rnd if value_seed is None else random.Random(value_seed)
    p = cv.fr.p
    table = [vrnd.randrange(p) for _ in range(min(table_size, 64))]
    cs = ConstraintSystem(cv, table, table_size)
    tbl = cs.table
    live = [cs.assign_variable(vrnd.randrange(p)) for _ in range(4)]
    while cs.n_gates < n_gates - n_public:
        g = cs.n_gates
        x, y = rnd.choice(live), rnd.choice(live)
        if lookup_every and g % lookup_every == lookup_every - 1:
            t = cs.assign_variable(tbl[vrnd.randrange(len(tbl))])
            cs.lookup_constrain(t)
            live.append(t)
        elif g % 3 == 0:
            live.append(cs.mul_gate(x, y))
        elif g % 3 == 1:
            live.append(cs.add_gate(x, y))
        else:
            ql, qr, qc = rnd.randrange(p), rnd.randrange(p), rnd.randrange(p)
            z = cs.assign_variable(ql * cs.value_of(x) + qr * cs.value_of(y) + qc)
            cs.arith_constrain(x, y, z, q_l=ql, q_r=qr, q_o=-1, q_c=qc)
            live.append(z)
        if len(live) > 64:
            live = live[-64:]
    for _ in range(n_public):
        cs.set_variable_public(rnd.choice(live))
    return cs


# ---------------------------------------------------------------------------------------
# Keys
# ---------------------------------------------------------------------------------------
PK_POLYS = ("q_m", "q_l", "q_r", "q_o", "q_c", "sigma1", "sigma2", "sigma3", "q_lookup", "q_table")
EPK_COSETS = ("q_m", "q_l", "q_r", "q_o", "q_c", "q_lookup", "q_table", "sigma1", "sigma2", "sigma3",
              "x", "zh", "l_1")


@dataclass
class ProverKey:  # proof_system/keys/mod.rs:29-77 (coefficient form, trailing zeros stripped)
    n: int
    polys: Dict[str, List[int]]


@dataclass
class ExtendedProverKey:  # proof_system/keys/mod.rs:153-174
    n: int
    cosets: Dict[str, List[int]]  # 13 vectors of 4n coset evaluations
    sigma1: List[int]             # n evaluations
    sigma2: List[int]
    sigma3: List[int]
    q_lookup: List[int]


@dataclass
class VerifierKey:  # proof_system/keys/mod.rs:180-201
    n: int
    pi_roots: List[int]
    commits: Dict[str, Point]


def extend_prover_key(be: Backend, pk: ProverKey, sigma1, sigma2, sigma3, q_lookup) -> ExtendedProverKey:
    """proof_system/keys/mod.rs:78-146."""
    n = pk.n
    p = be.cv.fr.p
    cos = {k: be.coset_fft(4 * n, pk.polys[k]) for k in
           ("q_m", "q_l", "q_r", "q_o", "q_c", "q_lookup", "q_table", "sigma1", "sigma2", "sigma3")}
    cos["x"] = be.coset_fft(4 * n, [0, 1])                       # mod.rs:110-113
    cos["zh"] = be.coset_fft(4 * n, [p - 1] + [0] * (n - 1) + [1])  # mod.rs:115-117 (x^n - 1)
    l1 = trim(be.ifft(n, [1] + [0] * (n - 1)))                   # util.rs:198-206
    cos["l_1"] = be.coset_fft(4 * n, l1)                         # mod.rs:119-120
    return ExtendedProverKey(n, cos, list(sigma1), list(sigma2), list(sigma3), list(q_lookup))


def commit(be: Backend, srs: Sequence[Point], poly: Sequence[int]) -> Point:
    """SonicKZG10::commit with no degree bound / hiding -> kzg10::commit = MSM(powers_of_g, coeffs)
    (commitment.rs:24; call sites prove.rs:133-135 etc.)."""
    if len(poly) > len(srs):
        raise ValueError("TooManyCoefficients")  # kzg10 check_degree_is_too_large
    return be.msm(srs[:len(poly)], list(poly))


def setup_evals(be: Backend, cs: ConstraintSystem) -> Dict[str, List[int]]:
    """The ten evaluation vectors setup.rs:62-90 transforms: padded selectors (setup.rs:28-35), sigma evaluations
    (permutation/mod.rs:139-156) and the table mask (lookup/table.rs:42-48)."""
    cv = be.cv
    p = cv.fr.p
    n = cs.circuit_bound()
    dom = be.domain(n)
    pad = n - cs.n_gates
    sel = {k: getattr(cs, k) + [0] * pad for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "q_lookup")}
    roots = dom.elements()
    ks = (1, K1, K2)
    sig = cs.sigma_mappings(n)
    sig_evals = [[ks[col] * roots[idx] % p for (col, idx) in sig[j]] for j in range(3)]  # permutation/mod.rs:139-156
    assert n > cs.table_size, "max table size is equal or larger than n"             # lookup/table.rs:43
    q_table = [0] * cs.table_size + [1] * (n - cs.table_size)                          # table.rs:42-48
    evals = dict(sel)
    evals.update(sigma1=sig_evals[0], sigma2=sig_evals[1], sigma3=sig_evals[2], q_table=q_table)
    return evals


def setup(be: Backend, srs: Sequence[Point], cs: ConstraintSystem, extend: bool = True):
    """proof_system/setup.rs:42-166."""
    n = cs.circuit_bound()
    roots = be.domain(n).elements()
    evals = setup_evals(be, cs)
    sel = evals
    sig_evals = [evals["sigma1"], evals["sigma2"], evals["sigma3"]]
    polys = {k: trim(be.ifft(n, evals[k])) for k in PK_POLYS}
    commits = {k: commit(be, srs, polys[k]) for k in PK_POLYS}
    pi_roots = [roots[i] for i in sorted(cs.pi.keys())]
    vk = VerifierKey(n, pi_roots, commits)
    pk = ProverKey(n, polys)
    epk = extend_prover_key(be, pk, sig_evals[0], sig_evals[1], sig_evals[2], sel["q_lookup"]) if extend else None
    return pk, epk, vk


def seed_transcript(vk: VerifierKey, tr) -> None:
    """proof_system/keys/mod.rs:260-275."""
    tr.append_u64("circuit_size", vk.n)
    for k in PK_POLYS:
        tr.append_commitment(k + "_commit", vk.commits[k])


# ---------------------------------------------------------------------------------------
# Prover pieces
# ---------------------------------------------------------------------------------------
def add_blinders_to_poly(p: int, poly: List[int], blinders: Sequence[int]) -> List[int]:
    """prove.rs:472-483: append the blinders after the (trimmed) coefficients and subtract them
    from the first k coefficients, i.e. + b(X) * (X^len - 1)."""
    out = list(poly) + [b % p for b in blinders]
    for i, b in enumerate(blinders):
        out[i] = (out[i] - b) % p
    return out


def combine_split(t: Sequence[int], f: Sequence[int]):
    """lookup/multiset.rs:103-146 (IndexMap keeps first-insertion order = order in t)."""
    counters: Dict[int, int] = {}
    for e in t:
        counters[e] = counters.get(e, 0) + 1
    for e in f:
        if e not in counters:
            raise KeyError("ElementNotIndexedInTable")
        counters[e] += 1
    evens, odds, parity = [], [], False
    for elem, count in counters.items():
        half = count // 2
        evens.extend([elem] * half)
        odds.extend([elem] * half)
        if count % 2 == 1:
            if parity:
                odds.append(elem)
                parity = False
            else:
                evens.append(elem)
                parity = True
    return evens, odds


def _batch_or_single_inverse(f, x):
    return f.inv(x)


def compute_z1_evals(cv: Curve, dom: Domain, beta, gamma, a, b, c, s1, s2, s3) -> List[int]:
    """permutation/mod.rs:181-254 (evaluation form, before the iFFT at :256)."""
    p = cv.fr.p
    n = dom.size
    roots = dom.elements()
    z = [1]
    state = 1
    for i in range(n - 1):
        br = beta * roots[i] % p
        num = (br + a[i] + gamma) * (K1 * br + b[i] + gamma) % p * (K2 * br + c[i] + gamma) % p
        den = (beta * s1[i] + a[i] + gamma) * (beta * s2[i] + b[i] + gamma) % p * (beta * s3[i] + c[i] + gamma) % p
        state = state * num % p * cv.fr.inv(den) % p
        z.append(state)
    return z


def compute_z2_evals(cv: Curve, dom: Domain, delta, epsilon, f, t, h1, h2) -> List[int]:
    """lookup/mod.rs:94-151 (evaluation form, before the iFFT at :153)."""
    p = cv.fr.p
    n = dom.size
    opd = (1 + delta) % p
    eopd = epsilon * opd % p
    z = [1]
    state = 1
    for i in range(n - 1):
        num = opd * (epsilon + f[i]) % p * (delta * t[i + 1] + eopd + t[i]) % p
        den = (delta * h2[i] + eopd + h1[i]) * (delta * h1[i + 1] + eopd + h2[i]) % p
        state = state * num % p * cv.fr.inv(den) % p
        z.append(state)
    return z


def quotient_evals(cv: Curve, n: int, epk: ExtendedProverKey, ch, cosets) -> List[int]:
    """proof_system/quotient_poly.rs:98-224 with keys/arithmetic.rs:67-81,
    keys/permutation.rs:97-137, keys/lookup.rs:81-122.  ``cosets`` maps name -> 4n evaluations;
    'next' = index + 4 (wrapping), quotient_poly.rs:52-94."""
    p = cv.fr.p
    alpha, beta, gamma, delta, epsilon = ch
    N = 4 * n
    E = epk.cosets
    a_, b_, c_, pi_ = cosets["a"], cosets["b"], cosets["c"], cosets["pi"]
    z1, z2, t_, h1, h2 = cosets["z1"], cosets["z2"], cosets["t"], cosets["h1"], cosets["h2"]
    alpha2 = alpha * alpha % p
    alpha3 = alpha2 * alpha % p
    alpha4 = alpha3 * alpha % p
    alpha5 = alpha4 * alpha % p
    opd = (delta + 1) % p
    eopd = epsilon * opd % p
    out = []
    for i in range(N):
        j = (i + 4) % N
        a, b, c = a_[i], b_[i], c_[i]
        arith = (a * b % p * E["q_m"][i] + a * E["q_l"][i] + b * E["q_r"][i] + c * E["q_o"][i]
                 + E["q_c"][i] + pi_[i]) % p
        bx = beta * E["x"][i] % p
        perm1 = alpha * z1[i] % p * (bx + a + gamma) % p * (bx * K1 + b + gamma) % p * (bx * K2 + c + gamma) % p
        perm2 = (-alpha) * z1[j] % p * (beta * E["sigma1"][i] + a + gamma) % p \
            * (beta * E["sigma2"][i] + b + gamma) % p * (beta * E["sigma3"][i] + c + gamma) % p
        perm3 = (z1[i] - 1) * E["l_1"][i] % p * alpha2 % p
        lk1 = alpha3 * z2[i] % p * opd % p * (epsilon + E["q_lookup"][i] * c) % p * (eopd + t_[i] + delta * t_[j]) % p
        lk2 = (-alpha3) * z2[j] % p * (eopd + h1[i] + delta * h2[i]) % p * (eopd + h2[i] + delta * h1[j]) % p
        lk3 = alpha4 * (z2[i] - 1) % p * E["l_1"][i] % p
        lk4 = alpha5 * E["q_table"][i] % p * t_[i] % p
        tot = (arith + perm1 + perm2 + perm3 + lk1 + lk2 + lk3 + lk4) % p
        out.append(tot * cv.fr.inv(E["zh"][i]) % p)  # quotient_poly.rs:220-224
    return out


def poly_scale(p, poly, s):
    s %= p
    return [] if s == 0 else [c * s % p for c in poly]


def poly_add(p, *polys):
    n = max((len(q) for q in polys), default=0)
    out = [0] * n
    for q in polys:
        for i, c in enumerate(q):
            out[i] = (out[i] + c) % p
    return trim(out)


def lagrange_evaluation(cv: Curve, n: int, point: int, zh_eval: int, tau: int) -> int:
    """util.rs:185-195 compute_lagrange_evaluation."""
    p = cv.fr.p
    return zh_eval * point % p * cv.fr.inv(n % p * (tau - point) % p) % p


@dataclass
class ProofEvaluations:  # proof_system/proof.rs:30-92 (serialisation order)
    a: int = 0
    b: int = 0
    c: int = 0
    sigma1: int = 0
    sigma2: int = 0
    z1_next: int = 0
    q_lookup: int = 0
    t: int = 0
    t_next: int = 0
    z2_next: int = 0
    h1_next: int = 0
    h2: int = 0

    ORDER = ("a", "b", "c", "sigma1", "sigma2", "z1_next", "q_lookup", "t", "t_next", "z2_next",
             "h1_next", "h2")

    def as_list(self):
        return [getattr(self, k) for k in self.ORDER]


@dataclass
class Proof:  # proof_system/proof.rs:106-155
    commits: Dict[str, Point] = field(default_factory=dict)
    aw_opening: Point = None
    saw_opening: Point = None
    evaluations: ProofEvaluations = field(default_factory=ProofEvaluations)

    COMMIT_ORDER = ("a", "b", "c", "t", "h1", "h2", "z1", "z2", "q_lo", "q_mid", "q_hi")

    def serialize(self, cv: Curve) -> bytes:
        """CanonicalSerialize of Proof<F, D, KZG10<E>>: 11 compressed G1, 2 x (compressed G1 ||
        Option<Fr>::None = 0x00), 12 Fr little-endian canonical."""
        out = bytearray()
        for k in self.COMMIT_ORDER:
            out += C.point_serialize_compressed(cv, self.commits[k])
        for w in (self.aw_opening, self.saw_opening):
            out += C.point_serialize_compressed(cv, w) + b"\x00"
        for v in self.evaluations.as_list():
            out += int(v).to_bytes(cv.fr.limbs64 * 8, "little")
        return bytes(out)


def proof_deserialize(cv: Curve, data: bytes) -> Proof:
    """Inverse of Proof.serialize (proof_system/proof.rs:98-155): 11 compressed G1, 2 x (compressed G1 || 0x00),
    12 Fr little-endian."""
    nb = (cv.fq.bits + 2 + 7) // 8
    fb = cv.fr.limbs64 * 8
    assert len(data) == 13 * nb + 2 + 12 * fb, "proof length"
    pos = 0
    commits = {}
    for k in Proof.COMMIT_ORDER:
        commits[k] = C.point_deserialize_compressed(cv, data[pos:pos + nb])
        pos += nb
    opens = []
    for _ in range(2):
        opens.append(C.point_deserialize_compressed(cv, data[pos:pos + nb]))
        pos += nb
        assert data[pos] == 0, "kzg10::Proof::random_v must be None"
        pos += 1
    ev = ProofEvaluations()
    for k in ProofEvaluations.ORDER:
        v = int.from_bytes(data[pos:pos + fb], "little")
        assert v < cv.fr.p, "non-canonical evaluation"
        setattr(ev, k, v)
        pos += fb
    return Proof(commits, opens[0], opens[1], ev)


def linearization(cv: Curve, dom: Domain, pk: ProverKey, ch, xi, polys):
    """proof_system/linearization_poly.rs:19-121 with keys/arithmetic.rs:37-46,
    keys/permutation.rs:34-69, keys/lookup.rs:29-65."""
    p = cv.fr.p
    f = cv.fr
    alpha, beta, gamma, delta, epsilon = ch
    n = dom.size
    shifted_xi = xi * dom.group_gen % p
    zh_eval = dom.evaluate_vanishing_polynomial(xi)
    l1 = lagrange_evaluation(cv, n, 1, zh_eval, xi)
    P = pk.polys
    ev = ProofEvaluations(
        a=poly_eval(f, polys["a"], xi), b=poly_eval(f, polys["b"], xi), c=poly_eval(f, polys["c"], xi),
        sigma1=poly_eval(f, P["sigma1"], xi), sigma2=poly_eval(f, P["sigma2"], xi),
        z1_next=poly_eval(f, polys["z1"], shifted_xi),
        q_lookup=poly_eval(f, P["q_lookup"], xi), t=poly_eval(f, polys["t"], xi),
        t_next=poly_eval(f, polys["t"], shifted_xi), z2_next=poly_eval(f, polys["z2"], shifted_xi),
        h1_next=poly_eval(f, polys["h1"], shifted_xi), h2=poly_eval(f, polys["h2"], xi))
    arith = poly_add(p, poly_scale(p, P["q_m"], ev.a * ev.b), poly_scale(p, P["q_l"], ev.a),
                     poly_scale(p, P["q_r"], ev.b), poly_scale(p, P["q_o"], ev.c), P["q_c"])
    bxi = beta * xi % p
    alpha2 = alpha * alpha % p
    s_z1 = (alpha * (bxi + ev.a + gamma) % p * (bxi * K1 + ev.b + gamma) % p * (bxi * K2 + ev.c + gamma)
            + l1 * alpha2) % p
    s_s3 = (-alpha) * beta % p * ev.z1_next % p * (beta * ev.sigma1 + ev.a + gamma) % p \
        * (beta * ev.sigma2 + ev.b + gamma) % p
    perm = poly_add(p, poly_scale(p, polys["z1"], s_z1), poly_scale(p, P["sigma3"], s_s3))
    alpha3 = alpha2 * alpha % p
    alpha4 = alpha3 * alpha % p
    opd = (delta + 1) % p
    eopd = epsilon * opd % p
    s_z2 = (alpha3 * opd % p * (epsilon + ev.q_lookup * ev.c) % p * (eopd + ev.t + delta * ev.t_next)
            + alpha4 * l1) % p
    s_h1 = (-alpha3) * ev.z2_next % p * (eopd + ev.h2 + delta * ev.h1_next) % p
    s_qt = alpha4 * alpha % p * ev.t % p
    lookup = poly_add(p, poly_scale(p, polys["z2"], s_z2), poly_scale(p, polys["h1"], s_h1),
                      poly_scale(p, P["q_table"], s_qt))
    xn2 = (zh_eval + 1) * xi % p * xi % p  # xi^(n+2), linearization_poly.rs:103
    qt = poly_add(p, poly_scale(p, poly_add(p, poly_scale(p, polys["q_hi"], xn2), polys["q_mid"]), xn2),
                  polys["q_lo"])
    qt = poly_scale(p, qt, -zh_eval)
    return poly_add(p, arith, perm, lookup, qt), ev


def kzg_open(be: Backend, srs, polys: Sequence[Sequence[int]], point: int, eta: int) -> Point:
    """SonicKZG10::open with opening challenge eta (challenge k = eta^k, k from 0) ->
    kzg10::open: witness = combined / (X - point) (remainder dropped), W = MSM(witness)
    (call sites prove.rs:381-420, 427-451)."""
    p = be.cv.fr.p
    comb: List[int] = []
    ch = 1
    for poly in polys:
        comb = poly_add(p, comb, poly_scale(p, poly, ch))
        ch = ch * eta % p
    if len(comb) < 2:
        return None
    q = [0] * (len(comb) - 1)
    carry = 0
    for i in range(len(comb) - 1, 0, -1):
        carry = (comb[i] + point * carry) % p
        q[i - 1] = carry
    return commit(be, srs, trim(q))


@dataclass
class ProverTrace:
    """Intermediate values exposed for per-kernel parity tests."""
    challenges: Dict[str, int] = field(default_factory=dict)
    polys: Dict[str, List[int]] = field(default_factory=dict)
    evals: Dict[str, List[int]] = field(default_factory=dict)


def prove(be: Backend, srs: Sequence[Point], pk: ProverKey, epk: Optional[ExtendedProverKey],
          vk: VerifierKey, cs: ConstraintSystem, transcript, blinders: Sequence[int],
          trace: Optional[ProverTrace] = None) -> Proof:
    """proof_system/prove.rs:59-470.  ``blinders`` are the 19 F::rand draws in reference order
    (BLINDER_LAYOUT); ``transcript`` must already be seeded (plonk.rs:105-108)."""
    cv = be.cv
    p = cv.fr.p
    n = cs.circuit_bound()
    assert n == pk.n
    dom = be.domain(n)
    assert len(blinders) == NUM_BLINDERS
    bl = {}
    off = 0
    for name, k in BLINDER_LAYOUT:
        bl[name] = [x % p for x in blinders[off:off + k]]
        off += k
    tr = trace if trace is not None else ProverTrace()

    if epk is None:  # prove.rs:88-103
        s1 = be.fft(n, pk.polys["sigma1"])
        s2 = be.fft(n, pk.polys["sigma2"])
        s3 = be.fft(n, pk.polys["sigma3"])
        ql = be.fft(n, pk.polys["q_lookup"])
        epk = extend_prover_key(be, pk, s1, s2, s3, ql)

    pi_vals = [cs.pi[k] for k in sorted(cs.pi.keys())]
    transcript.append_scalars("pi", pi_vals)  # prove.rs:110

    # round 1 (prove.rs:116-140)
    a_ev, b_ev, c_ev = cs.wire_evals(n)
    a_poly = add_blinders_to_poly(p, trim(be.ifft(n, a_ev)), bl["a"])
    b_poly = add_blinders_to_poly(p, trim(be.ifft(n, b_ev)), bl["b"])
    c_poly = add_blinders_to_poly(p, trim(be.ifft(n, c_ev)), bl["c"])
    com = {"a": commit(be, srs, a_poly), "b": commit(be, srs, b_poly), "c": commit(be, srs, c_poly)}
    for k in ("a", "b", "c"):
        transcript.append_commitment(k + "_commit", com[k])

    # round 2 (prove.rs:145-185)
    assert n > cs.table_size and len(cs.table) <= cs.table_size  # lookup/table.rs:52-61
    t_ev = list(cs.table) + [0] * (n - len(cs.table))
    t_poly = trim(be.ifft(n, t_ev))
    f_ev = [ql * c % p for ql, c in zip(epk.q_lookup, c_ev)]  # prove.rs:157-161
    h1_ev, h2_ev = combine_split(t_ev, f_ev)                     # prove.rs:163
    h1_poly = add_blinders_to_poly(p, trim(be.ifft(n, h1_ev)), bl["h1"])
    h2_poly = add_blinders_to_poly(p, trim(be.ifft(n, h2_ev)), bl["h2"])
    com.update(t=commit(be, srs, t_poly), h1=commit(be, srs, h1_poly), h2=commit(be, srs, h2_poly))
    for k in ("t", "h1", "h2"):
        transcript.append_commitment(k + "_commit", com[k])

    # round 3 (prove.rs:190-255)
    beta = transcript.challenge_scalar("beta")
    gamma = transcript.challenge_scalar("gamma")
    delta = transcript.challenge_scalar("delta")
    epsilon = transcript.challenge_scalar("epsilon")
    assert len({beta, gamma, delta, epsilon}) == 4, "challenges must be different"
    z1_ev = compute_z1_evals(cv, dom, beta, gamma, a_ev, b_ev, c_ev, epk.sigma1, epk.sigma2, epk.sigma3)
    z1_poly = add_blinders_to_poly(p, trim(be.ifft(n, z1_ev)), bl["z1"])
    z2_ev = compute_z2_evals(cv, dom, delta, epsilon, f_ev, t_ev, h1_ev, h2_ev)
    z2_poly = add_blinders_to_poly(p, trim(be.ifft(n, z2_ev)), bl["z2"])
    com.update(z1=commit(be, srs, z1_poly), z2=commit(be, srs, z2_poly))
    transcript.append_commitment("z1_commit", com["z1"])
    transcript.append_commitment("z2_commit", com["z2"])

    # round 4 (prove.rs:258-313)
    pi_ev = [0] * n
    for pos, v in cs.pi.items():
        pi_ev[pos] = v
    pi_poly = trim(be.ifft(n, pi_ev))
    alpha = transcript.challenge_scalar("alpha")
    ch = (alpha, beta, gamma, delta, epsilon)
    assert n >= 5  # quotient_poly.rs:44
    src = dict(z1=z1_poly, z2=z2_poly, a=a_poly, b=b_poly, c=c_poly, pi=pi_poly, t=t_poly, h1=h1_poly, h2=h2_poly)
    cosets = {k: be.coset_fft(4 * n, v) for k, v in src.items()}
    q_ev = quotient_evals(cv, n, epk, ch, cosets)
    q_poly = trim(be.coset_ifft(4 * n, q_ev))
    if len(q_poly) < 2 * (n + 2):
        raise IndexError("quotient polynomial too short to split (prove.rs:287-292 would panic)")
    q_lo = trim(q_poly[:n + 2])
    q_mid = trim(q_poly[n + 2:2 * (n + 2)])
    q_hi = trim(q_poly[2 * (n + 2):])
    b0, b1 = bl["q"]
    q_lo = q_lo + [b0]                                   # prove.rs:297
    if not q_mid or not q_hi:
        raise IndexError("empty quotient chunk (prove.rs:298/300 would panic)")
    q_mid[0] = (q_mid[0] - b0) % p                       # prove.rs:298
    q_mid = q_mid + [b1]                                 # prove.rs:299
    q_hi[0] = (q_hi[0] - b1) % p                         # prove.rs:300
    com.update(q_lo=commit(be, srs, q_lo), q_mid=commit(be, srs, q_mid), q_hi=commit(be, srs, q_hi))
    for k in ("q_lo", "q_mid", "q_hi"):
        transcript.append_commitment(k + "_commit", com[k])

    # round 5 (prove.rs:318-451)
    xi = transcript.challenge_scalar("xi")
    polys = dict(a=a_poly, b=b_poly, c=c_poly, z1=z1_poly, z2=z2_poly, h1=h1_poly, h2=h2_poly, t=t_poly,
                 q_lo=q_lo, q_mid=q_mid, q_hi=q_hi)
    r_poly, ev = linearization(cv, dom, pk, ch, xi, polys)
    for name, key in (("a_eval", "a"), ("b_eval", "b"), ("c_eval", "c"), ("sigma1_eval", "sigma1"),
                      ("sigma2_eval", "sigma2"), ("z1_next_eval", "z1_next"), ("q_lookup_eval", "q_lookup"),
                      ("t_eval", "t"), ("t_next_eval", "t_next"), ("z2_next_eval", "z2_next"),
                      ("h1_next_eval", "h1_next"), ("h2_eval", "h2")):
        transcript.append_scalar(name, getattr(ev, key))
    eta = transcript.challenge_scalar("eta")
    # prove.rs:372-375: r is committed (the commitment only feeds PC::open's unused argument)
    aw = kzg_open(be, srs, [r_poly, a_poly, b_poly, c_poly, pk.polys["sigma1"], pk.polys["sigma2"],
                            pk.polys["q_lookup"], t_poly, h2_poly], xi, eta)
    saw = kzg_open(be, srs, [z1_poly, z2_poly, t_poly, h1_poly], xi * dom.group_gen % p, eta)

    tr.challenges.update(alpha=alpha, beta=beta, gamma=gamma, delta=delta, epsilon=epsilon, xi=xi, eta=eta)
    tr.polys.update(polys, r=r_poly, pi=pi_poly, q=q_poly)
    tr.evals.update(a=a_ev, b=b_ev, c=c_ev, t=t_ev, f=f_ev, h1=h1_ev, h2=h2_ev, z1=z1_ev, z2=z2_ev, q=q_ev)
    return Proof(commits={k: com[k] for k in Proof.COMMIT_ORDER}, aw_opening=aw, saw_opening=saw, evaluations=ev)


# ---------------------------------------------------------------------------------------
# Verifier (acceptance oracle).  The pairing check of SonicKZG10::check is replaced by the
# equivalent G1 identity with the known test trapdoor tau:  C - v*G == (tau - z) * W.
# ---------------------------------------------------------------------------------------
def kzg_check_with_trapdoor(cv: Curve, tau: int, commits: Sequence[Point], point: int, values: Sequence[int],
                            w: Point, eta: int) -> bool:
    p = cv.fr.p
    comb_c: Point = None
    comb_v = 0
    ch = 1
    for cm, v in zip(commits, values):
        comb_c = C.add(cv, comb_c, C.scalar_mul(cv, ch, cm))
        comb_v = (comb_v + ch * v) % p
        ch = ch * eta % p
    lhs = C.add(cv, comb_c, C.neg(cv, C.scalar_mul(cv, comb_v, C.generator(cv))))
    rhs = C.scalar_mul(cv, (tau - point) % p, w)
    return lhs == rhs


def verify_prepare(cv: Curve, vk: VerifierKey, proof: Proof, transcript, pub_inputs: Sequence[int]):
    """proof_system/proof.rs:285-503 up to (not including) the two pairing checks of SonicKZG10::check: returns
    [(L1, W1), (L2, W2)] with L = sum_i eta^i C_i - (sum_i eta^i v_i) G + z W, so that the KZG check of each opening is
    e(L, h) == e(W, beta h)  (compute_r0 163-217, compute_linearization_commitment 220-282)."""
    p = cv.fr.p
    dom = Domain(cv.fr, vk.n)
    assert len(pub_inputs) == len(vk.pi_roots)
    transcript.append_scalars("pi", pub_inputs)
    cm = proof.commits
    for k in ("a", "b", "c", "t", "h1", "h2"):
        transcript.append_commitment(k + "_commit", cm[k])
    beta = transcript.challenge_scalar("beta")
    gamma = transcript.challenge_scalar("gamma")
    delta = transcript.challenge_scalar("delta")
    epsilon = transcript.challenge_scalar("epsilon")
    transcript.append_commitment("z1_commit", cm["z1"])
    transcript.append_commitment("z2_commit", cm["z2"])
    alpha = transcript.challenge_scalar("alpha")
    for k in ("q_lo", "q_mid", "q_hi"):
        transcript.append_commitment(k + "_commit", cm[k])
    xi = transcript.challenge_scalar("xi")
    zh = dom.evaluate_vanishing_polynomial(xi)
    l1 = lagrange_evaluation(cv, vk.n, 1, zh, xi)
    ev = proof.evaluations
    alpha2 = alpha * alpha % p
    alpha3 = alpha2 * alpha % p
    alpha4 = alpha3 * alpha % p
    opd = (1 + delta) % p
    eopd = epsilon * opd % p
    # compute_r0
    part1 = (-sum(lagrange_evaluation(cv, vk.n, pt, zh, xi) * v for v, pt in zip(pub_inputs, vk.pi_roots))) % p
    part2 = alpha * ev.z1_next % p * (ev.a + beta * ev.sigma1 + gamma) % p * (ev.b + beta * ev.sigma2 + gamma) % p \
        * (ev.c + gamma) % p
    part3 = l1 * alpha2 % p
    part4 = alpha3 * ev.z2_next % p * (eopd + delta * ev.h2) % p * (eopd + ev.h2 + delta * ev.h1_next) % p
    part5 = l1 * alpha4 % p
    r0 = (part1 + part2 + part3 + part4 + part5) % p
    # linearization commitment (13-point MSM, commitment.rs:32-45)
    V = vk.commits
    bz = beta * xi % p
    scal_pts = [
        (ev.a * ev.b, V["q_m"]), (ev.a, V["q_l"]), (ev.b, V["q_r"]), (ev.c, V["q_o"]), (1, V["q_c"]),
        ((alpha * (bz + ev.a + gamma) % p * (bz * K1 + ev.b + gamma) % p * (bz * K2 + ev.c + gamma) + l1 * alpha2) % p,
         cm["z1"]),
        ((-alpha) * beta % p * ev.z1_next % p * (beta * ev.sigma1 + ev.a + gamma) % p
         * (beta * ev.sigma2 + ev.b + gamma) % p, V["sigma3"]),
        ((alpha3 * opd % p * (epsilon + ev.q_lookup * ev.c) % p * (eopd + ev.t + delta * ev.t_next) + alpha4 * l1) % p,
         cm["z2"]),
        ((-alpha3) * ev.z2_next % p * (eopd + ev.h2 + delta * ev.h1_next) % p, cm["h1"]),
        (alpha4 * alpha % p * ev.t % p, V["q_table"]),
    ]
    xn2 = (zh + 1) * xi % p * xi % p
    scal_pts += [((-zh) % p, cm["q_lo"]), ((-zh) * xn2 % p, cm["q_mid"]), ((-zh) * xn2 % p * xn2 % p, cm["q_hi"])]
    r_commit = C.msm_naive(cv, [pt for _, pt in scal_pts], [s % p for s, _ in scal_pts])
    for name, key in (("a_eval", "a"), ("b_eval", "b"), ("c_eval", "c"), ("sigma1_eval", "sigma1"),
                      ("sigma2_eval", "sigma2"), ("z1_next_eval", "z1_next"), ("q_lookup_eval", "q_lookup"),
                      ("t_eval", "t"), ("t_next_eval", "t_next"), ("z2_next_eval", "z2_next"),
                      ("h1_next_eval", "h1_next"), ("h2_eval", "h2")):
        transcript.append_scalar(name, getattr(ev, key))
    eta = transcript.challenge_scalar("eta")

    def pair(commits, point, values, w):
        comb_c: Point = None
        comb_v, ch = 0, 1
        for c_, v in zip(commits, values):
            comb_c = C.add(cv, comb_c, C.scalar_mul(cv, ch, c_))
            comb_v = (comb_v + ch * v) % p
            ch = ch * eta % p
        lhs = C.add(cv, comb_c, C.neg(cv, C.scalar_mul(cv, comb_v, C.generator(cv))))
        return C.add(cv, lhs, C.scalar_mul(cv, point % p, w)), w

    return [pair([r_commit, cm["a"], cm["b"], cm["c"], V["sigma1"], V["sigma2"], V["q_lookup"], cm["t"], cm["h2"]], xi,
                 [r0, ev.a, ev.b, ev.c, ev.sigma1, ev.sigma2, ev.q_lookup, ev.t, ev.h2], proof.aw_opening),
            pair([cm["z1"], cm["z2"], cm["t"], cm["h1"]], xi * dom.group_gen % p,
                 [ev.z1_next, ev.z2_next, ev.t_next, ev.h1_next], proof.saw_opening)]


def verify(cv: Curve, tau: int, vk: VerifierKey, proof: Proof, transcript, pub_inputs: Sequence[int]) -> bool:
    """proof_system/proof.rs:285-503 with the pairing check e(L, h) == e(W, tau h) of each opening replaced by the
    equivalent G1 identity under the known test trapdoor: L == tau W."""
    return all(L == C.scalar_mul(cv, tau % cv.fr.p, W) for L, W in verify_prepare(cv, vk, proof, transcript, pub_inputs))


def new_seeded_transcript(cv: Curve, vk: VerifierKey, kind: str = "merlin"):
    """plonk.rs:105-106 / 120-121: T::new("ZKT Plonk"); vk.seed_transcript."""
    if kind == "merlin":
        tr = MerlinTranscript(cv, "ZKT Plonk")
    else:
        from .transcript import EthereumTranscript
        tr = EthereumTranscript(cv, "ZKT Plonk")
    seed_transcript(vk, tr)
    return tr
