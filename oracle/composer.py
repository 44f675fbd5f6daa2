"""The reference's circuit builder with LTVariable operands, and the gadgets built on it (oracle; TEST INFRASTRUCTURE
ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product).

Restates, gate by gate and in the reference's allocation order:
  * LTVariable                       plonk-core/src/constraint_system/variable.rs:40-91
  * Selectors + by_{left,right,out}_lt   constraint_system/composer.rs:25-116
  * add / sub / mul / div / square / linear_transform gates   constraint_system/arithmetic.rs:15-200
  * lookup_constrain, equal_constrain, bits_le_constrain, set_variable_public, conditional_select
                                     constraint_system/mod.rs:140-240, 318-373
  * boolean_gate                     constraint_system/boolean.rs:26-34
  * PlonkSpecRef / PoseidonRef::hash plonk-hashing/src/hasher/poseidon/spec.rs:18-112, 174-219, 239-316, 343-375
  * merkle_proof / PoECircuit        plonk-hashing/src/merkle/binary.rs:8-79
  * MerkleTree (native store)        gadgets/src/merkle_tree.rs:57-111
  * WithdrawCircuit::synthesize      circuits/src/withdraw.rs:57-150, public inputs as bin/src/main.rs:248-271 orders them

Both composer modes of the reference are one object here (oracle/plonk.py ConstraintSystem): a gate pushes its selectors
(Setup) AND its wires / assigned value (Proving), so one run yields the circuit and the witness.  The reference holds no
known answer for any of this ("parity unpinned"): what pins it is `check_satisfied()` on every circuit built, the gate
counts of SURVEY.md 8d.4 and the native == in-circuit hash equality the reference's own sanity_test checks
(spec.rs:386-420)."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

from . import poseidon as NP
from .plonk import ConstraintSystem, ZERO_VAR


class LT:
    """LTVariable<F> (variable.rs:40-48): value = coeff * var + offset."""
    __slots__ = ("var", "coeff", "offset")

    def __init__(self, var: int = ZERO_VAR, coeff: int = 1, offset: int = 0):
        self.var, self.coeff, self.offset = var, coeff, offset

    @staticmethod
    def constant(v: int) -> "LT":          # variable.rs:67-73
        return LT(ZERO_VAR, 1, v)

    @staticmethod
    def zero() -> "LT":                    # variable.rs:62-64
        return LT(ZERO_VAR, 1, 0)

    def linear_transform(self, p: int, coeff: int, offset: int) -> "LT":
        """variable.rs:77-86, literally: the new `coeff` shadows the argument BEFORE the offset is computed, so
        offset' = self.offset * (self.coeff * coeff) + offset.  Identical to the algebraic rule whenever
        self.coeff == 1 or self.offset == 0, which holds at every call the Poseidon gadget makes."""
        coeff = self.coeff * coeff % p
        offset = (self.offset * coeff + offset) % p
        return LT(self.var, coeff, offset)


class Selectors:
    """composer.rs:25-116."""
    __slots__ = ("p", "q_m", "q_l", "q_r", "q_o", "q_c", "q_lookup")

    def __init__(self, p: int, q_m=0, q_l=0, q_r=0, q_o=0, q_c=0, q_lookup=0):
        self.p = p
        self.q_m, self.q_l, self.q_r, self.q_o, self.q_c, self.q_lookup = q_m % p, q_l % p, q_r % p, q_o % p, q_c % p, q_lookup

    def by_left_lt(self, w: LT) -> "Selectors":     # composer.rs:84-93
        p = self.p
        q_m = self.q_m * w.coeff % p
        q_l = self.q_l * w.coeff % p
        self.q_r = (self.q_r + self.q_m * w.offset) % p
        self.q_c = (self.q_c + self.q_l * w.offset) % p
        self.q_m, self.q_l = q_m, q_l
        return self

    def by_right_lt(self, w: LT) -> "Selectors":    # composer.rs:96-105
        p = self.p
        q_m = self.q_m * w.coeff % p
        q_r = self.q_r * w.coeff % p
        self.q_l = (self.q_l + self.q_m * w.offset) % p
        self.q_c = (self.q_c + self.q_r * w.offset) % p
        self.q_m, self.q_r = q_m, q_r
        return self

    def by_out_lt(self, w: LT) -> "Selectors":      # composer.rs:108-114
        p = self.p
        q_o = self.q_o * w.coeff % p
        self.q_c = (self.q_c + self.q_o * w.offset) % p
        self.q_o = q_o
        return self


class Composer(ConstraintSystem):
    """ConstraintSystem<F, TABLE_SIZE> with the LTVariable gate set.  Variables are indices into `values`."""

    def lt(self, var: int) -> LT:                   # From<Variable> for LTVariable, variable.rs:50-58
        return LT(var, 1, 0)

    def value_of_lt(self, x: LT) -> int:            # variable.rs:135-142
        return (self.value_of(x.var) * x.coeff + x.offset) % self.p

    def _gate(self, w_l: int, w_r: int, w_o: int, s: Selectors, pi: Optional[int] = None):
        self.arith_constrain(w_l, w_r, w_o, q_m=s.q_m, q_l=s.q_l, q_r=s.q_r, q_o=s.q_o, q_c=s.q_c, q_lookup=s.q_lookup, pi=pi)

    # -- arithmetic.rs ------------------------------------------------------------------------------------------------
    def add_gate(self, x: LT, y: LT) -> int:        # arithmetic.rs:15-43
        z = self.assign_variable(self.value_of_lt(x) + self.value_of_lt(y))
        self._gate(x.var, y.var, z, Selectors(self.p, q_l=1, q_r=1, q_o=-1).by_left_lt(x).by_right_lt(y))
        return z

    def sub_gate(self, x: LT, y: LT) -> int:        # arithmetic.rs:46-74
        z = self.assign_variable(self.value_of_lt(x) - self.value_of_lt(y))
        self._gate(x.var, y.var, z, Selectors(self.p, q_l=1, q_r=-1, q_o=-1).by_left_lt(x).by_right_lt(y))
        return z

    def mul_gate(self, x: LT, y: LT) -> int:        # arithmetic.rs:77-104
        z = self.assign_variable(self.value_of_lt(x) * self.value_of_lt(y))
        self._gate(x.var, y.var, z, Selectors(self.p, q_m=1, q_o=-1).by_left_lt(x).by_right_lt(y))
        return z

    def div_gate(self, x: LT, y: LT) -> int:        # arithmetic.rs:107-135: y * z - x = 0, wires (y, z, x)
        z = self.assign_variable(self.value_of_lt(x) * pow(self.value_of_lt(y), -1, self.p))
        self._gate(y.var, z, x.var, Selectors(self.p, q_m=1, q_o=-1).by_left_lt(y).by_out_lt(x))
        return z

    def square_gate(self, x: LT) -> int:            # arithmetic.rs:138-164
        v = self.value_of_lt(x)
        y = self.assign_variable(v * v)
        self._gate(x.var, x.var, y, Selectors(self.p, q_m=1, q_o=-1).by_left_lt(x).by_right_lt(x))
        return y

    def linear_transform_gate(self, x: LT, y: LT, a: int, b: int, c: int) -> int:   # arithmetic.rs:167-200
        z = self.assign_variable(self.value_of_lt(x) * a + self.value_of_lt(y) * b + c)
        self._gate(x.var, y.var, z, Selectors(self.p, q_l=a, q_r=b, q_o=-1, q_c=c).by_left_lt(x).by_right_lt(y))
        return z

    # -- boolean.rs / mod.rs ------------------------------------------------------------------------------------------
    def boolean_gate(self, x: int) -> int:          # boolean.rs:26-34
        self._gate(x, x, x, Selectors(self.p, q_m=1, q_o=-1))
        return x

    def lookup_constrain(self, x: LT):              # mod.rs:140-160
        w_o = self.assign_variable(self.value_of_lt(x))
        self._gate(x.var, ZERO_VAR, w_o, Selectors(self.p, q_l=1, q_o=-1, q_lookup=1).by_left_lt(x))

    def equal_constrain(self, x: LT, y: LT):        # mod.rs:164-172
        self._gate(x.var, y.var, ZERO_VAR, Selectors(self.p, q_l=1, q_r=-1).by_left_lt(x).by_right_lt(y))

    def bits_le_constrain(self, bits: Sequence[int]) -> int:     # mod.rs:175-213 (multiplier squares each level, in u64)
        assert len(bits) & (len(bits) - 1) == 0 and bits, "bits length must be a power of two"
        vs = list(bits)
        multiplier = 2
        while len(vs) > 1:
            nxt = []
            for k in range(0, len(vs), 2):
                lo, hi = vs[k], vs[k + 1]
                new = self.assign_variable(self.value_of(lo) + self.value_of(hi) * multiplier)
                self._gate(lo, hi, new, Selectors(self.p, q_l=1, q_r=multiplier, q_o=-1))
                nxt.append(new)
            vs = nxt
            multiplier = multiplier * multiplier % (1 << 64)     # `multiplier *= multiplier` on a u64 (wraps in release)
        return vs[0]

    def set_variable_public(self, x: LT):           # mod.rs:216-240
        self._gate(ZERO_VAR, ZERO_VAR, x.var, Selectors(self.p, q_o=-1).by_out_lt(x), pi=self.value_of_lt(x))

    def conditional_select(self, bit: int, a: LT, b: LT) -> int:   # mod.rs:318-373
        bv = self.value_of(bit)
        assert bv in (0, 1)
        xv = bv * self.value_of_lt(a) % self.p
        yv = (1 - bv) * self.value_of_lt(b) % self.p
        x = self.assign_variable(xv)
        y = self.assign_variable(yv)
        z = self.assign_variable(xv + yv)
        self._gate(bit, a.var, x, Selectors(self.p, q_m=1, q_o=-1).by_right_lt(a))
        self._gate(bit, b.var, y, Selectors(self.p, q_m=-1, q_r=1, q_o=-1).by_right_lt(b))
        self._gate(x, y, z, Selectors(self.p, q_l=1, q_r=1, q_o=-1))
        return z


# ---------------------------------------------------------------------------------------------------------------------
# Poseidon: PoseidonConstants (the fields the gadget reads) and PoseidonRef<ConstraintSystem, PlonkSpecRef, _, WIDTH>
# ---------------------------------------------------------------------------------------------------------------------
class PoseidonParams:
    """constants.rs:12-22 / 55-93: width, half_full_rounds, partial_rounds, round_constants, mds (m[i][j]), domain_tag =
    2^arity - 1."""

    def __init__(self, p: int, width: int, half_full: int, partial: int, rc: Sequence[int], mds: Sequence[Sequence[int]],
                 domain_tag: Optional[int] = None):
        assert len(rc) >= width * (2 * half_full + partial), "Not enough round constants"      # constants.rs:61-64
        self.p, self.width, self.half_full, self.partial = p, width, half_full, partial
        self.rc, self.mds = [x % p for x in rc], [[x % p for x in row] for row in mds]
        self.domain_tag = ((1 << (width - 1)) - 1) % p if domain_tag is None else domain_tag % p

    @property
    def gates_per_hash(self) -> int:     # Rf (3W + W^2) + Rp (3 + W^2)
        W = self.width
        return 2 * self.half_full * (3 * W + W * W) + self.partial * (3 + W * W)

    def native(self, inputs: Sequence[int]) -> int:   # PoseidonRef<(), NativePlonkSpecRef, ..>::hash
        if len(inputs) > self.width - 1:
            raise ValueError("Poseidon Error: FullBuffer")           # spec.rs:253-255
        return NP.permute(self.p, self.width, self.half_full, self.partial, self.rc, self.mds, self.domain_tag, inputs)[0]


def _power_of_5(cs: Composer, x: LT) -> LT:          # spec.rs:107-111: three mul gates, three fresh variables
    t = cs.lt(cs.mul_gate(x, x))
    t = cs.lt(cs.mul_gate(t, t))
    return cs.lt(cs.mul_gate(t, x))


def _product_mds(cs: Composer, prm: PoseidonParams, state: List[LT]) -> List[LT]:   # spec.rs:73-88
    W, p = prm.width, prm.p
    result = [LT.zero() for _ in range(W)]
    for j in range(W):
        for i in range(W):
            tmp = state[i].linear_transform(p, prm.mds[i][j], 0)     # mul_constant, spec.rs:210-217
            result[j] = cs.lt(cs.add_gate(result[j], tmp))           # add, spec.rs:186-192: every term is a gate
    return result


def poseidon_hash(cs: Composer, prm: PoseidonParams, inputs: Sequence[LT]) -> LT:
    """FieldHasher::hash (spec.rs:364-372): reset (:239-245), input (:249-263), output_hash (:267-316)."""
    W, p = prm.width, prm.p
    if len(inputs) > W - 1:
        raise ValueError("Poseidon Error: FullBuffer")               # spec.rs:253-255 -> Error::SynthesisError
    st: List[LT] = [LT.constant(prm.domain_tag)] + list(inputs) + [LT.zero()] * (W - 1 - len(inputs))
    # bookkeeping for the tests (not in the reference): where this hash's variables start and what it was fed
    cs.__dict__.setdefault("hash_calls", []).append((len(cs.values), [(x.var, x.coeff, x.offset) for x in inputs]))
    off = 0
    for r in range(2 * prm.half_full + prm.partial):
        if r < prm.half_full or r >= prm.half_full + prm.partial:     # full_round, spec.rs:18-37
            st = [_power_of_5(cs, st[i].linear_transform(p, 1, prm.rc[off + i])) for i in range(W)]
        else:                                                         # partial_round, spec.rs:39-54
            st = [st[i].linear_transform(p, 1, prm.rc[off + i]) for i in range(W)]
            st[0] = _power_of_5(cs, st[0])
        off += W
        st = _product_mds(cs, prm, st)
    return st[1]


def gadget_trace(prm: PoseidonParams, input_values: Sequence[int]) -> List[int]:
    """The values of the gates_per_hash variables one hash allocates, in allocation order, for inputs that are plain
    variables (coeff 1, offset 0).  Computed WITHOUT the composer, straight from the definition of the gates, so that
    it checks the composer as much as the composer checks it: per full round x^2, x^4, x^5 of every element, then the
    W^2 running sums of product_mds (j outer, i inner); per partial round the three powers of element 0, then the sums."""
    W, p = prm.width, prm.p
    st = [prm.domain_tag] + [x % p for x in input_values] + [0] * (W - 1 - len(input_values))
    out: List[int] = []
    off = 0
    for r in range(2 * prm.half_full + prm.partial):
        full = r < prm.half_full or r >= prm.half_full + prm.partial
        st = [(st[i] + prm.rc[off + i]) % p for i in range(W)]
        off += W
        for i in range(W if full else 1):
            x2 = st[i] * st[i] % p
            x4 = x2 * x2 % p
            st[i] = x4 * st[i] % p
            out += [x2, x4, st[i]]
        nx = []
        for j in range(W):
            acc = 0
            for i in range(W):
                acc = (acc + prm.mds[i][j] * st[i]) % p
                out.append(acc)
            nx.append(acc)
        st = nx
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Merkle path gadget and the native tree
# ---------------------------------------------------------------------------------------------------------------------
def poe_synthesize(cs: Composer, prm: PoseidonParams, leaf_index: int, path_elements: Sequence[int], leaf: LT) -> LT:
    """PoECircuit::synthesize (binary.rs:42-78) + merkle_proof (:8-30); returns the root."""
    positions = []
    for layer in range(len(path_elements)):
        var = cs.assign_variable((leaf_index >> layer) & 1)
        positions.append(cs.boolean_gate(var))
    witness = [(positions[k], cs.lt(cs.assign_variable(node))) for k, node in enumerate(path_elements)]
    cur = leaf
    for is_left, node in witness:
        left = cs.conditional_select(is_left, node, cur)
        right = cs.conditional_select(is_left, cur, node)
        cur = poseidon_hash(cs, prm, [cs.lt(left), cs.lt(right)])
    return cur


class NativeMerkleTree:
    """gadgets/src/merkle_tree.rs:57-111: sparse tree, empty subtrees hash from H::empty_hash() = 0 upwards."""

    def __init__(self, prm: PoseidonParams, height: int):
        self.prm, self.height = prm, height
        self.tree: Dict[Tuple[int, int], int] = {}
        self.nodes: List[int] = []
        h = 0
        for _ in range(height):
            self.nodes.append(h)
            h = prm.native([h, h])
        self.root = 0
        self.next_index = 0

    def merkle_path(self, index: int) -> List[int]:
        return [self.tree.get((layer, (index >> layer) ^ 1), self.nodes[layer]) for layer in range(self.height)]

    def add_leaf(self, h: int) -> int:
        index = self.next_index
        self.next_index += 1
        for layer in range(self.height):
            idx = index >> layer
            self.tree[(layer, idx)] = h
            w = self.tree.get((layer, idx ^ 1), self.nodes[layer])
            h = self.prm.native([w, h]) if idx & 1 else self.prm.native([h, w])
        self.root = h
        return index


# ---------------------------------------------------------------------------------------------------------------------
# WithdrawCircuit
# ---------------------------------------------------------------------------------------------------------------------
def withdraw_gate_count(prm: PoseidonParams, inputs: int, height: int) -> int:
    """Per note (3 + H) P + 7 H + 4 (three hashes, div, public nullifier, H position bits, per level two selects of three
    gates and a hash, equal, lookup); global 2 P + 130 + INPUTS (public root, 64 bit gates + 63 recombination gates,
    INPUTS - 1 additions, the balance gate, two hashes, two public rows).  SURVEY.md 8d.4 estimated the global part as
    2 P + 133 + INPUTS; counting withdraw.rs:73-147 gate by gate gives 130."""
    P_ = prm.gates_per_hash
    return inputs * ((3 + height) * P_ + 7 * height + 4) + 2 * P_ + 130 + inputs


def withdraw_synthesize(cs: Composer, prm: PoseidonParams, secrets: Sequence[int], identifiers: Sequence[int],
                        amounts: Sequence[int], poes: Sequence[Tuple[int, Sequence[int]]], root: int, new_secret: int,
                        new_identifier: int, withdraw_amount: int) -> None:
    """WithdrawCircuit::synthesize, circuits/src/withdraw.rs:57-150, statement by statement."""
    p = cs.p
    amount_in = sum(amounts)
    assert amount_in >= withdraw_amount, "invalid withdraw amount"
    amount_out = amount_in - withdraw_amount
    amount_in_vars = [cs.assign_variable(a) for a in amounts]
    identifier_vars = [cs.assign_variable(i) for i in identifiers]
    one = LT.constant(1)
    pub_root = cs.lt(cs.assign_variable(root))
    cs.set_variable_public(pub_root)
    for amount_var, identifier_var, secret, (leaf_index, path) in zip(amount_in_vars, identifier_vars, secrets, poes):
        secret_var = cs.lt(cs.assign_variable(secret))
        commitment = poseidon_hash(cs, prm, [secret_var])
        secret_inv = cs.div_gate(one, secret_var)
        nullifier = poseidon_hash(cs, prm, [cs.lt(secret_inv)])
        cs.set_variable_public(nullifier)
        leaf = poseidon_hash(cs, prm, [cs.lt(identifier_var), cs.lt(amount_var), commitment])
        root_var = poe_synthesize(cs, prm, leaf_index, path, leaf)
        cs.equal_constrain(root_var, pub_root)
        cs.lookup_constrain(cs.lt(identifier_var))
    bits = []
    for k in range(64):                                              # view_bits::<Lsb0>() of a u64
        bits.append(cs.boolean_gate(cs.assign_variable((amount_out >> k) & 1)))
    amount_out_var = cs.bits_le_constrain(bits)
    left_var = amount_in_vars[0]
    right_var = ZERO_VAR
    for amount_var in amount_in_vars[1:]:
        right_var = cs.add_gate(cs.lt(right_var), cs.lt(amount_var))
    cs._gate(left_var, right_var, amount_out_var, Selectors(p, q_l=-1, q_r=-1, q_o=1), pi=withdraw_amount % p)
    new_secret_var = cs.lt(cs.assign_variable(new_secret))
    new_identifier_var = cs.lt(cs.assign_variable(new_identifier))
    new_commitment = poseidon_hash(cs, prm, [new_secret_var])
    new_leaf = poseidon_hash(cs, prm, [new_identifier_var, cs.lt(amount_out_var), new_commitment])
    cs.set_variable_public(new_identifier_var)
    cs.set_variable_public(new_leaf)


def withdraw_instance(cv, prm: PoseidonParams, inputs: int, height: int, seed: int = 1, table_size: int = 1024,
                      decoys: int = 5):
    """A consistent WithdrawCircuit input set the way the CLI assembles one (bin/src/main.rs:198-271): a native Merkle
    tree holding `decoys + inputs` leaves, the spent notes among them, the identifier set as the lookup table.  Returns
    (composer with the circuit synthesized, public inputs in the CLI's order)."""
    import random
    rnd = random.Random(seed)
    p = prm.p
    assert inputs <= 1 << height, "the tree cannot hold that many notes"
    decoys = min(decoys, (1 << height) - inputs)
    tree = NativeMerkleTree(prm, height)
    ident_set = [rnd.randrange(1, p) for _ in range(7)]
    notes = []
    for k in range(decoys + inputs):
        secret, ident, amount = rnd.randrange(1, p), rnd.choice(ident_set), rnd.randrange(1, 1 << 40)
        leaf = prm.native([ident, amount, prm.native([secret])])
        idx = tree.add_leaf(leaf)
        if k % 2 == 1 and len(notes) < inputs or decoys + inputs - k <= inputs - len(notes):
            notes.append((secret, ident, amount, idx))
    assert len(notes) == inputs
    new_secret, new_ident = rnd.randrange(1, p), rnd.choice(ident_set)
    withdraw_amount = rnd.randrange(1, sum(nt[2] for nt in notes))
    cs = Composer(cv, ident_set, table_size)
    withdraw_synthesize(cs, prm, [nt[0] for nt in notes], [nt[1] for nt in notes], [nt[2] for nt in notes],
                        [(nt[3], tree.merkle_path(nt[3])) for nt in notes], tree.root, new_secret, new_ident, withdraw_amount)
    amount_out = sum(nt[2] for nt in notes) - withdraw_amount
    nullifiers = [prm.native([pow(nt[0], -1, p)]) for nt in notes]
    new_leaf = prm.native([new_ident, amount_out, prm.native([new_secret])])
    public_inputs = [tree.root] + nullifiers + [withdraw_amount % p, new_ident, new_leaf]    # main.rs:263-269
    return cs, public_inputs
