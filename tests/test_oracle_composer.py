"""The oracle's restatement of the reference's composer with LTVariable operands, the Poseidon gadget on it, the Merkle
path gadget and WithdrawCircuit::synthesize (oracle/composer.py).  CPU only.  "Parity unpinned" by the reference (no
known answers for any of it); pinned here the way the reference's own tests pin it: every circuit satisfies its gates
(constraint_system/helper.rs test_gate_constraints), the in-circuit hash equals the native one (spec.rs:386-420
sanity_test), gate counts -- plus an independent, composer-free computation of the gadget's variables."""
import json
import os

import numpy as np
import pytest

from oracle import fields as F, composer as OC, plonk as P, coracle as K
from helpers import field_elems

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def reference_params(w):
    arr = np.load(os.path.join(HERE, "poseidon_bn254.npz"))
    with open(os.path.join(HERE, "poseidon_bn254.json")) as f:
        meta = json.load(f)["x%d" % w]
    to_int = lambda a: [sum(int(v) << (64 * i) for i, v in enumerate(row)) for row in a]
    rc, mds = to_int(arr["rc_x%d" % w]), to_int(arr["mds_x%d" % w])
    return OC.PoseidonParams(F.BN254.fr.p, w, meta["full_rounds"] // 2, meta["partial_rounds"], rc,
                             [mds[i * w:(i + 1) * w] for i in range(w)], meta["domain_tag"])


def test_linear_transform_is_the_references_not_the_algebraic_one():
    """variable.rs:77-86 computes the new offset with the ALREADY UPDATED coeff (the local shadows the argument).  The two
    rules agree whenever self.coeff == 1 or self.offset == 0 -- every call the Poseidon gadget makes -- and differ otherwise;
    the restatement follows the code."""
    p = F.BN254.fr.p
    x = OC.LT(3, 5, 11)
    y = x.linear_transform(p, 7, 2)
    assert (y.var, y.coeff, y.offset) == (3, 35, (11 * 35 + 2) % p)          # algebraic: 11 * 7 + 2
    z = OC.LT(3, 1, 11).linear_transform(p, 7, 2)
    assert (z.coeff, z.offset) == (7, 11 * 7 + 2)
    assert OC.LT.constant(9).linear_transform(p, 1, 4).offset == 13 and OC.LT.zero().linear_transform(p, 6, 0).offset == 0


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381], ids=lambda c: c.name)
def test_lt_gates_satisfy_their_constraints(cv):
    """arithmetic.rs:212-290's test_{add,sub,mul,div}_gate with LTVariable operands that carry coefficients and offsets,
    plus the boolean / select / lookup / public / bits rows: selectors folded by by_{left,right,out}_lt
    (composer.rs:84-114) must make every row vanish on the assigned values."""
    p = cv.fr.p
    cs = OC.Composer(cv, [77, 1234], 8)
    x, y = cs.assign_variable(field_elems(p, 1, 1)[0]), cs.assign_variable(field_elems(p, 2, 1)[0])
    lx, ly = OC.LT(x, 5, 9), OC.LT(y, p - 3, 12345)
    vx, vy = cs.value_of_lt(lx), cs.value_of_lt(ly)
    assert cs.value_of(cs.add_gate(lx, ly)) == (vx + vy) % p
    assert cs.value_of(cs.sub_gate(lx, ly)) == (vx - vy) % p
    assert cs.value_of(cs.mul_gate(lx, ly)) == vx * vy % p
    assert cs.value_of(cs.div_gate(lx, ly)) * vy % p == vx
    assert cs.value_of(cs.square_gate(lx)) == vx * vx % p
    assert cs.value_of(cs.linear_transform_gate(lx, ly, 4, 6, 8)) == (4 * vx + 6 * vy + 8) % p
    assert cs.value_of(cs.add_gate(OC.LT.constant(5), OC.LT.zero())) == 5          # Variable::Zero operands
    bit1, bit0 = cs.boolean_gate(cs.assign_variable(1)), cs.boolean_gate(cs.assign_variable(0))
    assert cs.value_of(cs.conditional_select(bit1, lx, ly)) == vx and cs.value_of(cs.conditional_select(bit0, lx, ly)) == vy
    t = cs.assign_variable(1234 - 9)
    cs.lookup_constrain(OC.LT(t, 1, 9))
    cs.equal_constrain(OC.LT(t, 1, 9), OC.LT.constant(1234))
    cs.set_variable_public(lx)
    bits = [cs.boolean_gate(cs.assign_variable((0xB5 >> k) & 1)) for k in range(8)]
    assert cs.value_of(cs.bits_le_constrain(bits)) == 0xB5
    assert cs.pi[cs.n_gates - 1 - 8 - 7] == vx
    assert cs.check_satisfied()
    for v in range(len(cs.values)):
        if v in (bit1, bit0) or v in bits:
            continue
        cs.values[v] = (cs.values[v] + 1) % p
        assert not cs.check_satisfied(), v
        cs.values[v] = (cs.values[v] - 1) % p
    assert cs.check_satisfied()


@pytest.mark.parametrize("w", [3, 4, 5])
def test_poseidon_gadget_on_the_reference_parameter_sets(w):
    """spec.rs:386-420 sanity_test on Bn254x3 / x4 / x5: in-circuit hash == native hash, all gates satisfied; gates per hash
    804 / 1288 / 1888 = variables per hash; the variables equal a composer-free computation; a flipped
    variable breaks exactly the rows that read it."""
    cv = F.BN254
    p = cv.fr.p
    prm = reference_params(w)
    assert prm.gates_per_hash == {3: 804, 4: 1288, 5: 1888}[w]
    for arity in range(w):
        ins = field_elems(p, 50 + 10 * w + arity, arity)
        cs = OC.Composer(cv, [1], 8)
        vs = [cs.assign_variable(x) for x in ins]
        out = OC.poseidon_hash(cs, prm, [cs.lt(v) for v in vs])
        assert cs.n_gates == prm.gates_per_hash == len(cs.values) - arity
        assert cs.value_of_lt(out) == prm.native(ins)
        assert cs.values[arity:] == OC.gadget_trace(prm, ins)
        assert (out.coeff, out.offset) == (1, 0) and out.var == len(cs.values) - 1 - (w - 2) * w
        assert cs.hash_calls == [(arity, [(v, 1, 0) for v in vs])]
        assert cs.check_satisfied()
        for v in (arity, arity + 1, arity + 3 * w, len(cs.values) // 2, len(cs.values) - 1):
            cs.values[v] ^= 1
            assert not cs.check_satisfied()
            cs.values[v] ^= 1
    with pytest.raises(ValueError):
        OC.poseidon_hash(OC.Composer(cv, [1], 8), prm, [OC.LT.zero()] * w)        # FullBuffer, spec.rs:253-255


def test_poseidon_gadget_takes_lt_inputs_like_the_reference():
    """Inputs with coefficients / offsets (LTVariable operands) are folded into the first round's selectors: satisfied, and
    the value is the native hash of the transformed inputs."""
    cv = F.BLS12_381
    p = cv.fr.p
    prm = OC.PoseidonParams(p, 3, 1, 2, field_elems(p, 5, 12), [field_elems(p, 60 + i, 3) for i in range(3)])
    cs = OC.Composer(cv, [1], 8)
    a, b = cs.assign_variable(1111), cs.assign_variable(2222)
    la, lb = OC.LT(a, 3, 0), OC.LT(b, 1, 17)
    out = OC.poseidon_hash(cs, prm, [la, lb])
    assert cs.value_of_lt(out) == prm.native([3333, 2239]) and cs.check_satisfied()


def test_native_merkle_tree_and_path_gadget():
    """gadgets/src/merkle_tree.rs:57-111 against plonk-hashing/src/merkle/binary.rs:144-180's check: the root the path
    gadget computes in-circuit is the native tree's root, for leaves on both sides and with empty siblings."""
    cv = F.BN254
    prm = reference_params(3)
    H = 5
    tree = OC.NativeMerkleTree(prm, H)
    leaves = field_elems(cv.fr.p, 700, 6)
    for lf in leaves:
        tree.add_leaf(lf)
    assert tree.nodes[0] == 0 and tree.nodes[1] == prm.native([0, 0])
    for idx in (0, 3, 5):
        path = tree.merkle_path(idx)
        cur = leaves[idx]
        for layer, node in enumerate(path):
            cur = prm.native([node, cur]) if (idx >> layer) & 1 else prm.native([cur, node])
        assert cur == tree.root
        cs = OC.Composer(cv, [1], 8)
        root = OC.poe_synthesize(cs, prm, idx, path, cs.lt(cs.assign_variable(leaves[idx])))
        assert cs.value_of_lt(root) == tree.root and cs.check_satisfied()
        assert cs.n_gates == H * (7 + prm.gates_per_hash)


@pytest.mark.parametrize("w,inputs,height", [(4, 1, 3), (5, 2, 2), (4, 3, 2)])
def test_withdraw_circuit_gate_count_public_inputs_and_satisfaction(w, inputs, height):
    """circuits/src/withdraw.rs:57-150 on small const generics: gate count by the closed formula, 4 + INPUTS public inputs
    in the CLI's order (bin/src/main.rs:263-269: root, nullifiers, withdraw amount, new identifier, new leaf), satisfied;
    tampering with a secret breaks it.  Poseidon x3 cannot hash a leaf (three inputs need WIDTH >= 4: FullBuffer), which
    is why the binary's features offer x3 only nominally."""
    cv = F.BN254
    prm = reference_params(w)
    cs, pis = OC.withdraw_instance(cv, prm, inputs, height, seed=3 * w + inputs)
    assert cs.n_gates == OC.withdraw_gate_count(prm, inputs, height)
    assert [cs.pi[k] for k in sorted(cs.pi)] == pis and len(pis) == 4 + inputs
    assert len(cs.hash_calls) == inputs * (3 + height) + 2
    assert cs.check_satisfied()
    secret_var = inputs * 2 + 1                      # amounts, identifiers, root, then the first note's secret
    cs.values[secret_var] = (cs.values[secret_var] + 1) % cv.fr.p
    assert not cs.check_satisfied()
    with pytest.raises(ValueError):
        OC.withdraw_instance(cv, reference_params(3), 1, 1)


def test_withdraw_sizes_of_the_shipped_feature_sets():
    """bin/Cargo.toml:24-41: default features (height-48, notes-3, x4) and the largest shipped set (height-64, notes-4, x5);
    n = 2^20 needs 8 notes (SURVEY.md 8d.4 estimated three gates more per circuit; see withdraw_gate_count)."""
    assert OC.withdraw_gate_count(reference_params(4), 3, 48) == 200793       # -> n = 2^18
    assert OC.withdraw_gate_count(reference_params(5), 4, 64) == 511702       # -> n = 2^19
    assert OC.withdraw_gate_count(reference_params(5), 8, 64) == 1019498      # -> n = 2^20
    assert OC.withdraw_gate_count(reference_params(4), 1, 7) == 15640         # -> n = 2^14 (configs[0])


def test_gadget_circuit_proves_and_verifies_on_the_cpu():
    """Two chained x3 gadgets through the oracle's prover and verifier (the GPU twin is tests/test_gpu_poseidon.py)."""
    cv = F.BN254
    p = cv.fr.p
    prm = reference_params(3)
    cs = OC.Composer(cv, [5, 6, 7], 8)
    a, b = cs.assign_variable(123), cs.assign_variable(456)
    h1 = OC.poseidon_hash(cs, prm, [cs.lt(a), cs.lt(b)])
    h2 = OC.poseidon_hash(cs, prm, [h1])
    cs.set_variable_public(h2)
    assert cs.check_satisfied() and cs.pi[cs.n_gates - 1] == prm.native([prm.native([123, 456])])
    n = cs.circuit_bound()
    tau = 0xABCDE
    srs = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    proof = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), field_elems(p, 3, P.NUM_BLINDERS))
    assert P.verify(cv, tau, vk, proof, P.new_seeded_transcript(cv, vk), [cs.pi[cs.n_gates - 1]])
    assert not P.verify(cv, tau, vk, proof, P.new_seeded_transcript(cv, vk), [cs.pi[cs.n_gates - 1] + 1])
