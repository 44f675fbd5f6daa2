"""bench.py's workload generator (tools/withdraw_workload.py: the withdraw circuit in closed form) against the oracle's
operational restatement of the reference's composer (oracle/composer.py: LTVariable transforms folded gate by gate):
same selectors, wires, public inputs, permutation, host-side witness and hash-call bookkeeping, row for row.  CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import withdraw_workload as WW  # noqa: E402

from oracle import fields as F, composer as OC, plonk as P, coracle as K  # noqa: E402


def _oracle_twin(cv, hs, inst, table_size=1024):
    prm = OC.PoseidonParams(cv.fr.p, hs.width, hs.half_full, hs.partial, hs.rc, hs.mds, hs.tag)
    cs = OC.Composer(cv, inst["ident_set"], table_size)
    OC.withdraw_synthesize(cs, prm, inst["secrets"], inst["identifiers"], inst["amounts"], inst["poes"], inst["root"],
                           inst["new_secret"], inst["new_identifier"], inst["withdraw_amount"])
    return prm, cs


@pytest.mark.parametrize("cvname,width,inputs,height", [("bn254", 4, 1, 3), ("bn254", 5, 2, 2), ("bn254", 4, 3, 2), ("bls12_381", 5, 2, 1)])
def test_closed_form_layout_equals_the_composer(cvname, width, inputs, height):
    cv = F.CURVES[cvname]
    p = cv.fr.p
    hs = WW.reference_hasher(p, width) if cvname == "bn254" else WW.synthetic_hasher(p, width, 4, 3)
    inst = WW.make_instance(hs, inputs, height, seed=5 + width)
    L = WW.layout(hs, inst)
    prm, cs = _oracle_twin(cv, hs, inst)
    assert cs.check_satisfied()
    assert L.n_gates == cs.n_gates == OC.withdraw_gate_count(prm, inputs, height)
    for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "q_lookup"):
        assert L.q[k] == getattr(cs, k), k
    conv = lambda ws: [WW.ZERO if v == P.ZERO_VAR else v for v in ws]
    assert L.w[0] == conv(cs.w_l) and L.w[1] == conv(cs.w_r) and L.w[2] == conv(cs.w_o)
    assert L.pi == cs.pi
    assert L.hash_calls == [(b, tuple(v for (v, _, _) in ins)) for b, ins in cs.hash_calls]
    # host-side witness: everything outside the hash traces; inside them the layout holds zeros (the device writes there)
    want = list(cs.values)
    for b, _ in cs.hash_calls:
        want[b:b + hs.per_hash] = [0] * hs.per_hash
    assert L.values == want
    # the native hasher of the workload == the oracle's
    assert hs.native(inst["secrets"][:1]) == prm.native(inst["secrets"][:1])
    # permutation and the ten setup vectors
    n = cs.circuit_bound()
    log_n = n.bit_length() - 1
    sig = cs.sigma_mappings(n)
    for col, got in enumerate(WW.sigma_columns(L, n)):
        assert got.tolist() == [c * n + r for (c, r) in sig[col]], col
    be = K.CBackend(cv, None)
    ev = P.setup_evals(be, cs)
    mine = WW.setup_vectors(L, log_n, cv.fr.generator if hasattr(cv.fr, "generator") else (5 if cvname == "bn254" else 7))
    for k in P.PK_POLYS:
        assert mine[k] == ev[k], k


def test_shapes_fill_their_domains():
    p = F.BN254.fr.p
    for log_n, (w, inputs, height) in WW.SHAPES.items():
        hs = WW.reference_hasher(p, w)
        gates = inputs * ((3 + height) * hs.per_hash + 7 * height + 4) + 2 * hs.per_hash + 130 + inputs
        assert (1 << (log_n - 1)) < gates <= (1 << log_n), (log_n, gates)
    with pytest.raises(ValueError):
        WW.layout(WW.reference_hasher(p, 3), WW.make_instance(WW.reference_hasher(p, 4), 1, 2, 1))


def test_hash_calls_are_scheduled_by_dependency():
    """zkt_plonk_amd.PoseidonGadget.levels (host logic, no GPU): hashes of one k_poseidon_gadget launch must be independent, so
    a hash fed by another hash's output variable goes into a later launch.  In the withdraw circuit those are exactly the
    leaf hashes (they take the commitment hash, withdraw.rs:91-94 and :141-144); the Merkle-path hashes take select outputs
    (host-made variables) and stay in the first launch."""
    from zkt_plonk_amd.poseidon import PoseidonGadget, VARIABLE_ZERO
    p = F.BN254.fr.p
    hs = WW.reference_hasher(p, 4)
    L = WW.layout(hs, WW.make_instance(hs, 3, 5, seed=9))
    g = PoseidonGadget.__new__(PoseidonGadget)
    g.width, g.vars_per_hash, g.calls = hs.width, hs.per_hash, list(L.hash_calls)
    lv = g.levels()
    assert [len(x) for x in lv] == [len(L.hash_calls) - 4, 4]
    assert all(len(L.hash_calls[k][1]) == 3 for k in lv[1])
    g.calls = [(10, (1,)), (2000, (2, 3, 10 + 57)), (4000, (VARIABLE_ZERO,)), (6000, (2500, 4500))]
    assert g.levels() == [[0, 2], [1], [3]]
    g.calls = [(10, (2000,)), (2000, (11,))]
    with pytest.raises(ValueError):
        g.levels()
