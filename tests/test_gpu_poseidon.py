"""f3: the batched Poseidon permutation (witness synthesis for Poseidon-heavy circuits) against the
oracle's restatement of plonk-hashing's native spec.  "Parity unpinned" by the reference (constants are generated at
run time, no known answers in the tree): arbitrary constants, every width the withdraw instances use (x3, x4, x5)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import fields as F, coracle as K, poseidon as OP
from helpers import field_elems


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381], ids=lambda c: c.name)
@pytest.mark.parametrize("width,half_full,partial", [(3, 4, 57), (4, 4, 56), (5, 4, 60), (2, 1, 1), (8, 2, 3), (6, 1, 2), (7, 3, 1)])
def test_batched_poseidon_matches_the_native_spec(cv, width, half_full, partial):
    import zkt_plonk_amd as z
    p = cv.fr.p
    rounds = 2 * half_full + partial
    rc = field_elems(p, 11 * width, rounds * width)
    mds = [field_elems(p, 100 + i + width, width) for i in range(width)]
    tag = ((1 << (width - 1)) - 1) % p
    ctx = z.Context(cv.name, 0)
    for arity in sorted({0, 1, width - 1}):
        batch = 300
        ins = [field_elems(p, 7000 + b + arity, arity) for b in range(batch)]
        ins[0] = [0] * arity
        ins[1] = [p - 1] * arity
        arr = K.fr_to_mont(cv, [x for row in ins for x in row]).reshape(batch, arity, 4) if arity else np.zeros((batch, 0, 4), np.uint64)
        got, states = ctx.poseidon_hash_batch(width, half_full, partial, K.fr_to_mont(cv, rc),
                                              K.fr_to_mont(cv, [x for row in mds for x in row]), K.fr_to_mont(cv, [tag])[0], arr,
                                              trace=True)
        hashes = K.fr_from_mont(cv, got)
        for b in (0, 1, 2, 57, batch - 1):
            want, trace = OP.permute(p, width, half_full, partial, rc, mds, tag, ins[b])
            assert hashes[b] == want, (arity, b)
            assert K.fr_from_mont(cv, states[b].reshape(-1, 4)) == [x for row in trace for x in row], (arity, b)
        assert len(set(hashes[2:])) == batch - 2 if arity else len(set(hashes)) == 1
    import zkt_plonk_amd._lib as L
    with pytest.raises(L.ZktError):
        ctx.poseidon_hash_batch(3, 4, 57, K.fr_to_mont(cv, rc[:65 * 3] if len(rc) >= 195 else field_elems(p, 1, 195)),
                                K.fr_to_mont(cv, field_elems(p, 2, 9)), K.fr_to_mont(cv, [tag])[0],
                                np.zeros((4, 3, 4), np.uint64))          # arity 3 does not fit width 3
    ctx.close()


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381], ids=lambda c: c.name)
def test_poseidon_committed_fixture(cv):
    """tests/golden/vectors_r02.json: one width-5 permutation with splitmix64 constants."""
    import json, os
    import zkt_plonk_amd as z
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors_r02.json")) as f:
        q = json.load(f)[cv.name]["poseidon"]
    p = cv.fr.p
    W = q["width"]
    rc = field_elems(p, q["rc_seed"], (2 * q["half_full"] + q["partial"]) * W)
    mds = [x for i in range(W) for x in field_elems(p, q["mds_seed"] + i, W)]
    ins = K.fr_to_mont(cv, field_elems(p, q["input_seed"], 4)).reshape(1, 4, 4)
    ctx = z.Context(cv.name, 0)
    got = ctx.poseidon_hash_batch(W, q["half_full"], q["partial"], K.fr_to_mont(cv, rc), K.fr_to_mont(cv, mds),
                                  K.fr_to_mont(cv, [q["domain_tag"]])[0], ins)
    assert "%x" % K.fr_from_mont(cv, got)[0] == q["hash"]
    ctx.close()


def _reference_params(w):
    """The BN254 parameter set the withdraw circuit hashes with (tests/golden/poseidon_bn254.*, made by
    tests/golden/make_poseidon_golden.py from gadgets/src/poseidon/bn254_x{3,4,5}.rs through the restated parse_vec)."""
    import json, os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    arr = np.load(os.path.join(here, "poseidon_bn254.npz"))
    with open(os.path.join(here, "poseidon_bn254.json")) as f:
        meta = json.load(f)["x%d" % w]
    to_int = lambda a: [sum(int(v) << (64 * i) for i, v in enumerate(row)) for row in a]
    rc, mds = to_int(arr["rc_x%d" % w]), to_int(arr["mds_x%d" % w])
    return meta, rc, [mds[i * w:(i + 1) * w] for i in range(w)]


@pytest.mark.parametrize("w", [3, 4, 5])
def test_device_poseidon_on_the_reference_parameter_sets(w):
    """zkt_poseidon_load + zkt_poseidon_hash_batch_dev (device pointers in and out, parameters resident) on Bn254x3 / x4 /
    x5 -- FULL_ROUNDS 8, PARTIAL_ROUNDS 55 / 56 / 56 -- against the committed digests and, value by value on a sample,
    against the oracle run here.  "Parity unpinned": the reference holds no hash known-answer."""
    import hashlib
    import zkt_plonk_amd as z
    from helpers import digest
    cv = F.BN254
    p = cv.fr.p
    meta, rc, mds = _reference_params(w)
    half_full, partial, batch, arity = meta["full_rounds"] // 2, meta["partial_rounds"], meta["batch"], meta["arity"]
    rounds = 2 * half_full + partial
    ins = [field_elems(p, meta["input_seed"] + b, arity) for b in range(batch)]
    ctx = z.Context(cv.name, 0)
    h = ctx.poseidon_load(w, half_full, partial, K.fr_to_mont(cv, rc), K.fr_to_mont(cv, [x for row in mds for x in row]),
                          K.fr_to_mont(cv, [meta["domain_tag"]])[0])
    d_in, d_out, d_st = ctx.alloc(batch * arity * 32), ctx.alloc(batch * 32), ctx.alloc(batch * (rounds + 1) * w * 32)
    ctx.upload(d_in, K.fr_to_mont(cv, [x for row in ins for x in row]))
    ctx.poseidon_hash_batch_dev(h, d_in, batch, arity, d_out, d_st)
    ctx.synchronize()
    hashes = K.fr_from_mont(cv, ctx.download(d_out, (batch, 4)))
    states = K.fr_from_mont(cv, ctx.download(d_st, (batch * (rounds + 1) * w, 4)))
    assert digest(hashes) == meta["hashes_sha256"] and "%x" % hashes[0] == meta["hash0"]
    assert digest(states) == meta["states_sha256"]
    for b in (0, 17, batch - 1):
        want, trace = OP.permute(p, w, half_full, partial, rc, mds, meta["domain_tag"], ins[b])
        assert hashes[b] == want
        assert states[b * (rounds + 1) * w:(b + 1) * (rounds + 1) * w] == [x for row in trace for x in row]
    # without the states, and the host-pointer form on the same parameters
    ctx.upload(d_out, np.zeros((batch, 4), np.uint64))
    ctx.poseidon_hash_batch_dev(h, d_in, batch, arity, d_out)
    ctx.synchronize()
    assert K.fr_from_mont(cv, ctx.download(d_out, (batch, 4))) == hashes
    got = ctx.poseidon_hash_batch(w, half_full, partial, K.fr_to_mont(cv, rc), K.fr_to_mont(cv, [x for row in mds for x in row]),
                                  K.fr_to_mont(cv, [meta["domain_tag"]])[0],
                                  K.fr_to_mont(cv, [x for row in ins for x in row]).reshape(batch, arity, 4))
    assert K.fr_from_mont(cv, got) == hashes
    import zkt_plonk_amd._lib as L
    with pytest.raises(L.ZktError):      # output_hash always runs a partial round (spec.rs:284-298): none is no schedule
        ctx.poseidon_load(w, half_full, 0, K.fr_to_mont(cv, rc[:2 * half_full * w]), K.fr_to_mont(cv, [x for row in mds for x in row]),
                          K.fr_to_mont(cv, [1])[0])
    ctx.poseidon_free(h)
    for d in (d_in, d_out, d_st):
        ctx.free(d)
    ctx.close()


def test_poseidon_states_feed_the_prover_without_leaving_the_device():
    """The witness of a Poseidon-heavy circuit never crosses PCIe: zkt_poseidon_hash_batch_dev writes every round's state
    into a device buffer that IS the head of the variable map of zkt_prove_inputs (wires_on_device = 1, prove.rs:49-55
    wire_evals on the device).  The circuit copies state words through gates and multiplies some of them; proof bytes ==
    the oracle's proof over the oracle's own Poseidon trace."""
    import zkt_plonk_amd as z
    from oracle import plonk as P
    cv = F.BN254
    p = cv.fr.p
    w = 3
    meta, rc, mds = _reference_params(w)
    half_full, partial, arity, batch = meta["full_rounds"] // 2, meta["partial_rounds"], meta["arity"], 4
    rounds = 2 * half_full + partial
    ins = [field_elems(p, 9100 + b, arity) for b in range(batch)]
    S = batch * (rounds + 1) * w
    # ---- oracle: the same trace as the first S variables, then the gates
    cs = P.ConstraintSystem(cv, [5, 6, 7], 8)
    out_vars = []
    for b in range(batch):
        hsh, trace = OP.permute(p, w, half_full, partial, rc, mds, meta["domain_tag"], ins[b])
        base = len(cs.values)
        for row in trace:
            for x in row:
                cs.assign_variable(x)
        out_vars.append(base + rounds * w + 1)                      # elements[1] after the last round = the hash
        assert cs.values[out_vars[-1]] == hsh
    assert len(cs.values) == S
    for v in range(0, S, 5):
        cs.arith_constrain(v, P.ZERO_VAR, v, q_l=1, q_o=-1)         # the state word, copied through a gate
    acc = out_vars[0]
    for v in out_vars[1:]:
        acc = cs.mul_gate(acc, v)                                    # product of the hashes (new variables past S)
    cs.set_variable_public(out_vars[0])
    cs.set_variable_public(acc)
    assert cs.check_satisfied()
    n = cs.circuit_bound()
    tau = 0xF00D
    srs = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    blinders = field_elems(p, 31, P.NUM_BLINDERS)
    want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders).serialize(cv)
    # ---- device: Poseidon states straight into the variable map
    ctx = z.Context(cv.name, 0)
    ctx.srs_load(srs)
    z.GpuProver(ctx, n.bit_length() - 1, {k: K.fr_to_mont(cv, pk.polys[k]) if pk.polys[k] else np.zeros((0, 4), np.uint64)
                                          for k in z.PK_ORDER})
    h = ctx.poseidon_load(w, half_full, partial, K.fr_to_mont(cv, rc), K.fr_to_mont(cv, [x for row in mds for x in row]),
                          K.fr_to_mont(cv, [meta["domain_tag"]])[0])
    n_vars = len(cs.values)
    d_vars, d_in, d_hash = ctx.alloc(n_vars * 32), ctx.alloc(batch * arity * 32), ctx.alloc(batch * 32)
    ctx.upload(d_in, K.fr_to_mont(cv, [x for row in ins for x in row]))
    ctx.poseidon_hash_batch_dev(h, d_in, batch, arity, d_hash, d_vars)           # states = variables [0, S)
    ctx.upload(d_vars + S * 32, K.fr_to_mont(cv, cs.values[S:]))                 # the few host-made variables behind them
    to_idx = lambda ws: np.array([0xFFFFFFFF if v == P.ZERO_VAR else v for v in ws], dtype=np.uint32)
    d_idx = []
    for ws in (cs.w_l, cs.w_r, cs.w_o):
        d = ctx.alloc(4 * len(ws))
        ctx.upload(d, to_idx(ws))
        d_idx.append(d)
    pi_pos = sorted(cs.pi)
    prep = ctx.prepare_vars_dev(d_vars, n_vars, d_idx[0], d_idx[1], d_idx[2], cs.n_gates, K.fr_to_mont(cv, cs.table), pi_pos,
                                K.fr_to_mont(cv, [cs.pi[k] for k in pi_pos]), K.fr_to_mont(cv, blinders))
    tr = z.seed_transcript(z.Transcript("merlin", "ZKT Plonk"), vk.n, vk.commits)
    assert ctx.prove_prepared(prep, tr) == want
    ctx.poseidon_free(h)
    ctx.close()
