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


def _gadget_params(cv, w):
    """PoseidonParams for the oracle composer + the device handle's arguments: the reference's BN254 x3 / x4 / x5 sets, or
    splitmix constants with a short schedule on BLS12-381 (its sets exist only as run-time generated constants)."""
    from oracle import composer as OC
    p = cv.fr.p
    if cv.name == "bn254":
        meta, rc, mds = _reference_params(w)
        return OC.PoseidonParams(p, w, meta["full_rounds"] // 2, meta["partial_rounds"], rc, mds, meta["domain_tag"])
    half_full, partial = 2, 5
    rc = field_elems(p, 40 + w, (2 * half_full + partial) * w)
    mds = [field_elems(p, 400 + 7 * w + i, w) for i in range(w)]
    return OC.PoseidonParams(p, w, half_full, partial, rc, mds)


def _load(ctx, cv, prm):
    return ctx.poseidon_load(prm.width, prm.half_full, prm.partial, K.fr_to_mont(cv, prm.rc),
                             K.fr_to_mont(cv, [x for row in prm.mds for x in row]), K.fr_to_mont(cv, [prm.domain_tag])[0])


@pytest.mark.parametrize("kernel", [1, 2], ids=["thread_per_hash", "lanes_per_hash"])
@pytest.mark.parametrize("cvname,w", [("bn254", 3), ("bn254", 4), ("bn254", 5), ("bls12_381", 2), ("bls12_381", 5), ("bls12_381", 8),
                                      ("bls12_381", 6)])
def test_gadget_witness_equals_the_composers_variables(cvname, w, kernel):
    """k_poseidon_gadget against the oracle's gate-by-gate restatement of PlonkSpecRef on the composer
    (oracle/composer.py: spec.rs:174-219 on constraint_system/arithmetic.rs:15-104): for every hash the
    2 half_full (3W + W^2) + partial (3 + W^2) values the composer assigns, in its allocation order -- 804 / 1288 / 1888
    on the reference's x3 / x4 / x5 -- for arities 0, 1 and W - 1, inputs given as values and as variable indices
    (Variable::Zero included), dense and scattered trace bases.  Both kernels: one thread per hash (large batches) and
    W^2 lanes per hash (a single proof's hashes: four dependent products per round instead of 28 - 40)."""
    import zkt_plonk_amd as z
    import zkt_plonk_amd._lib as L
    from oracle import composer as OC
    cv = F.CURVES[cvname]
    p = cv.fr.p
    prm = _gadget_params(cv, w)
    per = prm.gates_per_hash
    if cvname == "bn254":
        assert per == {3: 804, 4: 1288, 5: 1888}[w]
    ctx = z.Context(cv.name, 0)
    h = _load(ctx, cv, prm)
    assert ctx.poseidon_gadget_vars_per_hash(h) == per
    for arity in sorted({0, 1, w - 1}):
        batch = 150
        ins = [field_elems(p, 8100 + 3 * b + arity, arity) for b in range(batch)]
        ins[0] = [0] * arity
        ins[1] = [p - 1] * arity
        # (a) input values, dense traces behind a gap of 5 variables
        n_vars = 5 + batch * per
        d_vars, d_in, d_out = ctx.alloc(n_vars * 32), ctx.alloc(max(1, batch * arity) * 32), ctx.alloc(batch * 32)
        ctx.upload(d_vars, np.zeros((n_vars, 4), np.uint64))
        if arity:
            ctx.upload(d_in, K.fr_to_mont(cv, [x for row in ins for x in row]))
        ctx.poseidon_gadget_witness_dev(h, batch, arity, d_vars, n_vars, d_inputs=d_in if arity else 0, trace_base0=5,
                                        d_out_hashes=d_out, kernel=kernel)
        ctx.poseidon_gadget_check(h)
        got = K.fr_from_mont(cv, ctx.download(d_vars, (n_vars, 4)))
        hashes = K.fr_from_mont(cv, ctx.download(d_out, (batch, 4)))
        assert got[:5] == [0] * 5
        for b in (0, 1, 2, 77, batch - 1):
            want = OC.gadget_trace(prm, ins[b])
            assert got[5 + b * per:5 + (b + 1) * per] == want, (arity, b)
            assert hashes[b] == prm.native(ins[b]) == want[per - 1 - (w - 2) * w]
            # ... and the composer itself, gate by gate (the values it assigns while synthesising the hash)
            if b < 3:
                cs = OC.Composer(cv, [1], 8)
                vs = [cs.assign_variable(x) for x in ins[b]]
                out = OC.poseidon_hash(cs, prm, [cs.lt(v) for v in vs])
                assert cs.values[arity:] == want and cs.n_gates == per and cs.check_satisfied()
                assert out.var == arity + per - 1 - (w - 2) * w and (out.coeff, out.offset) == (1, 0)
        # (b) inputs as variable indices (every third hash's first input = Variable::Zero), traces at scattered bases
        if arity:
            n_vars = batch * arity + batch * (per + 3)
            vals = [x for row in ins for x in row]
            idx = np.arange(batch * arity, dtype=np.uint32).reshape(batch, arity)
            idx[::3, 0] = 0xFFFFFFFF
            bases = np.array([batch * arity + (batch - 1 - b) * (per + 3) + 2 for b in range(batch)], dtype=np.uint32)
            d_v2, d_idx, d_base = ctx.alloc(n_vars * 32), ctx.alloc(idx.nbytes), ctx.alloc(bases.nbytes)
            ctx.upload(d_v2, np.concatenate([K.fr_to_mont(cv, vals), np.zeros((n_vars - len(vals), 4), np.uint64)]))
            ctx.upload(d_idx, idx)
            ctx.upload(d_base, bases)
            ctx.poseidon_gadget_witness_dev(h, batch, arity, d_v2, n_vars, d_input_vars=d_idx, d_trace_base=d_base, kernel=kernel)
            ctx.poseidon_gadget_check(h)
            got2 = K.fr_from_mont(cv, ctx.download(d_v2, (n_vars, 4)))
            assert got2[:batch * arity] == vals
            for b in (0, 1, 2, 3, 76, batch - 1):
                row = [0 if (b % 3 == 0 and k == 0) else ins[b][k] for k in range(arity)]
                assert got2[int(bases[b]):int(bases[b]) + per] == OC.gadget_trace(prm, row), (arity, b)
                assert got2[int(bases[b]) - 2:int(bases[b])] == [0, 0]
            # an index outside the map: that hash is skipped, the flag is raised once and cleared
            bases[7] = n_vars - per + 1
            ctx.upload(d_base, bases)
            before = ctx.download(d_v2, (n_vars, 4))
            ctx.poseidon_gadget_witness_dev(h, batch, arity, d_v2, n_vars, d_input_vars=d_idx, d_trace_base=d_base, kernel=kernel)
            with pytest.raises(L.ZktError):
                ctx.poseidon_gadget_check(h)
            ctx.poseidon_gadget_check(h)
            assert np.array_equal(ctx.download(d_v2, (n_vars, 4)), before)      # the skipped hash wrote nothing, the others the same
            idx[9, arity - 1] = n_vars
            bases[7] = 0
            ctx.upload(d_base, bases)
            ctx.upload(d_idx, idx)
            ctx.poseidon_gadget_witness_dev(h, batch, arity, d_v2, n_vars, d_input_vars=d_idx, d_trace_base=d_base, kernel=kernel)
            with pytest.raises(L.ZktError):
                ctx.poseidon_gadget_check(h)
            for d in (d_v2, d_idx, d_base):
                ctx.free(d)
        with pytest.raises(L.ZktError):      # dense traces that do not fit the map are refused on the host
            ctx.poseidon_gadget_witness_dev(h, batch, arity, d_vars, batch * per - 1, d_inputs=d_in if arity else 0)
        if arity:
            with pytest.raises(L.ZktError):  # both input forms / neither
                ctx.poseidon_gadget_witness_dev(h, batch, arity, d_vars, n_vars)
        for d in (d_vars, d_in, d_out):
            ctx.free(d)
    with pytest.raises(L.ZktError):
        ctx.poseidon_gadget_witness_dev(h, 1, w, 1, 1, d_inputs=1)       # FullBuffer (spec.rs:253-257)
    # zkt_poseidon_gadget_validate: the two structural rules a launch must obey, checked on request
    per = ctx.poseidon_gadget_vars_per_hash(h)
    nv = 3 * per + 8
    d_map, d_b, d_i = ctx.alloc(nv * 32), ctx.alloc(8), ctx.alloc(8)
    for bases, ins, ok in (([8, 8 + per], [0, 1], True), ([8, 8 + per - 1], [0, 1], False),      # overlapping traces
                           ([8, 8 + per], [0, 8 + per + 3], False),                                # input made by this launch
                           ([8, 8 + 2 * per + 1], [0, 1], False)):                                 # a trace leaves the map
        ctx.upload(d_b, np.asarray(bases, dtype=np.uint32))
        ctx.upload(d_i, np.asarray(ins, dtype=np.uint32))
        if ok:
            ctx.poseidon_gadget_witness_dev(h, 2, 1, d_map, nv, d_input_vars=d_i, d_trace_base=d_b, validate_only=True)
        else:
            with pytest.raises(L.ZktError):
                ctx.poseidon_gadget_witness_dev(h, 2, 1, d_map, nv, d_input_vars=d_i, d_trace_base=d_b, validate_only=True)
    for d in (d_map, d_b, d_i):
        ctx.free(d)
    ctx.poseidon_free(h)
    ctx.close()


def _device_witness(ctx, cv, cs, gadget, corrupt=None):
    """The composer's variable map with every Poseidon trace produced ON THE DEVICE: the host uploads only the variables the
    gadget does not make (zeros where the traces go), one k_poseidon_gadget launch per dependency level
    (zkt_plonk_amd.PoseidonGadget: a hash fed by another hash's output waits for it) fills all the hashes' variables from
    their input variables' indices.  Returns the device pointer."""
    per = gadget.vars_per_hash
    n_vars = len(cs.values)
    host = list(cs.values)
    gadget.calls = []
    for base, ins in cs.hash_calls:
        assert all(co == 1 and off == 0 for (_, co, off) in ins), "gadget inputs are plain variables"
        host[base:base + per] = [0] * per
        out = gadget.hash(base, [0xFFFFFFFF if v == P.ZERO_VAR else v for (v, _, _) in ins])
        assert out == base + per - 1 - (gadget.width - 2) * gadget.width
    d_vars = ctx.alloc(n_vars * 32)
    ctx.upload(d_vars, K.fr_to_mont(cv, host))
    gadget.stage()
    assert gadget.fill(d_vars, n_vars) == len(gadget.levels())
    if corrupt is not None:
        v = K.fr_from_mont(cv, ctx.download(d_vars + corrupt * 32, (1, 4)))[0]
        ctx.upload(d_vars + corrupt * 32, K.fr_to_mont(cv, [(v + 1) % cv.fr.p]))
    return d_vars


def _gadget(z, ctx, cv, prm):
    return z.PoseidonGadget(ctx, prm.width, prm.half_full, prm.partial, K.fr_to_mont(cv, prm.rc),
                            K.fr_to_mont(cv, [x for row in prm.mds for x in row]), K.fr_to_mont(cv, [prm.domain_tag])[0])


def _prove_from_device_witness(ctx, cv, cs, d_vars, blinders, vk_n, vk_commits):
    import zkt_plonk_amd as z
    to_idx = lambda ws: np.array([0xFFFFFFFF if v == P.ZERO_VAR else v for v in ws], dtype=np.uint32)
    d_idx = []
    for ws in (cs.w_l, cs.w_r, cs.w_o):
        d = ctx.alloc(4 * len(ws))
        ctx.upload(d, to_idx(ws))
        d_idx.append(d)
    pi_pos = sorted(cs.pi)
    prep = ctx.prepare_vars_dev(d_vars, len(cs.values), d_idx[0], d_idx[1], d_idx[2], cs.n_gates, K.fr_to_mont(cv, cs.table),
                                pi_pos, K.fr_to_mont(cv, [cs.pi[k] for k in pi_pos]), K.fr_to_mont(cv, blinders))
    tr = z.seed_transcript(z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=8 * cv.fq.limbs64), vk_n, vk_commits)
    try:
        return ctx.prove_prepared(prep, tr)
    finally:
        for d in d_idx:
            ctx.free(d)


from oracle import plonk as P


@pytest.mark.parametrize("w", [3, 4, 5])
def test_poseidon_gadget_circuit_proves_from_the_device_made_witness(w):
    """A circuit of REAL Poseidon gadgets on the reference's x3 / x4 / x5 sets, built by the oracle composer gate by gate
    (chained hashes of different arities, the last one public): its witness is produced on the device -- the host
    uploads the free variables only, k_poseidon_gadget fills every gate output -- and goes to zkt_prove as the variable
    map + the composer's w_l / w_r / w_o (wires_on_device = 1).  Proof bytes == the oracle's proof over the composer's own
    values.  One flipped intermediate (an x^4 in a partial round; a running MDS sum) makes the proof fail: every gadget
    variable is constrained."""
    import zkt_plonk_amd as z
    import zkt_plonk_amd._lib as L
    from oracle import composer as OC
    cv = F.BN254
    p = cv.fr.p
    prm = _gadget_params(cv, w)
    per = prm.gates_per_hash
    cs = OC.Composer(cv, [5, 6, 7], 8)
    free = [cs.assign_variable(x) for x in field_elems(p, 9200 + w, 2 * (w - 1) + 2)]
    h1 = OC.poseidon_hash(cs, prm, [cs.lt(v) for v in free[:w - 1]])            # full arity
    h2 = OC.poseidon_hash(cs, prm, [cs.lt(free[w - 1])])                        # one input
    h3 = OC.poseidon_hash(cs, prm, [])                                          # none
    h4 = OC.poseidon_hash(cs, prm, [cs.lt(v) for v in free[w:2 * w - 1]][:w - 1])
    prod = cs.mul_gate(h1, h2)
    s = cs.add_gate(cs.lt(prod), h3)
    cs.set_variable_public(h4)
    cs.set_variable_public(cs.lt(s))
    assert cs.n_gates == 4 * per + 4 and len(cs.hash_calls) == 4 and cs.check_satisfied()
    n = cs.circuit_bound()
    tau = 0xF00D
    srs = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    blinders = field_elems(p, 31 + w, P.NUM_BLINDERS)
    want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders).serialize(cv)
    ctx = z.Context(cv.name, 0)
    ctx.srs_load(srs)
    z.GpuProver(ctx, n.bit_length() - 1, {k: K.fr_to_mont(cv, pk.polys[k]) if pk.polys[k] else np.zeros((0, 4), np.uint64)
                                          for k in z.PK_ORDER})
    g = _gadget(z, ctx, cv, prm)
    assert g.vars_per_hash == per
    d_vars = _device_witness(ctx, cv, cs, g)
    assert K.fr_from_mont(cv, ctx.download(d_vars, (len(cs.values), 4))) == cs.values      # the whole map, value by value
    got = _prove_from_device_witness(ctx, cv, cs, d_vars, blinders, vk.n, vk.commits)
    assert got == want
    assert P.verify(cv, tau, vk, P.proof_deserialize(cv, got), P.new_seeded_transcript(cv, vk), [cs.pi[k] for k in sorted(cs.pi)])
    ctx.free(d_vars)
    # a wrong intermediate cannot prove: x^4 of the s-box of the first partial round of hash 2, and a running sum of hash 4
    W = w
    x4_partial = cs.hash_calls[1][0] + prm.half_full * (3 * W + W * W) + 1
    mds_sum = cs.hash_calls[3][0] + 3 * W + W + 1
    for bad in (x4_partial, mds_sum):
        d_bad = _device_witness(ctx, cv, cs, g, corrupt=bad)
        with pytest.raises(L.ZktError) as e:
            _prove_from_device_witness(ctx, cv, cs, d_bad, blinders, vk.n, vk.commits)
        assert e.value.code == 9   # ZKT_ERR_QUOTIENT_TOO_SHORT: the circuit is not satisfied
        ctx.free(d_bad)
    g.close()
    ctx.close()


def test_config0_the_withdraw_circuit_itself_at_2_14():
    """BASELINE.json configs[0] literally: WithdrawCircuit (circuits/src/withdraw.rs:57-150) on BN254 with Poseidon x4, one
    note, HEIGHT 7 = 15 640 gates -> n = 2^14, synthesised by the oracle composer statement by statement (Poseidon
    commitments / nullifier / leaves, the Merkle path gadget, the identifier lookup, the 64-bit range and balance rows; 5
    public inputs in the CLI's order).  The 12 hashes' 15 456 variables are made on the device, the other 148 come from
    the host; proof bytes == the CPU oracle's array prover, the verifier accepts with the public inputs the CLI would
    pass (bin/src/main.rs:263-269)."""
    import zkt_plonk_amd as z
    from oracle import composer as OC, fastplonk as FP
    cv = F.BN254
    p = cv.fr.p
    prm = _gadget_params(cv, 4)
    cs, public_inputs = OC.withdraw_instance(cv, prm, inputs=1, height=7, seed=11)
    assert cs.n_gates == OC.withdraw_gate_count(prm, 1, 7) == 15640 and cs.check_satisfied()
    assert len(cs.hash_calls) == 3 + 7 + 2
    n = cs.circuit_bound()
    assert n == 1 << 14
    tau = 0x5EED5EED1234567890ABCDEF % p
    srs = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs)
    evals = {k: K.fr_to_mont(cv, v) for k, v in P.setup_evals(be, cs).items()}
    keys = FP.setup(cv, srs, 14, evals)
    vk = keys.verifier_key(cv, cs.pi.keys())
    assert [cs.pi[k] for k in sorted(cs.pi)] == public_inputs
    a, b, c = cs.wire_evals(cs.n_gates)
    blinders = field_elems(p, 1414, P.NUM_BLINDERS)
    want = FP.prove(cv, srs, keys, K.fr_to_mont(cv, a), K.fr_to_mont(cv, b), K.fr_to_mont(cv, c), K.fr_to_mont(cv, cs.table),
                    dict(cs.pi), P.new_seeded_transcript(cv, vk), blinders)
    ctx = z.Context(cv.name, 0)
    ctx.srs_load(srs)
    prover, commits = z.GpuProver.setup(ctx, 14, evals)
    g = _gadget(z, ctx, cv, prm)
    d_vars = _device_witness(ctx, cv, cs, g)
    assert [len(l) for l in g.levels()] == [10, 2]        # the two leaf hashes take a commitment hash: second launch
    assert K.fr_from_mont(cv, ctx.download(d_vars, (len(cs.values), 4))) == cs.values
    got = _prove_from_device_witness(ctx, cv, cs, d_vars, blinders, vk.n, vk.commits)
    assert got == want and len(got) == 802
    assert P.verify(cv, tau, vk, P.proof_deserialize(cv, got), P.new_seeded_transcript(cv, vk), public_inputs)
    bad = list(public_inputs)
    bad[1] = (bad[1] + 1) % p                                  # another nullifier
    assert not P.verify(cv, tau, vk, P.proof_deserialize(cv, got), P.new_seeded_transcript(cv, vk), bad)
    ctx.free(d_vars)
    g.close()
    ctx.close()


@pytest.mark.parametrize("cvname,width,inputs,height,log_n", [("bn254", 4, 3, 48, 18), ("bls12_381", 5, 1, 3, 14)],
                         ids=["bn254-cli-default-2^18", "bls12_381-2^14"])
def test_withdraw_circuit_at_the_clis_default_size(cvname, width, inputs, height, log_n):
    """The circuit the reference's binary builds with its DEFAULT features (bin/Cargo.toml:25: bn254, height-48, notes-3,
    poseidon-bn254-x4: 200 793 gates, n = 2^18, 7 public inputs), and a BLS12-381 instance of the same circuit (generated
    Poseidon constants: the reference ships BN254 tables only; width 5, one note, HEIGHT 3: n = 2^14).  Synthesised by the
    oracle composer; the hashes' variables (155 x 1288 on BN254) come from the device in one launch per dependency level,
    through zkt_plonk_amd.PoseidonGadget; proof bytes == the CPU oracle's array prover; the oracle's verifier accepts with
    the CLI's public inputs and rejects a wrong root."""
    import sys, os
    import zkt_plonk_amd as z
    from oracle import composer as OC, fastplonk as FP
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import withdraw_workload as WW
    cv = F.CURVES[cvname]
    p = cv.fr.p
    if cvname == "bn254":
        prm = _gadget_params(cv, width)
    else:
        hs = WW.synthetic_hasher(p, width)
        prm = OC.PoseidonParams(p, width, hs.half_full, hs.partial, hs.rc, hs.mds, hs.tag)
    cs, public_inputs = OC.withdraw_instance(cv, prm, inputs=inputs, height=height, seed=18)
    assert cs.n_gates == OC.withdraw_gate_count(prm, inputs, height) and cs.check_satisfied()
    n = cs.circuit_bound()
    assert n == 1 << log_n and len(public_inputs) == 4 + inputs
    tau = 0x5EED5EED1234567890ABCDEF % p
    ctx = z.Context(cv.name, 0)
    ctx.srs_generate(tau, n + 8)
    srs = ctx.srs_download(0, n + 8)
    be = K.CBackend(cv, srs)
    evals = {k: K.fr_to_mont(cv, v) for k, v in P.setup_evals(be, cs).items()}
    keys = FP.setup(cv, srs, log_n, evals, commitments=False)
    prover, commits = z.GpuProver.setup(ctx, log_n, evals)
    L = cv.fq.limbs64
    rinv = pow(1 << (64 * L), -1, cv.fq.p)
    keys.commits = {name: (None if inf else (sum(int(v) << (64 * i) for i, v in enumerate(xy[:L])) * rinv % cv.fq.p,
                                             sum(int(v) << (64 * i) for i, v in enumerate(xy[L:])) * rinv % cv.fq.p))
                    for name, (xy, inf) in commits.items()}
    assert FP.commit(cv, srs, keys.pk["q_m"]) == keys.commits["q_m"] and FP.commit(cv, srs, keys.pk["sigma2"]) == keys.commits["sigma2"]
    vk = keys.verifier_key(cv, cs.pi.keys())
    a, b, c = cs.wire_evals(cs.n_gates)
    blinders = field_elems(p, 1800 + log_n, P.NUM_BLINDERS)
    want = FP.prove(cv, srs, keys, K.fr_to_mont(cv, a), K.fr_to_mont(cv, b), K.fr_to_mont(cv, c), K.fr_to_mont(cv, cs.table),
                    dict(cs.pi), P.new_seeded_transcript(cv, vk), blinders)
    g = _gadget(z, ctx, cv, prm)
    d_vars = _device_witness(ctx, cv, cs, g)
    assert [len(x) for x in g.levels()] == [inputs * (2 + height) + 1, inputs + 1]
    assert K.fr_from_mont(cv, ctx.download(d_vars, (len(cs.values), 4))) == cs.values
    got = _prove_from_device_witness(ctx, cv, cs, d_vars, blinders, vk.n, vk.commits)
    assert got == want and len(got) == (802 if cvname == "bn254" else 1010)
    assert P.verify(cv, tau, vk, P.proof_deserialize(cv, got), P.new_seeded_transcript(cv, vk), public_inputs)
    assert not P.verify(cv, tau, vk, P.proof_deserialize(cv, got), P.new_seeded_transcript(cv, vk), [public_inputs[0] + 1] + public_inputs[1:])
    ctx.free(d_vars)
    g.close()
    ctx.close()


def test_config3_the_withdraw_circuit_itself_at_2_20():
    """BASELINE.json configs[3] literally: WithdrawCircuit<Fr, u64, _, Bn254x5, INPUTS = 8, HEIGHT = 64>
    (circuits/src/withdraw.rs:57-150; 1 019 498 gates, n = 2^20, 12 public inputs, 538 Poseidon gadgets of 1888 gates) on
    BN254 -- the workload of bench.py, here under pytest.  Rows laid out by tools/withdraw_workload.py (pinned row for row
    against the oracle composer at smaller shapes in tests/test_withdraw_workload.py), the hashes' 1 015 744 variables made on
    the device (PoseidonGadget.fill), the prover handed the variable map and the wire index vectors in HBM.  Checked: the
    wire values equal the oracle composer's own synthesis of the same withdrawal (oracle/composer.py, gate by gate), which
    satisfies every gate; the proof bytes equal the CPU oracle's array prover (oracle/fastplonk.py) on the same SRS, witness
    and blinders."""
    import os
    import sys
    import zkt_plonk_amd as z
    from oracle import composer as OC, fastplonk as FP
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import withdraw_workload as WW
    cv = F.BN254
    p = cv.fr.p
    log_n, width, inputs, height = 20, 5, 8, 64
    n = 1 << log_n
    hs = WW.reference_hasher(p, width)
    inst = WW.make_instance(hs, inputs, height, seed=0x5EED)
    lay = WW.layout(hs, inst)
    gates, n_vars = lay.n_gates, len(lay.values)
    assert gates == 1019498 and len(lay.pi) == 12 and len(lay.hash_calls) == 538 and hs.per_hash == 1888
    tau = 0x5EED5EED1234567890ABCDEF % p
    ctx = z.Context(cv.name, 0)
    try:
        ctx.srs_generate(tau, n + 8)
        srs = ctx.srs_download(0, n + 8)
        sel = WW.setup_vectors(lay, log_n, 5)
        evals = {name: K.fr_to_mont(cv, sel[name]) for name in z.PK_ORDER}
        del sel
        prover, commits = z.GpuProver.setup(ctx, log_n, evals)
        L = cv.fq.limbs64
        rinv = pow(1 << (64 * L), -1, cv.fq.p)
        vk_pts = {name: (None if inf else (sum(int(v) << (64 * i) for i, v in enumerate(xy[:L])) * rinv % cv.fq.p,
                                           sum(int(v) << (64 * i) for i, v in enumerate(xy[L:])) * rinv % cv.fq.p))
                  for name, (xy, inf) in commits.items()}
        to_mont = lambda vals: K.fr_to_mont(cv, vals)
        gadget = z.PoseidonGadget(ctx, hs.width, hs.half_full, hs.partial, to_mont(hs.rc),
                                  to_mont([x for row in hs.mds for x in row]), to_mont([hs.tag])[0])
        assert gadget.vars_per_hash == hs.per_hash
        for base, ins in lay.hash_calls:
            gadget.hash(base, ins)
        gadget.stage()
        d_vars = ctx.alloc(n_vars * 32)
        ctx.upload(d_vars, to_mont(lay.values))                 # zeros where the device writes
        assert gadget.fill(d_vars, n_vars) == 2                 # the leaf hashes wait for the commitment hashes
        idx = [np.asarray(w, dtype=np.uint32) for w in lay.w]
        d_idx = []
        for x in idx:
            d = ctx.alloc(4 * len(x))
            ctx.upload(d, x)
            d_idx.append(d)
        full = np.concatenate([ctx.download(d_vars, (n_vars, 4)), np.zeros((1, 4), np.uint64)])
        wires = [full[np.where(x == WW.ZERO, n_vars, x)] for x in idx]
        # the checker's own synthesis of the same withdrawal: satisfied, and the same wire values
        prm = OC.PoseidonParams(p, hs.width, hs.half_full, hs.partial, hs.rc, hs.mds, hs.tag)
        cs = OC.Composer(cv, inst["ident_set"], 1024)
        OC.withdraw_synthesize(cs, prm, inst["secrets"], inst["identifiers"], inst["amounts"], inst["poes"], inst["root"],
                               inst["new_secret"], inst["new_identifier"], inst["withdraw_amount"])
        assert cs.n_gates == gates == OC.withdraw_gate_count(prm, inputs, height) and cs.check_satisfied()
        for got_w, want_w in zip(wires, cs.wire_evals(cs.n_gates)):
            assert np.array_equal(got_w, K.fr_to_mont(cv, want_w))
        assert dict(cs.pi) == dict(lay.pi)
        del cs
        # proof bytes against the CPU oracle
        table = to_mont(inst["ident_set"])
        pi_pos = sorted(lay.pi)
        blinders = field_elems(p, 2020, P.NUM_BLINDERS)
        prep = ctx.prepare_vars_dev(d_vars, n_vars, d_idx[0], d_idx[1], d_idx[2], gates, table, pi_pos,
                                    to_mont([lay.pi[k] for k in pi_pos]), to_mont(blinders))
        tr = z.seed_transcript(z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=8 * L), n, vk_pts)
        got = ctx.prove_prepared(prep, tr)
        assert ctx.lagrange_info()["log_n"] == log_n           # h1 / h2 / z2 went through the Lagrange-basis table
        keys = FP.setup(cv, srs, log_n, evals, commitments=False)
        keys.commits = dict(vk_pts)
        vk = keys.verifier_key(cv, lay.pi.keys())
        want = FP.prove(cv, srs, keys, wires[0], wires[1], wires[2], table, dict(lay.pi), P.new_seeded_transcript(cv, vk), blinders)
        assert got == want and len(got) == 802
        gadget.close()
    finally:
        ctx.close()
