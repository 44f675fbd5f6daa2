"""f3: the batched Poseidon permutation (witness synthesis for Poseidon-heavy circuits) against the
oracle's restatement of plonk-hashing's native spec.  "Parity unpinned" by the reference (constants are generated at
run time, no known answers in the tree): arbitrary constants, every width the withdraw instances use (x3, x4, x5)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import fields as F, coracle as K, poseidon as OP
from helpers import field_elems


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381], ids=lambda c: c.name)
@pytest.mark.parametrize("width,half_full,partial", [(3, 4, 57), (4, 4, 56), (5, 4, 60), (2, 1, 0), (8, 2, 3)])
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
