"""f2: the readers for the reference CLI's --ck / --pk / --vk / --epk files against the oracle's writers
(oracle/keyfile.py).  "Parity unpinned": the reference holds no key file; what is pinned is the round trip and the
rejection of malformed files.  Host only."""
import os

import numpy as np
import pytest

from oracle import fields as F, plonk as P, coracle as K, keyfile as KF
from zkt_plonk_amd import _lib


def _setup(cv):
    cs = P.test_circuit(cv)
    n = cs.circuit_bound()
    srs = K.srs_mont(cv, 0xF11E, 4 * n + 1)
    be = K.CBackend(cv, srs)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, False)
    return cs, n, srs, pk, vk


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381], ids=lambda c: c.name)
def test_committer_prover_and_verifier_key_round_trip(cv, tmp_path):
    cs, n, srs, pk, vk = _setup(cv)
    pts = K.points_from_mont(cv, srs)
    pts[3] = None                                            # an identity among the powers must survive the trip
    ck = tmp_path / "ck.bin"
    ck.write_bytes(KF.committer_key_bytes(cv, pts))
    got = _lib.keyfile_committer_key(str(ck), cv.name)
    want = srs.copy()
    want[3] = 0
    assert np.array_equal(got, want)
    part = _lib.keyfile_committer_key(str(ck), cv.name, max_powers=n + 8)      # the prover's share: a prefix, no full read
    assert np.array_equal(part, want[:n + 8])
    pkf = tmp_path / "pk.bin"
    pkf.write_bytes(KF.prover_key_bytes(cv, pk))
    polys = _lib.keyfile_prover_key(str(pkf), cv.name)
    for k, name in enumerate(P.PK_POLYS):
        assert K.fr_from_mont(cv, polys[k]) == pk.polys[name], name
    vkf = tmp_path / "vk.bin"
    vkf.write_bytes(KF.verifier_key_bytes(cv, vk))
    n_, roots, commits, inf = _lib.keyfile_verifier_key(str(vkf), cv.name)
    assert n_ == vk.n and K.fr_from_mont(cv, roots) == vk.pi_roots
    for k, name in enumerate(P.PK_POLYS):
        want_pt = vk.commits[name]
        assert (None if inf[k] else K.points_from_mont(cv, commits[k:k + 1])[0]) == want_pt, name


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381], ids=lambda c: c.name)
def test_extended_prover_key_round_trip_and_rejections(cv, tmp_path):
    """--epk (bin/src/main.rs:108-109): the seventeen vectors of ExtendedProverKey<F> (keys/mod.rs:148-174) come back one at
    a time as the oracle's extend_prover_key made them; a truncated file, trailing bytes, a length beyond the file and a
    value that is not below the modulus are refused."""
    cs = P.test_circuit(cv)
    n = cs.circuit_bound()
    be = K.CBackend(cv, K.srs_mont(cv, 0xF11E, n + 8))
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    vecs = KF.extended_prover_key_vectors(epk)
    blob = KF.extended_prover_key_bytes(cv, epk)
    assert len(blob) == 17 * 8 + 32 * (13 * 4 * n + 4 * n)
    f = tmp_path / "epk.bin"
    f.write_bytes(blob)
    lens, none = _lib.keyfile_extended_prover_key(str(f), cv.name)
    assert none is None and lens == [len(vecs[k]) for k in KF.EPK_ORDER]
    for i, name in enumerate(KF.EPK_ORDER):
        lens, got = _lib.keyfile_extended_prover_key(str(f), cv.name, i)
        assert K.fr_from_mont(cv, got) == [v % cv.fr.p for v in vecs[name]], name
    # zh_coset takes four values, x_coset is g * w^i: the writer's input is what keys/mod.rs:110-117 says
    assert len(set(vecs["zh_coset"])) == 4
    bad = tmp_path / "bad.bin"
    for mutate in (lambda b: b[:-1], lambda b: b + b"\x00", lambda b: b"\xff" * 8 + b[8:],
                   lambda b: b[:8] + b"\xff" * 32 + b[40:]):
        bad.write_bytes(mutate(blob))
        with pytest.raises(_lib.ZktError):
            _lib.keyfile_extended_prover_key(str(bad), cv.name, 0)
    with pytest.raises(_lib.ZktError):
        _lib.keyfile_extended_prover_key(str(tmp_path / "missing.bin"), cv.name)


def test_malformed_key_files_are_rejected(tmp_path):
    cv = F.BN254
    cs, n, srs, pk, vk = _setup(cv)
    good = KF.prover_key_bytes(cv, pk)
    f = tmp_path / "bad.bin"
    for mutate in (lambda b: b[:-1],                                     # truncated
                   lambda b: b + b"\x00",                                # trailing garbage
                   lambda b: b[:8] + b"\xff" * 8 + b[16:],               # absurd length
                   lambda b: b[:len(b) - 1] + b"\x07"):                  # Option tag that is neither 0 nor 1
        f.write_bytes(mutate(good))
        with pytest.raises(_lib.ZktError):
            _lib.keyfile_prover_key(str(f), cv.name)
    # a coefficient that is not below the modulus
    label = len("q_m")
    off = 8 + label + 8
    f.write_bytes(good[:off] + b"\xff" * 32 + good[off + 32:])
    with pytest.raises(_lib.ZktError):
        _lib.keyfile_prover_key(str(f), cv.name)
    ckb = KF.committer_key_bytes(cv, K.points_from_mont(cv, srs[:16]))
    f.write_bytes(ckb[:40])
    with pytest.raises(_lib.ZktError):
        _lib.keyfile_committer_key(str(f), cv.name)
    with pytest.raises(_lib.ZktError):
        _lib.keyfile_committer_key(str(tmp_path / "missing.bin"), cv.name)


def test_readers_and_verifier_survive_mutated_inputs(tmp_path):
    """Parsers of untrusted bytes (key files, proofs): a few hundred random truncations / byte flips / length-field
    edits must end in a clean result or a ZktError, never in a crash or an out-of-bounds access (this test also runs
    under AddressSanitizer + UBSan, tests/test_host_sanitize.py)."""
    import random
    import zkt_plonk_amd as z
    cv = F.BN254
    cs, n, srs, pk, vk = _setup(cv)
    rnd = random.Random(1234)
    be0 = K.CBackend(cv, srs[:n + 8])
    pk0, epk0, vk0 = P.setup(be0, [None] * (n + 8), P.synthetic_circuit(cv, 5, 2, seed=1), True)
    blobs = {"pk": KF.prover_key_bytes(cv, pk), "vk": KF.verifier_key_bytes(cv, vk),
             "ck": KF.committer_key_bytes(cv, K.points_from_mont(cv, srs[:40])), "epk": KF.extended_prover_key_bytes(cv, epk0)}
    readers = {"pk": lambda p: _lib.keyfile_prover_key(p, cv.name), "vk": lambda p: _lib.keyfile_verifier_key(p, cv.name),
               "ck": lambda p: _lib.keyfile_committer_key(p, cv.name, max_powers=rnd.choice([0, 1, 7, 40, 1000])),
               "epk": lambda p: _lib.keyfile_extended_prover_key(p, cv.name, rnd.randrange(-1, 17))}

    def mutate(b):
        b = bytearray(b)
        kind = rnd.randrange(4)
        if kind == 0 and len(b) > 1:
            del b[rnd.randrange(len(b)):]
        elif kind == 1:
            for _ in range(rnd.randrange(1, 6)):
                b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        elif kind == 2:                      # stomp on a plausible length field
            at = rnd.choice([0, 8, 16, 24]) if len(b) > 32 else 0
            b[at:at + 8] = rnd.choice([0, 1, 2 ** 32, 2 ** 63, 2 ** 64 - 1, len(b)]).to_bytes(8, "little")
        else:
            b += bytes(rnd.randrange(1, 40))
        return bytes(b)

    outcomes = {"ok": 0, "err": 0}
    f = tmp_path / "m.bin"
    for i in range(240):
        name = ("pk", "vk", "ck", "epk")[i % 4]
        f.write_bytes(mutate(blobs[name]))
        try:
            readers[name](str(f))
            outcomes["ok"] += 1
        except _lib.ZktError:
            outcomes["err"] += 1
    assert outcomes["err"] > 100 and outcomes["ok"] + outcomes["err"] == 240
    # proofs through the verifier
    from helpers import field_elems
    be = K.CBackend(cv, K.srs_mont(cv, 0xF11E, n + 8))
    pk2, epk2, vk2 = P.setup(be, [None] * (n + 8), cs, True)
    proof = P.prove(be, [None] * (n + 8), pk2, epk2, vk2, cs, P.new_seeded_transcript(cv, vk2),
                    field_elems(cv.fr.p, 3, P.NUM_BLINDERS)).serialize(cv)
    commits = K.points_to_mont(cv, [vk2.commits[k] for k in z.PK_ORDER])
    inf = [vk2.commits[k] is None for k in z.PK_ORDER]
    pis = [cs.pi[k] for k in sorted(cs.pi)]
    seen_err = 0
    for i in range(120):
        tr = z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=32)
        z.seed_transcript(tr, vk2.n, vk2.commits)
        try:
            _lib.verify_prepare(cv.name, vk2.n, commits, inf, K.fr_to_mont(cv, vk2.pi_roots), K.fr_to_mont(cv, pis),
                                mutate(proof), be.srs_arr[0], tr)
        except _lib.ZktError:
            seen_err += 1
    assert seen_err > 30
