"""f2: the readers for the reference CLI's --ck / --pk / --vk files against the oracle's writers
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
