"""Committed round-2 fixtures (tests/golden/vectors_r02.json, generator: tests/golden/make_golden_r02.py): the oracle
still reproduces them, and the product's host-only entry points (verifier, key-file readers) hit the same targets.
GPU targets of the same file are checked in the -m gpu tests (class transform, Poseidon, the 2^14 proof digest)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import fields as F, plonk as P, coracle as K, keyfile as KF, poseidon as OP
from helpers import field_elems, digest, unhex_point
import zkt_plonk_amd as z
from zkt_plonk_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g2():
    with open(os.path.join(ROOT, "tests", "golden", "vectors_r02.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381], ids=lambda c: c.name)
def test_verifier_pairs_and_key_files_of_the_golden_test_circuit(cv, g2, golden, tmp_path):
    g = golden[cv.name]
    tau = int(g["tau"], 16)
    cs = P.test_circuit(cv)
    n = cs.circuit_bound()
    srs = K.srs_mont(cv, tau, 4 * n + 1)
    be = K.CBackend(cv, srs)
    pk, epk, vk = P.setup(be, [None] * (4 * n + 1), cs, True)
    raw = bytes.fromhex(g["test_circuit"]["proof_bytes"])
    want = [[unhex_point(a), unhex_point(b)] for a, b in g2[cv.name]["verify_pairs"]]
    # oracle
    pairs = P.verify_prepare(cv, vk, P.proof_deserialize(cv, raw), P.new_seeded_transcript(cv, vk), [10, 2])
    assert [[L, W] for L, W in pairs] == want
    # product (host): same pairs from the same bytes
    tr = z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
    z.seed_transcript(tr, vk.n, vk.commits)
    got, inf = _lib.verify_prepare(cv.name, vk.n, K.points_to_mont(cv, [vk.commits[k] for k in z.PK_ORDER]),
                                   [vk.commits[k] is None for k in z.PK_ORDER], K.fr_to_mont(cv, vk.pi_roots),
                                   K.fr_to_mont(cv, [10, 2]), raw, srs[0], tr)
    pts = [None if inf[i] else K.points_from_mont(cv, got[i:i + 1])[0] for i in range(4)]
    assert [[pts[0], pts[1]], [pts[2], pts[3]]] == want
    # key files: the writers are frozen by their digests, the readers take the files back
    files = g2[cv.name]["key_files"]
    blobs = dict(ck=KF.committer_key_bytes(cv, K.points_from_mont(cv, srs)), pk=KF.prover_key_bytes(cv, pk),
                 vk=KF.verifier_key_bytes(cv, vk))
    for k, blob in blobs.items():
        assert hashlib.sha256(blob).hexdigest() == files[k + "_sha256"], k
        (tmp_path / k).write_bytes(blob)
    assert np.array_equal(_lib.keyfile_committer_key(str(tmp_path / "ck"), cv.name), srs)
    assert [K.fr_from_mont(cv, a) for a in _lib.keyfile_prover_key(str(tmp_path / "pk"), cv.name)] == [pk.polys[k] for k in P.PK_POLYS]


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381], ids=lambda c: c.name)
def test_oracle_reproduces_the_class_transform_and_poseidon_fixtures(cv, g2):
    p = cv.fr.p
    for e in g2[cv.name]["ntt_class"]:
        x = field_elems(p, e["seed"], e["in_len"])
        full = K.fr_from_mont(cv, K.ntt_mont(cv, e["log_big"], False, True, K.fr_to_mont(cv, x)))
        assert digest(full[e["cls"]::e["G"]]) == e["sha256"]
    q = g2[cv.name]["poseidon"]
    W = q["width"]
    rc = field_elems(p, q["rc_seed"], (2 * q["half_full"] + q["partial"]) * W)
    mds = [field_elems(p, q["mds_seed"] + i, W) for i in range(W)]
    h, _ = OP.permute(p, W, q["half_full"], q["partial"], rc, mds, q["domain_tag"], field_elems(p, q["input_seed"], 4))
    assert "%x" % h == q["hash"]


def test_config0_proof_digest(g2):
    """BASELINE.json configs[0]: the CPU oracle's proof of the bench workload at n = 2^14 (BN254) has the committed digest;
    tests/test_gpu_prove.py holds the GPU proof to the same digest."""
    import bench as B
    from oracle import fastplonk as FP
    cv = F.BN254
    e = g2["config0_bn254_2_14"]
    log_n, n = 14, 1 << 14
    tau = int(e["tau"], 16)
    circ = B.synthetic_circuit(B.FIELDS[cv.name], log_n)
    srs = K.srs_mont(cv, tau, n + 8)
    keys = FP.setup(cv, srs, log_n, {k: K.fr_to_mont(cv, circ["sel"][k]) for k in P.PK_POLYS})
    assert {k: unhex_point(v) for k, v in e["vk"].items()} == keys.commits
    vk = keys.verifier_key(cv, circ["pi"].keys())
    gts = circ["gates"]
    proof = FP.prove(cv, srs, keys, K.fr_to_mont(cv, circ["a"][:gts]), K.fr_to_mont(cv, circ["b"][:gts]),
                     K.fr_to_mont(cv, circ["c"][:gts]), K.fr_to_mont(cv, circ["table"]), circ["pi"],
                     P.new_seeded_transcript(cv, vk), field_elems(cv.fr.p, e["blinder_seed"], P.NUM_BLINDERS))
    assert hashlib.sha256(proof).hexdigest() == e["proof_sha256"]
