#!/usr/bin/env python3
"""Round-2 additions to the committed golden fixtures (tests/golden/vectors_r02.json), generated from the CPU oracle:
the class transform of a sharded coset NTT, the verifier's (L, W) pairs for the golden TestCircuit proof, digests of the
key files the oracle writes for that circuit, one Poseidon permutation, and the digest of the bench workload's proof at
n = 2^14 (BASELINE.json configs[0]).  DATA only; none of it is pinned by the reference, it freezes the
oracle against itself and gives the GPU tests fixed targets.

Run from the repo root:  python tests/golden/make_golden_r02.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import fields as F, plonk as P, coracle as K, keyfile as KF, poseidon as OP, fastplonk as FP  # noqa: E402
from helpers import field_elems, digest  # noqa: E402
import bench as B  # noqa: E402


def hx(v):
    return "%x" % v


def pt(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def main():
    old = json.load(open(os.path.join(HERE, "vectors.json")))
    out = {}
    for cv in (F.BN254, F.BLS12_381):
        p = cv.fr.p
        g = old[cv.name]
        tau = int(g["tau"], 16)
        # class transform: outputs of the 2^12 coset transform with index = cls mod G, input of 1032 coefficients
        x = field_elems(p, 0xC1A55, 1032)
        full = K.fr_from_mont(cv, K.ntt_mont(cv, 12, False, True, K.fr_to_mont(cv, x)))
        cls = [dict(log_big=12, G=G, cls=c, in_len=1032, seed=0xC1A55, sha256=digest(full[c::G]))
               for G, c in ((2, 1), (4, 3), (8, 5))]
        # verifier pairs and key files of the golden TestCircuit proof
        cs = P.test_circuit(cv)
        n = cs.circuit_bound()
        srs = K.srs_mont(cv, tau, 4 * n + 1)
        be = K.CBackend(cv, srs)
        pk, epk, vk = P.setup(be, [None] * (4 * n + 1), cs, True)
        proof = P.proof_deserialize(cv, bytes.fromhex(g["test_circuit"]["proof_bytes"]))
        pairs = P.verify_prepare(cv, vk, proof, P.new_seeded_transcript(cv, vk), [10, 2])
        files = dict(ck_sha256=hashlib.sha256(KF.committer_key_bytes(cv, K.points_from_mont(cv, srs))).hexdigest(),
                     pk_sha256=hashlib.sha256(KF.prover_key_bytes(cv, pk)).hexdigest(),
                     vk_sha256=hashlib.sha256(KF.verifier_key_bytes(cv, vk)).hexdigest())
        # one Poseidon permutation (width 5: the x4 hasher), arbitrary constants from splitmix64
        W, HF, PR = 5, 4, 60
        rc = field_elems(p, 0x905E1D, (2 * HF + PR) * W)
        mds = [field_elems(p, 0x3D5 + i, W) for i in range(W)]
        ins = field_elems(p, 0x1A9, 4)
        h, _ = OP.permute(p, W, HF, PR, rc, mds, 15, ins)
        out[cv.name] = dict(ntt_class=cls, verify_pairs=[[pt(L), pt(Wt)] for L, Wt in pairs], key_files=files,
                            poseidon=dict(width=W, half_full=HF, partial=PR, rc_seed=0x905E1D, mds_seed=0x3D5, domain_tag=15,
                                          input_seed=0x1A9, hash=hx(h)))
    # bench workload at n = 2^14 on BN254 (configs[0]): digest of the proof bytes
    cv = F.BN254
    log_n, n = 14, 1 << 14
    tau = 0x5EED5EED1234567890ABCDEF % cv.fr.p
    circ = B.synthetic_circuit(B.FIELDS[cv.name], log_n)
    srs = K.srs_mont(cv, tau, n + 8)
    keys = FP.setup(cv, srs, log_n, {k: K.fr_to_mont(cv, circ["sel"][k]) for k in P.PK_POLYS})
    vk = keys.verifier_key(cv, circ["pi"].keys())
    gts = circ["gates"]
    proof = FP.prove(cv, srs, keys, K.fr_to_mont(cv, circ["a"][:gts]), K.fr_to_mont(cv, circ["b"][:gts]),
                     K.fr_to_mont(cv, circ["c"][:gts]), K.fr_to_mont(cv, circ["table"]), circ["pi"],
                     P.new_seeded_transcript(cv, vk), field_elems(cv.fr.p, 2034, P.NUM_BLINDERS))
    out["config0_bn254_2_14"] = dict(tau=hx(tau), blinder_seed=2034, proof_sha256=hashlib.sha256(proof).hexdigest(),
                                     vk={k: pt(v) for k, v in keys.commits.items()})
    path = os.path.join(HERE, "vectors_r02.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
