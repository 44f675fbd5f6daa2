#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle.

Run from the repo root:  python tests/golden/make_golden.py
The fixtures are DATA (inputs + expected outputs); the reference holds no golden vectors for
NTT / MSM / proof bytes, so these come from the Python big-integer oracle,
which is itself pinned by the reference's literal KATs and published constants
(see oracle/__init__.py).  Inputs are derived from splitmix64 so any test can regenerate them.
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import fields as F, curve as C, plonk as P  # noqa: E402
from oracle.ntt import Domain  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
M64 = (1 << 64) - 1


def splitmix64(seed):
    s = seed & M64
    while True:
        s = (s + 0x9E3779B97F4A7C15) & M64
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        yield z ^ (z >> 31)


def field_elems(p, seed, count):
    """Uniform-ish field elements: 4 splitmix64 words little-endian, reduced mod p."""
    g = splitmix64(seed)
    out = []
    for _ in range(count):
        v = 0
        for i in range(4):
            v |= next(g) << (64 * i)
        out.append(v % p)
    return out


TAU = 0x5EED5EED5EED5EED1234567890ABCDEF0FEDCBA9876543211357924680ACE


def digest(vals):
    import hashlib
    return hashlib.sha256(b"".join(int(v).to_bytes(32, "little") for v in vals)).hexdigest()


def hx(v):
    return "%x" % v


def pt(P_):
    return None if P_ is None else [hx(P_[0]), hx(P_[1])]


def main():
    out = {}
    for cv in (F.BN254, F.BLS12_381):
        p = cv.fr.p
        ntt = []
        for n, in_len in ((8, 8), (16, 11), (1024, 1024), (1024, 259)):
            x = field_elems(p, 0x5EED + n + in_len, in_len)
            d = Domain(cv.fr, n)
            res = dict(fft=d.fft(x), ifft=d.ifft(x), coset_fft=d.coset_fft(x), coset_ifft=d.coset_ifft(x))
            entry = dict(n=n, in_len=in_len, seed=0x5EED + n + in_len)
            if n <= 16:   # small cases carry the literal vectors, large ones a SHA-256 of the LE bytes
                entry["input"] = [hx(v) for v in x]
                entry.update({k: [hx(v) for v in vals] for k, vals in res.items()})
            entry.update({k + "_sha256": digest(vals) for k, vals in res.items()})
            ntt.append(entry)
        tau = TAU % p
        msm = []
        srs = C.srs_powers(cv, tau, 1000)
        for n in (1, 2, 31, 32, 33, 1000):
            sc = field_elems(p, 0xA11CE + n, n)
            if n >= 31:
                sc[0], sc[1], sc[2], sc[5] = 0, 1, p - 1, 0
            msm.append(dict(n=n, seed=0xA11CE + n, scalars=[hx(v) for v in sc] if n <= 33 else None,
                            result=pt(C.msm_naive(cv, srs[:n], sc))))
        # cancellation to infinity and an infinity base
        msm.append(dict(n=2, seed=None, scalars=[hx(tau), hx(p - 1)], result=pt(C.msm_naive(cv, srs[:2], [tau, p - 1]))))
        # full proof of the reference's TestCircuit (plonk.rs:144-218), n = 128
        cs = P.test_circuit(cv)
        n = cs.circuit_bound()
        be = P.Backend(cv)
        srs_p = srs[:4 * n + 1] if 4 * n + 1 <= len(srs) else C.srs_powers(cv, tau, 4 * n + 1)
        pk, epk, vk = P.setup(be, srs_p, cs, True)
        blinders = field_elems(p, 0xB11D, P.NUM_BLINDERS)
        trace = P.ProverTrace()
        proof = P.prove(be, srs_p, pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders, trace)
        assert P.verify(cv, tau, vk, proof, P.new_seeded_transcript(cv, vk), [10, 2])
        out[cv.name] = dict(
            tau=hx(tau), ntt=ntt, msm=msm,
            srs_first=[pt(x) for x in srs[:4]],
            test_circuit=dict(n=n, blinder_seed=0xB11D, public_inputs=["a", "2"],
                              vk={k: pt(v) for k, v in vk.commits.items()},
                              challenges={k: hx(v) for k, v in trace.challenges.items()},
                              evaluations=[hx(v) for v in proof.evaluations.as_list()],
                              commitments={k: pt(v) for k, v in proof.commits.items()},
                              aw_opening=pt(proof.aw_opening), saw_opening=pt(proof.saw_opening),
                              proof_bytes=proof.serialize(cv).hex()))
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", os.path.join(HERE, "vectors.json"), os.path.getsize(os.path.join(HERE, "vectors.json")), "bytes")


if __name__ == "__main__":
    main()
