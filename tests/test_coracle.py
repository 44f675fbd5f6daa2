"""Pins the C++ part of the oracle against the Python big-integer oracle (CPU)."""
import numpy as np
import pytest

from oracle import fields as F, curve as C, plonk as P, coracle as K
from oracle.ntt import Domain
from helpers import field_elems, digest, unhex_point


def test_parameter_tables():
    for i, f in enumerate((F.BN254_FR, F.BN254_FQ, F.BLS12_381_FR, F.BLS12_381_FQ)):
        assert K.params(i) == dict(p=f.p, inv=f.inv64, r=f.R, r2=f.R2)


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381])
def test_ntt_matches_python_and_golden(cv, golden):
    p = cv.fr.p
    for e in golden[cv.name]["ntt"]:
        log_n = e["n"].bit_length() - 1
        x = K.fr_to_mont(cv, field_elems(p, e["seed"], e["in_len"]))
        for k, inv, cos in (("fft", 0, 0), ("ifft", 1, 0), ("coset_fft", 0, 1), ("coset_ifft", 1, 1)):
            y = K.fr_from_mont(cv, K.ntt_mont(cv, log_n, inv, cos, x))
            assert digest(y) == e[k + "_sha256"]
    for log_n in (0, 1, 2, 5):
        d = Domain(cv.fr, 1 << log_n)
        x = field_elems(p, 4242 + log_n, 1 << log_n)
        assert K.fr_from_mont(cv, K.ntt_mont(cv, log_n, 0, 1, K.fr_to_mont(cv, x))) == d.coset_fft(x)


@pytest.mark.parametrize("cv", [F.BN254, F.BLS12_381])
def test_srs_and_msm_match_python_and_golden(cv, golden):
    g = golden[cv.name]
    tau = int(g["tau"], 16)
    p = cv.fr.p
    sm = K.srs_mont(cv, tau, 1000)
    assert K.points_from_mont(cv, sm[:4]) == [unhex_point(q) for q in g["srs_first"]]
    for e in g["msm"]:
        n = e["n"]
        if e["seed"] is None:
            sc = [int(s, 16) for s in e["scalars"]]
        else:
            sc = field_elems(p, e["seed"], n)
            if n >= 31:
                sc[0], sc[1], sc[2], sc[5] = 0, 1, p - 1, 0
        out, inf = K.msm_mont(cv, sm[:n], K.fr_to_mont(cv, sc))
        got = None if inf else K.points_from_mont(cv, out)[0]
        assert got == unhex_point(e["result"])
        # canonical (non-Montgomery) scalar entry point
        out2, inf2 = K.msm_mont(cv, sm[:n], K.ints_to_limbs(sc, 4), scalars_mont=False)
        assert (None if inf2 else K.points_from_mont(cv, out2)[0]) == got


def test_cbackend_proof_equals_python_backend(golden):
    cv = F.BN254
    g = golden[cv.name]
    tau = int(g["tau"], 16)
    cs = P.test_circuit(cv)
    n = cs.circuit_bound()
    be = K.CBackend(cv, K.srs_mont(cv, tau, 4 * n + 1))
    pk, epk, vk = P.setup(be, None_srs(4 * n + 1), cs, True)
    proof = P.prove(be, None_srs(4 * n + 1), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk),
                    field_elems(cv.fr.p, g["test_circuit"]["blinder_seed"], P.NUM_BLINDERS))
    assert proof.serialize(cv).hex() == g["test_circuit"]["proof_bytes"]


def None_srs(count):
    return [None] * count  # CBackend.msm takes its bases from srs_arr; only the length is checked
