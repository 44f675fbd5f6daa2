"""CPU: the product's field arithmetic (29-bit-limb Montgomery product, lazy add/sub, conversions), executed
on the host through zkt_host_field_op, against Python big integers."""
import ctypes
import random

import pytest

import zkt_plonk_amd as z
from oracle import fields as F

FIELDS = [(0, 0, F.BN254_FR), (0, 1, F.BN254_FQ), (1, 0, F.BLS12_381_FR), (1, 1, F.BLS12_381_FQ)]


def _op(L, curve, which, op, a, b, nwords):
    A = (ctypes.c_uint32 * nwords)(*[(a >> (32 * i)) & 0xFFFFFFFF for i in range(nwords)])
    B = (ctypes.c_uint32 * nwords)(*[(b >> (32 * i)) & 0xFFFFFFFF for i in range(nwords)])
    O = (ctypes.c_uint32 * nwords)()
    assert L.zkt_host_field_op(curve, which, op, A, B, O) == 0
    return sum(int(O[i]) << (32 * i) for i in range(nwords))


@pytest.mark.parametrize("curve,which,f", FIELDS, ids=lambda x: getattr(x, "name", str(x)))
def test_host_field_ops_match_big_integers(curve, which, f):
    L = z.lib()
    L.zkt_host_field_op.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_uint32)] * 3
    p, nw = f.p, f.limbs64 * 2
    R = 1 << (32 * nw)
    Rinv = pow(R, -1, p)
    rnd = random.Random(curve * 2 + which)
    specials = [0, 1, 2, p - 1, p - 2, R % p, (R * R) % p, (1 << (f.bits - 1)) % p, p // 2, p // 3]
    pairs = [(a, b) for a in specials for b in specials] + [(rnd.randrange(p), rnd.randrange(p)) for _ in range(400)]
    for a, b in pairs:
        want = a * b * Rinv % p
        assert _op(L, curve, which, 0, a, b, nw) == want
        assert _op(L, curve, which, 1, a, b, nw) == want
        assert _op(L, curve, which, 2, a, b, nw) == a
        # op 3: Montgomery-form inputs a~ = xR, b~ = yR  ->  (3 (x^2 - y^2)) R
        x, y = a * Rinv % p, b * Rinv % p
        assert _op(L, curve, which, 3, a, b, nw) == 3 * (x * x - y * y) * R % p
        # op 6: a^2 + b^2 via the double product (one reduction) cross-checked with two squarings
        assert _op(L, curve, which, 6, a, b, nw) == (x * x + y * y) * R % p
        # op 7: (a - b) * b with the carry-free difference feeding the product
        assert _op(L, curve, which, 7, a, b, nw) == (x - y) * y * R % p
        # op 8: the NTT butterfly: 2 (a - b) * y through limb-wise sums, the wide carry-free difference and the product by
        # a constant with a precomputed quotient (no Montgomery factor: the result keeps a's form)
        assert _op(L, curve, which, 8, a, b, nw) == 2 * (a - b) * y % p
        # op 9: 4 (a + b) through two levels of limb-wise sums and two lazy reductions
        assert _op(L, curve, which, 9, a, b, nw) == 4 * (a + b) % p


@pytest.mark.parametrize("curve,which,f", FIELDS, ids=lambda x: getattr(x, "name", str(x)))
def test_host_inversions(curve, which, f):
    """hostinv.hpp (binary extended Euclid, used between GPU phases) and the kernels' Fermat ladder against pow()."""
    L = z.lib()
    L.zkt_host_field_op.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_uint32)] * 3
    p, nw = f.p, f.limbs64 * 2
    R = 1 << (32 * nw)
    rnd = random.Random(77 + curve * 2 + which)
    vals = [1, 2, 3, p - 1, p - 2, R % p, (R * R) % p, p // 2, 1 << (f.bits - 1)] + [rnd.randrange(1, p) for _ in range(60)]
    for a in vals:
        x = a * pow(R, -1, p) % p                  # a is the Montgomery form of x
        want = pow(x, -1, p) * R % p
        assert _op(L, curve, which, 4, a, 0, nw) == want
    for a in vals[:12]:
        x = a * pow(R, -1, p) % p
        assert _op(L, curve, which, 5, a, 0, nw) == pow(x, -1, p) * R % p
    assert _op(L, curve, which, 4, 0, 0, nw) == 0
