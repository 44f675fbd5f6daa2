"""G1 arithmetic, MSM and point encodings with Python big integers
(oracle; test infrastructure only).

Points are ``None`` (infinity) or ``(x, y)`` canonical integers.
Restates ark-ec 0.3 (third-party, absent from /root/reference):
``VariableBaseMSM::multi_scalar_mul`` as called from
plonk-core/src/commitment.rs:42,78 and (through ark-poly-commit kzg10
commit/open) from proof_system/prove.rs:134,179,250,307,374,381,427.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

from .fields import Curve

Point = Optional[Tuple[int, int]]


def is_on_curve(cv: Curve, P: Point) -> bool:
    if P is None:
        return True
    x, y = P
    q = cv.fq.p
    return (y * y - (x * x * x + cv.b)) % q == 0


def neg(cv: Curve, P: Point) -> Point:
    if P is None:
        return None
    return (P[0], (-P[1]) % cv.fq.p)


def add(cv: Curve, P: Point, Q: Point) -> Point:
    """Textbook affine chord-and-tangent (a = 0)."""
    if P is None:
        return Q
    if Q is None:
        return P
    q = cv.fq.p
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % q == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, q) % q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, q) % q
    x3 = (lam * lam - x1 - x2) % q
    y3 = (lam * (x1 - x3) - y1) % q
    return (x3, y3)


def double(cv: Curve, P: Point) -> Point:
    return add(cv, P, P)


# --- Jacobian coordinates for speed in the Python oracle (X/Z^2, Y/Z^3) -----------------
def _jdouble(q, P):
    X, Y, Z = P
    if Z == 0:
        return P
    A = X * X % q
    B = Y * Y % q
    C = B * B % q
    D = 2 * ((X + B) * (X + B) - A - C) % q
    E = 3 * A % q
    F = E * E % q
    X3 = (F - 2 * D) % q
    Y3 = (E * (D - X3) - 8 * C) % q
    Z3 = 2 * Y * Z % q
    return (X3, Y3, Z3)


def _jadd_mixed(q, P, Q):
    """P Jacobian + Q affine (x, y)."""
    X1, Y1, Z1 = P
    if Z1 == 0:
        return (Q[0], Q[1], 1)
    Z1Z1 = Z1 * Z1 % q
    U2 = Q[0] * Z1Z1 % q
    S2 = Q[1] * Z1 % q * Z1Z1 % q
    if U2 == X1:
        if S2 == Y1:
            return _jdouble(q, P)
        return (1, 1, 0)
    H = (U2 - X1) % q
    HH = H * H % q
    I = 4 * HH % q
    J = H * I % q
    r = 2 * (S2 - Y1) % q
    V = X1 * I % q
    X3 = (r * r - J - 2 * V) % q
    Y3 = (r * (V - X3) - 2 * Y1 * J) % q
    Z3 = ((Z1 + H) * (Z1 + H) - Z1Z1 - HH) % q
    return (X3, Y3, Z3)


def _jadd(q, P, Q):
    if P[2] == 0:
        return Q
    if Q[2] == 0:
        return P
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    Z1Z1 = Z1 * Z1 % q
    Z2Z2 = Z2 * Z2 % q
    U1 = X1 * Z2Z2 % q
    U2 = X2 * Z1Z1 % q
    S1 = Y1 * Z2 % q * Z2Z2 % q
    S2 = Y2 * Z1 % q * Z1Z1 % q
    if U1 == U2:
        if S1 == S2:
            return _jdouble(q, P)
        return (1, 1, 0)
    H = (U2 - U1) % q
    I = (2 * H) * (2 * H) % q
    J = H * I % q
    r = 2 * (S2 - S1) % q
    V = U1 * I % q
    X3 = (r * r - J - 2 * V) % q
    Y3 = (r * (V - X3) - 2 * S1 * J) % q
    Z3 = ((Z1 + Z2) * (Z1 + Z2) - Z1Z1 - Z2Z2) % q * H % q
    return (X3, Y3, Z3)


def _to_affine(q, P) -> Point:
    X, Y, Z = P
    if Z == 0:
        return None
    zi = pow(Z, -1, q)
    zi2 = zi * zi % q
    return (X * zi2 % q, Y * zi2 % q * zi % q)


def scalar_mul(cv: Curve, k: int, P: Point) -> Point:
    """k*P by left-to-right double-and-add (k reduced mod r; k may be any int)."""
    if P is None:
        return None
    k %= cv.fr.p
    q = cv.fq.p
    acc = (1, 1, 0)
    for bit in bin(k)[2:] if k else "":
        acc = _jdouble(q, acc)
        if bit == "1":
            acc = _jadd_mixed(q, acc, P)
    return _to_affine(q, acc)


def generator(cv: Curve) -> Point:
    return (cv.gx, cv.gy)


def msm_naive(cv: Curve, bases: Sequence[Point], scalars: Sequence[int]) -> Point:
    """sum_i scalars[i]*bases[i] over min(len) pairs; the canonical group element that
    VariableBaseMSM::multi_scalar_mul must also return (algorithm independent)."""
    q = cv.fq.p
    acc = (1, 1, 0)
    for P, s in zip(bases, scalars):
        if P is None or s % cv.fr.p == 0:
            continue
        R = scalar_mul(cv, s, P)
        if R is not None:
            acc = _jadd_mixed(q, acc, R)
    return _to_affine(q, acc)


def ln_without_floats(a: int) -> int:
    """ark-ec 0.3 msm: (log2(a) * 69 / 100) with floor log2."""
    return (a.bit_length() - 1) * 69 // 100


def msm_window_bits(size: int) -> int:
    """Window c chosen by ark-ec 0.3 VariableBaseMSM."""
    return 3 if size < 32 else ln_without_floats(size) + 2


def msm_reference_adds(n: int, scalar_bits: int) -> int:
    """Algorithmic G1 additions of the reference's Pippenger at its own window
    (BASELINE.md section 2 / SURVEY.md section 8d): ceil(l/c)*n + 2*ceil(l/c)*(2^c - 1)."""
    c = msm_window_bits(n)
    w = -(-scalar_bits // c)
    return w * n + 2 * w * ((1 << c) - 1)


def msm_pippenger(cv: Curve, bases: Sequence[Point], scalars: Sequence[int]) -> Point:
    """Step-for-step restatement of ark-ec 0.3 VariableBaseMSM::multi_scalar_mul
    (bucket method, unsigned c-bit windows, unit scalars handled in window 0,
    running-sum bucket reduction, Horner over windows)."""
    q = cv.fq.p
    size = min(len(bases), len(scalars))
    pairs = [(scalars[i], bases[i]) for i in range(size) if scalars[i] != 0]
    c = msm_window_bits(size)
    num_bits = cv.fr.bits
    zero = (1, 1, 0)
    window_sums = []
    for w_start in range(0, num_bits, c):
        res = zero
        buckets = [zero] * ((1 << c) - 1)
        for s, base in pairs:
            if base is None:
                continue
            if s == 1:
                if w_start == 0:
                    res = _jadd_mixed(q, res, base)
            else:
                d = (s >> w_start) % (1 << c)
                if d != 0:
                    buckets[d - 1] = _jadd_mixed(q, buckets[d - 1], base)
        running = zero
        for b in reversed(buckets):
            running = _jadd(q, running, b)
            res = _jadd(q, res, running)
        window_sums.append(res)
    total = zero
    for s_i in reversed(window_sums[1:]):
        total = _jadd(q, total, s_i)
        for _ in range(c):
            total = _jdouble(q, total)
    return _to_affine(q, _jadd(q, window_sums[0], total))


# --- encodings -------------------------------------------------------------------------
def fe_to_le_bytes(x: int, nbytes: int) -> bytes:
    return int(x).to_bytes(nbytes, "little")


def point_to_bytes_uncompressed(cv: Curve, P: Point) -> bytes:
    """ark-ec 0.3 ``impl ToBytes for GroupAffine``: x || y || infinity(u8), each coordinate the
    canonical integer as little-endian u64 limbs.  This is what MerlinTranscript's
    append_commitment feeds to merlin (plonk-core/src/transcript.rs:83-88).
    GroupAffine::zero() is (0, 1, infinity = true)."""
    nb = cv.fq.limbs64 * 8
    if P is None:
        return fe_to_le_bytes(0, nb) + fe_to_le_bytes(1, nb) + b"\x01"
    return fe_to_le_bytes(P[0], nb) + fe_to_le_bytes(P[1], nb) + b"\x00"


def point_serialize_compressed(cv: Curve, P: Point) -> bytes:
    """ark-serialize 0.3 CanonicalSerialize for short-Weierstrass GroupAffine:
    x little-endian in ceil((MODULUS_BITS + 2)/8) bytes with SWFlags in the top bits of the
    last byte: bit 7 = PositiveY (y > -y as canonical integers), bit 6 = infinity
    (infinity serialises x = 0).  Proof wire format, proof_system/proof.rs:98-155."""
    nb = (cv.fq.bits + 2 + 7) // 8
    if P is None:
        b = bytearray(nb)
        b[-1] |= 1 << 6
        return bytes(b)
    x, y = P
    b = bytearray(fe_to_le_bytes(x, nb))
    if y > (cv.fq.p - y) % cv.fq.p:
        b[-1] |= 1 << 7
    return bytes(b)


def point_deserialize_compressed(cv: Curve, data: bytes) -> Point:
    nb = (cv.fq.bits + 2 + 7) // 8
    assert len(data) == nb
    b = bytearray(data)
    flags = b[-1] & 0xC0
    b[-1] &= 0x3F
    if flags & 0x40:
        return None
    x = int.from_bytes(bytes(b), "little")
    q = cv.fq.p
    rhs = (x * x * x + cv.b) % q
    y = sqrt_mod(rhs, q)
    if y is None:
        raise ValueError("x not on curve")
    ny = (q - y) % q
    big, small = (y, ny) if y > ny else (ny, y)
    return (x, big if flags & 0x80 else small)


def sqrt_mod(a: int, p: int) -> Optional[int]:
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    # Tonelli-Shanks
    s, t = 0, p - 1
    while t % 2 == 0:
        s += 1
        t //= 2
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, tt, r = s, pow(z, t, p), pow(a, t, p), pow(a, (t + 1) // 2, p)
    while tt != 1:
        i, x = 0, tt
        while x != 1:
            x = x * x % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        tt, r = tt * c % p, r * b % p
    return r


def srs_powers(cv: Curve, tau: int, count: int) -> List[Point]:
    """[tau^i]G for i < count: the ``powers_of_g`` of a KZG10 SRS with known (test) trapdoor
    (ark-poly-commit kzg10 setup semantics; trapdoor fixed so that openings can be checked
    in G1 without a pairing)."""
    out: List[Point] = []
    s = 1
    g = generator(cv)
    for _ in range(count):
        out.append(scalar_mul(cv, s, g))
        s = s * tau % cv.fr.p
    return out
