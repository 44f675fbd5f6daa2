"""Field / curve parameters and big-integer field helpers (oracle; test infrastructure only).

Restates the arkworks-0.3 parameter sets the reference instantiates
(plonk-core/Cargo.toml:37-39 -> ark-bn254 / ark-bls12-381 0.3; the crates are
third-party and absent from /root/reference, so the published constants are
restated and checked numerically in tests/test_oracle_fields.py):

* ``FftParameters``: TWO_ADICITY, the 2^s-th root of unity = GENERATOR^((r-1)/2^s)
* ``FpParameters``: MODULUS, R = 2^(64*limbs) mod p, INV = -p^-1 mod 2^64
* in-memory form = little-endian u64 limbs of a*R mod p (Montgomery form)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence


@dataclass(frozen=True)
class PrimeField:
    name: str
    p: int
    limbs64: int
    # FFT parameters (only meaningful for scalar fields)
    two_adicity: int = 0
    generator: int = 0  # multiplicative generator (coset shift of ark-poly coset_fft)

    @property
    def bits(self) -> int:
        return self.p.bit_length()

    @property
    def R(self) -> int:
        return (1 << (64 * self.limbs64)) % self.p

    @property
    def R2(self) -> int:
        return (self.R * self.R) % self.p

    @property
    def inv64(self) -> int:
        """-p^-1 mod 2^64 (ark-ff FpParameters::INV)."""
        return (-pow(self.p, -1, 1 << 64)) % (1 << 64)

    @property
    def inv32(self) -> int:
        return (-pow(self.p, -1, 1 << 32)) % (1 << 32)

    @property
    def two_adic_root(self) -> int:
        """FftParameters::TWO_ADIC_ROOT_OF_UNITY = g^((p-1)/2^s)."""
        return pow(self.generator, (self.p - 1) >> self.two_adicity, self.p)

    def root_of_unity(self, n: int) -> int:
        """FftField::get_root_of_unity(n): square the 2^s-th root (s - log2 n) times
        (ark-ff 0.3 fields/mod.rs; reached via D::new(n), prove.rs:77)."""
        assert n & (n - 1) == 0 and n >= 1
        log_n = n.bit_length() - 1
        if log_n > self.two_adicity:
            raise ValueError("domain too large for two-adicity")
        w = self.two_adic_root
        for _ in range(self.two_adicity - log_n):
            w = w * w % self.p
        return w

    # ---- Montgomery memory form <-> int -------------------------------------------
    def to_mont(self, a: int) -> int:
        return (a % self.p) * self.R % self.p

    def from_mont(self, a: int) -> int:
        return a * pow(self.R, -1, self.p) % self.p

    def inv(self, a: int) -> int:
        a %= self.p
        if a == 0:
            raise ZeroDivisionError("inverse of zero")
        return pow(a, -1, self.p)


def int_to_limbs64(x: int, n: int) -> List[int]:
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def limbs64_to_int(limbs: Sequence[int]) -> int:
    x = 0
    for i, l in enumerate(limbs):
        x |= int(l) << (64 * i)
    return x


@dataclass(frozen=True)
class Curve:
    """Short-Weierstrass G1 y^2 = x^3 + b over fq, scalar field fr."""
    name: str
    curve_id: int
    fr: PrimeField
    fq: PrimeField
    b: int
    gx: int
    gy: int


BN254_FR = PrimeField(
    "bn254_fr",
    21888242871839275222246405745257275088548364400416034343698204186575808495617,
    4, two_adicity=28, generator=5)
BN254_FQ = PrimeField(
    "bn254_fq",
    21888242871839275222246405745257275088696311157297823662689037894645226208583,
    4)
BN254 = Curve("bn254", 0, BN254_FR, BN254_FQ, 3, 1, 2)

BLS12_381_FR = PrimeField(
    "bls12_381_fr",
    0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    4, two_adicity=32, generator=7)
BLS12_381_FQ = PrimeField(
    "bls12_381_fq",
    0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    6)
BLS12_381 = Curve(
    "bls12_381", 1, BLS12_381_FR, BLS12_381_FQ, 4,
    0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
    0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)

CURVES = {"bn254": BN254, "bls12_381": BLS12_381}

# plonk-core/src/permutation/constants.rs:13-20
K1 = 7
K2 = 13
