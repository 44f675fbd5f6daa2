"""Radix-2 evaluation domains with Python big integers (oracle; test infrastructure only).

Restates ark-poly 0.3 ``Radix2EvaluationDomain`` (third-party, absent from /root/reference)
as used through plonk-core/src/util.rs:63-140:

* ``fft``        : zero-pad coefficients to ``size``; out[i] = sum_j c_j * w^(i*j), natural order
* ``ifft``       : inverse of the above, scaled by size^-1
* ``coset_fft``  : c_j <- c_j * g^j (g = F::multiplicative_generator()), then fft
* ``coset_ifft`` : ifft, then c_j <- c_j * g^-j
"""
from __future__ import annotations

from typing import List, Sequence

from .fields import PrimeField


class Domain:
    """ark_poly::Radix2EvaluationDomain::new(n) (GeneralEvaluationDomain picks it for 2^k)."""

    def __init__(self, f: PrimeField, n: int):
        size = 1 if n <= 1 else 1 << (n - 1).bit_length()
        self.f = f
        self.size = size
        self.log_size = size.bit_length() - 1
        if self.log_size > f.two_adicity:
            raise ValueError("InvalidEvalDomainSize")
        self.group_gen = f.root_of_unity(size)
        self.group_gen_inv = f.inv(self.group_gen)
        self.size_inv = f.inv(size)
        self.coset_gen = f.generator
        self.coset_gen_inv = f.inv(f.generator)

    # -- helpers used by the prover (util.rs:27-59, permutation/mod.rs:204) ---------------
    def elements(self) -> List[int]:
        out, w, p = [], 1, self.f.p
        for _ in range(self.size):
            out.append(w)
            w = w * self.group_gen % p
        return out

    def element(self, i: int) -> int:
        return pow(self.group_gen, i, self.f.p)

    def evaluate_vanishing_polynomial(self, x: int) -> int:
        return (pow(x, self.size, self.f.p) - 1) % self.f.p

    # -- transforms -----------------------------------------------------------------------
    def _transform(self, a: List[int], w: int) -> List[int]:
        """Iterative radix-2 DIT; natural-order input and output."""
        p, n, logn = self.f.p, self.size, self.log_size
        a = list(a)
        for i in range(n):
            j = int(format(i, "0%db" % logn)[::-1], 2) if logn else 0
            if i < j:
                a[i], a[j] = a[j], a[i]
        m = 1
        while m < n:
            wm = pow(w, n // (2 * m), p)
            for k in range(0, n, 2 * m):
                t = 1
                for j in range(m):
                    u = a[k + j]
                    v = a[k + j + m] * t % p
                    a[k + j] = (u + v) % p
                    a[k + j + m] = (u - v) % p
                    t = t * wm % p
            m *= 2
        return a

    def _pad(self, coeffs: Sequence[int]) -> List[int]:
        # ark-poly 0.3 fft_in_place: coeffs.resize(size, 0) -- the reference never passes
        # more than `size` coefficients (polys have <= n+3 coeffs on the 4n domain).
        if len(coeffs) > self.size:
            raise ValueError("more coefficients than the domain size")
        return [c % self.f.p for c in coeffs] + [0] * (self.size - len(coeffs))

    def fft(self, coeffs: Sequence[int]) -> List[int]:
        return self._transform(self._pad(coeffs), self.group_gen)

    def ifft(self, evals: Sequence[int]) -> List[int]:
        p = self.f.p
        out = self._transform(self._pad(evals), self.group_gen_inv)
        return [x * self.size_inv % p for x in out]

    def coset_fft(self, coeffs: Sequence[int]) -> List[int]:
        p = self.f.p
        a = self._pad(coeffs)
        g = 1
        for i in range(len(a)):
            a[i] = a[i] * g % p
            g = g * self.coset_gen % p
        return self._transform(a, self.group_gen)

    def coset_ifft(self, evals: Sequence[int]) -> List[int]:
        p = self.f.p
        a = self.ifft(evals)
        g = 1
        for i in range(len(a)):
            a[i] = a[i] * g % p
            g = g * self.coset_gen_inv % p
        return a


def dft_naive(f: PrimeField, coeffs: Sequence[int], n: int, w: int) -> List[int]:
    """O(n^2) definition used to pin the fast transforms on small sizes."""
    p = f.p
    c = list(coeffs) + [0] * (n - len(coeffs))
    return [sum(c[j] * pow(w, i * j, p) for j in range(n)) % p for i in range(n)]


def trim(coeffs: Sequence[int]) -> List[int]:
    """DensePolynomial::from_coefficients_vec: strip trailing zero coefficients."""
    c = list(coeffs)
    while c and c[-1] == 0:
        c.pop()
    return c


def poly_eval(f: PrimeField, coeffs: Sequence[int], x: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % f.p
    return acc
