"""Reduced Tate pairing on BN254 and BLS12-381 with Python big integers (oracle; test infrastructure only).

The reference leaves pairings to ark-ec 0.3 (`SonicKZG10::check`, reached from plonk-core/src/proof_system/proof.rs:420-500;
the crate is absent from /root/reference).  arkworks computes the optimal ate pairing; a verifier only asks whether a
PRODUCT of pairings is one, and every non-degenerate bilinear pairing on G1 x G2 answers that question identically (they
are powers of one another with an exponent prime to r).  This oracle therefore restates the simplest one:

    t(P, Q) = f_{r,P}(psi(Q)) ^ ((p^12 - 1) / r)        P in G1 = E(Fq)[r], Q in G2 = E'(Fq2)[r]

* tower: Fq2 = Fq[u]/(u^2 + 1), Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v); xi = 9 + u (BN254), 1 + u (BLS12-381)
* twist: BN254 D-type  E': y^2 = x^3 + 3 / xi,  psi(x, y) = (x w^2, y w^3);
         BLS12-381 M-type E': y^2 = x^3 + 4 xi, psi(x, y) = (x / w^2, y / w^3)
* Miller loop over the bits of r with affine lines through multiples of P evaluated at psi(Q); vertical lines have values
  in Fq6 and die in the final exponentiation (denominator elimination), which is one generic power.

"Parity unpinned" by the reference (no pairing vector in the tree): pinned by bilinearity, non-degeneracy, and agreement
with the trapdoor identity L == tau W on real openings (tests/test_pairing_host.py)."""
from __future__ import annotations

from typing import Optional, Tuple

from . import curve as C
from .fields import Curve

# ---- Fq2 (pairs), Fq6 (triples of Fq2), Fq12 (pairs of Fq6) ------------------------------------------------------


class Tower:
    def __init__(self, cv: Curve):
        self.cv = cv
        self.p = cv.fq.p
        self.xi = (9, 1) if cv.name == "bn254" else (1, 1)
        self.d_type = cv.name == "bn254"
        self.zero2, self.one2 = (0, 0), (1, 0)
        self.zero6 = (self.zero2,) * 3
        self.one6 = (self.one2, self.zero2, self.zero2)
        self.one12 = (self.one6, self.zero6)
        # twist coefficient b' and the untwisting constants w^2, w^3 (or their inverses)
        b = (cv.b, 0)
        self.b_twist = self.mul2(b, self.inv2(self.xi)) if self.d_type else self.mul2(b, self.xi)
        w = (self.zero6, self.one6)
        w2 = self.mul12(w, w)
        w3 = self.mul12(w2, w)
        self.ux, self.uy = (w2, w3) if self.d_type else (self.inv12(w2), self.inv12(w3))
        self.final_exp = (self.p ** 12 - 1) // cv.fr.p
        assert (self.p ** 12 - 1) % cv.fr.p == 0

    # Fq2
    def add2(self, a, b): return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)
    def sub2(self, a, b): return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)
    def neg2(self, a): return ((-a[0]) % self.p, (-a[1]) % self.p)
    def mul2(self, a, b):
        p = self.p
        return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)
    def inv2(self, a):
        p = self.p
        d = pow(a[0] * a[0] + a[1] * a[1], -1, p)
        return (a[0] * d % p, (-a[1]) * d % p)
    # Fq6 = Fq2[v] / (v^3 - xi)
    def add6(self, a, b): return tuple(self.add2(x, y) for x, y in zip(a, b))
    def sub6(self, a, b): return tuple(self.sub2(x, y) for x, y in zip(a, b))
    def neg6(self, a): return tuple(self.neg2(x) for x in a)
    def mul6(self, a, b):
        m, A, xi = self.mul2, self.add2, self.xi
        t = [self.zero2] * 5
        for i in range(3):
            for j in range(3):
                t[i + j] = A(t[i + j], m(a[i], b[j]))
        return (A(t[0], m(xi, t[3])), A(t[1], m(xi, t[4])), t[2])
    def mulv6(self, a):   # a * v
        return (self.mul2(self.xi, a[2]), a[0], a[1])
    def inv6(self, a):
        m, S, A, xi = self.mul2, self.sub2, self.add2, self.xi
        c0 = S(m(a[0], a[0]), m(xi, m(a[1], a[2])))
        c1 = S(m(xi, m(a[2], a[2])), m(a[0], a[1]))
        c2 = S(m(a[1], a[1]), m(a[0], a[2]))
        t = A(m(a[0], c0), m(xi, A(m(a[2], c1), m(a[1], c2))))
        ti = self.inv2(t)
        return (m(c0, ti), m(c1, ti), m(c2, ti))
    # Fq12 = Fq6[w] / (w^2 - v)
    def mul12(self, a, b):
        a0b0, a1b1 = self.mul6(a[0], b[0]), self.mul6(a[1], b[1])
        c0 = self.add6(a0b0, self.mulv6(a1b1))
        c1 = self.sub6(self.sub6(self.mul6(self.add6(a[0], a[1]), self.add6(b[0], b[1])), a0b0), a1b1)
        return (c0, c1)
    def inv12(self, a):
        t = self.inv6(self.sub6(self.mul6(a[0], a[0]), self.mulv6(self.mul6(a[1], a[1]))))
        return (self.mul6(a[0], t), self.neg6(self.mul6(a[1], t)))
    def pow12(self, a, e):
        r = self.one12
        for bit in bin(e)[2:]:
            r = self.mul12(r, r)
            if bit == "1":
                r = self.mul12(r, a)
        return r
    def scal12(self, a, s):   # Fq12 times an element of Fq
        return tuple(tuple((x[0] * s % self.p, x[1] * s % self.p) for x in h) for h in a)
    def from_fq(self, s):
        return (((s % self.p, 0), self.zero2, self.zero2), self.zero6)
    def from_fq2(self, x):
        return ((x, self.zero2, self.zero2), self.zero6)

    # ---- G2 on the twist (affine, Fq2 coordinates), None = infinity ------------------------------------------------
    def g2_on_curve(self, Q):
        if Q is None:
            return True
        x, y = Q
        return self.sub2(self.mul2(y, y), self.add2(self.mul2(self.mul2(x, x), x), self.b_twist)) == (0, 0)
    def g2_add(self, P, Q):
        if P is None: return Q
        if Q is None: return P
        (x1, y1), (x2, y2) = P, Q
        if x1 == x2:
            if self.add2(y1, y2) == (0, 0):
                return None
            lam = self.mul2(self.mul2((3, 0), self.mul2(x1, x1)), self.inv2(self.mul2((2, 0), y1)))
        else:
            lam = self.mul2(self.sub2(y2, y1), self.inv2(self.sub2(x2, x1)))
        x3 = self.sub2(self.sub2(self.mul2(lam, lam), x1), x2)
        return (x3, self.sub2(self.mul2(lam, self.sub2(x1, x3)), y1))
    def g2_mul(self, k, Q):
        R, A = None, Q
        while k:
            if k & 1:
                R = self.g2_add(R, A)
            A = self.g2_add(A, A)
            k >>= 1
        return R
    def g2_neg(self, Q):
        return None if Q is None else (Q[0], self.neg2(Q[1]))

    # ---- Miller loop f_{r,P}(psi(Q)) and the reduced pairing --------------------------------------------------------
    def miller(self, P, Q):
        if P is None or Q is None:
            return self.one12
        p, cv = self.p, self.cv
        xq = self.mul12(self.from_fq2(Q[0]), self.ux)
        yq = self.mul12(self.from_fq2(Q[1]), self.uy)
        f = self.one12
        T = P
        sub12 = lambda a, b: (self.sub6(a[0], b[0]), self.sub6(a[1], b[1]))

        def line(T, lam):   # (y_Q - y_T) - lam (x_Q - x_T)
            return sub12(sub12(yq, self.from_fq(T[1])), self.scal12(sub12(xq, self.from_fq(T[0])), lam))

        for bit in bin(cv.fr.p)[3:]:
            lam = 3 * T[0] * T[0] * pow(2 * T[1], -1, p) % p
            f = self.mul12(self.mul12(f, f), line(T, lam))
            T = C.add(cv, T, T)
            if bit == "1":
                if T[0] == P[0]:          # T = -P: the chord is vertical (value in Fq6, killed by the final power)
                    T = None
                    continue
                lam = (P[1] - T[1]) * pow(P[0] - T[0], -1, p) % p
                f = self.mul12(f, line(T, lam))
                T = C.add(cv, T, P)
        assert T is None
        return f

    def pairing(self, P, Q):
        return self.pow12(self.miller(P, Q), self.final_exp)

    def product_is_one(self, pairs) -> bool:
        """prod_i t(P_i, Q_i) == 1 with one final exponentiation."""
        f = self.one12
        for P, Q in pairs:
            f = self.mul12(f, self.miller(P, Q))
        return self.pow12(f, self.final_exp) == self.one12


# G2 generators of ark-bn254 / ark-bls12-381 0.3 (published constants; checked to lie on the twist and to have order r
# in tests/test_pairing_host.py): x = c0 + c1 u, y = c0 + c1 u
G2_GENERATORS = {
    "bn254": ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
               11559732032986387107991004021392285783925812861821192530917403151452391805634),
              (8495653923123431417604973247489272438418190587263600148770280649306958101930,
               4082367875863433681332203403145435568316851327593401208105741076214120093531)),
    "bls12_381": ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
                   0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
                  (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
                   0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)),
}
