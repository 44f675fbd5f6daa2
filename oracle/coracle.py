"""ctypes binding of the C++ part of the oracle (test infrastructure only).

``build()`` compiles oracle/coracle.cpp with g++ into oracle/_build/liboracle.so.
All arrays are numpy ``uint64`` with arkworks' in-memory layout: one field element =
little-endian u64 limbs of the Montgomery form; an affine point = x limbs || y limbs,
(0, 0) standing for the point at infinity.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .fields import Curve, PrimeField
from .curve import Point
from . import plonk as _plonk

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "coracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def _cpu_share() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))  # a one-GPU box grants a 16-CPU share whatever the host's core count


def lib():
    global _lib
    if _lib is None:
        os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_share()))
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        L.orc_ntt.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, u64p, ctypes.c_size_t, u64p]
        L.orc_msm.argtypes = [ctypes.c_int, u64p, u64p, ctypes.c_size_t, ctypes.c_int, u64p,
                              ctypes.POINTER(ctypes.c_int)]
        L.orc_srs.argtypes = [ctypes.c_int, u64p, u64p, u64p, ctypes.c_size_t, u64p]
        L.orc_fr_convert.argtypes = [ctypes.c_int, ctypes.c_int, u64p, ctypes.c_size_t, u64p]
        L.orc_fq_convert.argtypes = [ctypes.c_int, ctypes.c_int, u64p, ctypes.c_size_t, u64p]
        L.orc_params.argtypes = [ctypes.c_int, u64p]
        pp = ctypes.POINTER(u64p)
        szp = ctypes.POINTER(ctypes.c_size_t)
        L.orc_domain_points.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, u64p]
        L.orc_fr_vec_op.argtypes = [ctypes.c_int, ctypes.c_int, u64p, u64p, ctypes.c_size_t, u64p]
        L.orc_z1_evals.argtypes = [ctypes.c_int, ctypes.c_int] + [u64p] * 9
        L.orc_z2_evals.argtypes = [ctypes.c_int, ctypes.c_int] + [u64p] * 7
        L.orc_quotient_evals.argtypes = [ctypes.c_int, ctypes.c_int, u64p, pp, pp, u64p]
        L.orc_combine_split.argtypes = [u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p, u64p, szp]
        L.orc_lincomb.argtypes = [ctypes.c_int, ctypes.c_int, pp, szp, u64p, ctypes.c_size_t, u64p]
        L.orc_poly_eval.argtypes = [ctypes.c_int, u64p, ctypes.c_size_t, u64p, u64p]
        L.orc_div_linear.argtypes = [ctypes.c_int, u64p, ctypes.c_size_t, u64p, u64p]
        L.orc_num_threads.restype = ctypes.c_int
        L.orc_set_threads.argtypes = [ctypes.c_int]
        L.orc_set_threads(int(os.environ.get("ORACLE_THREADS", _cpu_share())))  # env vars are read too late once
        _lib = L                                                               # another OpenMP user is loaded
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


# ---- int <-> limb arrays -----------------------------------------------------------------------
def ints_to_limbs(vals: Sequence[int], limbs: int) -> np.ndarray:
    nb = limbs * 8
    buf = b"".join(int(v).to_bytes(nb, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(len(vals), limbs).copy()


def limbs_to_ints(arr: np.ndarray) -> List[int]:
    arr = np.ascontiguousarray(arr, dtype=np.uint64)
    nb = arr.shape[-1] * 8
    raw = arr.tobytes()
    return [int.from_bytes(raw[i:i + nb], "little") for i in range(0, len(raw), nb)]


def fr_to_mont(cv: Curve, vals: Sequence[int]) -> np.ndarray:
    """canonical ints -> (n, 4) Montgomery limb array."""
    a = ints_to_limbs(vals, 4)
    out = np.empty_like(a)
    assert lib().orc_fr_convert(cv.curve_id, 1, _p(a), len(vals), _p(out)) == 0
    return out


def fr_from_mont(cv: Curve, arr: np.ndarray) -> List[int]:
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
    out = np.empty_like(arr)
    assert lib().orc_fr_convert(cv.curve_id, 0, _p(arr), arr.shape[0], _p(out)) == 0
    return limbs_to_ints(out)


def points_to_mont(cv: Curve, pts: Sequence[Point]) -> np.ndarray:
    """affine points -> (n, 2*limbs) Montgomery array; infinity -> all zero."""
    L = cv.fq.limbs64
    flat = []
    for P in pts:
        flat.extend((0, 0) if P is None else (P[0], P[1]))
    a = ints_to_limbs(flat, L)
    out = np.empty_like(a)
    assert lib().orc_fq_convert(cv.curve_id, 1, _p(a), a.shape[0], _p(out)) == 0
    return out.reshape(len(pts), 2 * L)


def points_from_mont(cv: Curve, arr: np.ndarray) -> List[Point]:
    L = cv.fq.limbs64
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, L)
    out = np.empty_like(arr)
    assert lib().orc_fq_convert(cv.curve_id, 0, _p(arr), arr.shape[0], _p(out)) == 0
    vals = limbs_to_ints(out)
    pts: List[Point] = []
    for i in range(0, len(vals), 2):
        pts.append(None if vals[i] == 0 and vals[i + 1] == 0 else (vals[i], vals[i + 1]))
    return pts


# ---- transforms ----------------------------------------------------------------------------------
def ntt_mont(cv: Curve, log_n: int, inverse: bool, coset: bool, arr: np.ndarray) -> np.ndarray:
    """(len, 4) Montgomery array (len <= 2^log_n, zero padded) -> (2^log_n, 4)."""
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
    out = np.empty((1 << log_n, 4), dtype=np.uint64)
    rc = lib().orc_ntt(cv.curve_id, log_n, int(inverse), int(coset), _p(arr), arr.shape[0], _p(out))
    if rc:
        raise ValueError("orc_ntt failed: %d" % rc)
    return out


def msm_mont(cv: Curve, bases: np.ndarray, scalars: np.ndarray, scalars_mont: bool = True) -> Tuple[np.ndarray, bool]:
    L = cv.fq.limbs64
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 2 * L)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    n = min(bases.shape[0], scalars.shape[0])
    out = np.zeros(2 * L, dtype=np.uint64)
    inf = ctypes.c_int(0)
    rc = lib().orc_msm(cv.curve_id, _p(bases), _p(scalars), n, int(scalars_mont), _p(out), ctypes.byref(inf))
    if rc:
        raise ValueError("orc_msm failed: %d" % rc)
    return out, bool(inf.value)


def srs_mont(cv: Curve, tau: int, count: int) -> np.ndarray:
    """[tau^i]G, i < count, as a (count, 2*limbs) Montgomery array."""
    L = cv.fq.limbs64
    gx = ints_to_limbs([cv.gx], L)
    gy = ints_to_limbs([cv.gy], L)
    t = ints_to_limbs([tau % cv.fr.p], 4)
    out = np.empty((count, 2 * L), dtype=np.uint64)
    assert lib().orc_srs(cv.curve_id, _p(gx), _p(gy), _p(t), count, _p(out)) == 0
    return out


# ---- the prover's O(n) loops on Montgomery arrays (pinned against oracle/plonk.py in tests/test_coracle.py) ----
def _arr(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)


def _ptrs(arrs):
    return (ctypes.POINTER(ctypes.c_uint64) * len(arrs))(*[_p(a) for a in arrs])


def domain_points(cv: Curve, log_n: int, coset: bool = False) -> np.ndarray:
    out = np.empty((1 << log_n, 4), dtype=np.uint64)
    assert lib().orc_domain_points(cv.curve_id, log_n, int(coset), _p(out)) == 0
    return out


def vec_op(cv: Curve, op: str, a, b) -> np.ndarray:
    a, b = _arr(a), _arr(b)
    assert a.shape == b.shape
    out = np.empty_like(a)
    assert lib().orc_fr_vec_op(cv.curve_id, {"add": 0, "sub": 1, "mul": 2}[op], _p(a), _p(b), a.shape[0], _p(out)) == 0
    return out


def z1_evals(cv: Curve, log_n: int, beta, gamma, a, b, c, s1, s2, s3) -> np.ndarray:
    """permutation/mod.rs:181-254; beta, gamma: (4,) Montgomery limbs; vectors: (n, 4)."""
    out = np.empty((1 << log_n, 4), dtype=np.uint64)
    vs = [_arr(x) for x in (a, b, c, s1, s2, s3)]
    assert all(v.shape[0] == 1 << log_n for v in vs)
    rc = lib().orc_z1_evals(cv.curve_id, log_n, _p(_arr(beta)), _p(_arr(gamma)), *[_p(v) for v in vs], _p(out))
    if rc:
        raise ZeroDivisionError("zero denominator in the permutation grand product")
    return out


def z2_evals(cv: Curve, log_n: int, delta, epsilon, f, t, h1, h2) -> np.ndarray:
    """lookup/mod.rs:94-151."""
    out = np.empty((1 << log_n, 4), dtype=np.uint64)
    vs = [_arr(x) for x in (f, t, h1, h2)]
    assert all(v.shape[0] == 1 << log_n for v in vs)
    rc = lib().orc_z2_evals(cv.curve_id, log_n, _p(_arr(delta)), _p(_arr(epsilon)), *[_p(v) for v in vs], _p(out))
    if rc:
        raise ZeroDivisionError("zero denominator in the lookup grand product")
    return out


EPK_ORDER = ("q_m", "q_l", "q_r", "q_o", "q_c", "q_lookup", "q_table", "sigma1", "sigma2", "sigma3", "x", "zh", "l_1")
WIT_ORDER = ("a", "b", "c", "pi", "z1", "z2", "t", "h1", "h2")


def quotient_evals(cv: Curve, log_n: int, ch, epk: dict, wit: dict) -> np.ndarray:
    """quotient_poly.rs:98-224; ch: (5, 4) alpha beta gamma delta epsilon; epk / wit: name -> (4n, 4)."""
    N = 4 << log_n
    e = [_arr(epk[k]) for k in EPK_ORDER]
    w = [_arr(wit[k]) for k in WIT_ORDER]
    assert all(v.shape[0] == N for v in e + w)
    out = np.empty((N, 4), dtype=np.uint64)
    rc = lib().orc_quotient_evals(cv.curve_id, log_n, _p(_arr(ch).reshape(-1)), _ptrs(e), _ptrs(w), _p(out))
    if rc:
        raise ZeroDivisionError("vanishing polynomial is zero on the coset")
    return out


def combine_split(t, f):
    """lookup/multiset.rs:103-146 on Montgomery arrays -> (h1, h2)."""
    t, f = _arr(t), _arr(f)
    h1 = np.empty((t.shape[0] + f.shape[0], 4), dtype=np.uint64)
    h2 = np.empty_like(h1)
    lens = (ctypes.c_size_t * 2)()
    rc = lib().orc_combine_split(_p(t), t.shape[0], _p(f), f.shape[0], _p(h1), _p(h2), lens)
    if rc:
        raise KeyError("ElementNotIndexedInTable")
    return h1[:lens[0]].copy(), h2[:lens[1]].copy()


def lincomb(cv: Curve, polys, scalars, out_len: int) -> np.ndarray:
    """sum_k scalars[k] * polys[k], zero padded / cut to out_len coefficients."""
    ps = [_arr(q) for q in polys]
    keep = [q if q.shape[0] else np.zeros((1, 4), dtype=np.uint64) for q in ps]
    lens = (ctypes.c_size_t * len(ps))(*[q.shape[0] for q in ps])
    sc = _arr(scalars)
    assert sc.shape[0] == len(ps)
    out = np.empty((out_len, 4), dtype=np.uint64)
    assert lib().orc_lincomb(cv.curve_id, len(ps), _ptrs(keep), lens, _p(sc.reshape(-1)), out_len, _p(out)) == 0
    return out


def poly_eval(cv: Curve, poly, point) -> np.ndarray:
    poly = _arr(poly)
    out = np.zeros(4, dtype=np.uint64)
    if poly.shape[0]:
        assert lib().orc_poly_eval(cv.curve_id, _p(poly), poly.shape[0], _p(_arr(point)), _p(out)) == 0
    return out


def div_linear(cv: Curve, poly, z) -> np.ndarray:
    """(p(X) - p(z)) / (X - z): len - 1 coefficients."""
    poly = _arr(poly)
    out = np.empty((max(poly.shape[0] - 1, 0), 4), dtype=np.uint64)
    if poly.shape[0] > 1:
        assert lib().orc_div_linear(cv.curve_id, _p(poly), poly.shape[0], _p(_arr(z)), _p(out)) == 0
    return out


def trim_len(arr) -> int:
    """DensePolynomial::from_coefficients_vec: length after stripping trailing zero coefficients."""
    nz = np.flatnonzero(_arr(arr).any(axis=1))
    return int(nz[-1]) + 1 if nz.size else 0


def params(which: int) -> dict:
    buf = np.zeros(32, dtype=np.uint64)
    n = lib().orc_params(which, _p(buf))
    v = [int(x) for x in buf]
    to_int = lambda l: sum(x << (64 * i) for i, x in enumerate(l))
    return dict(p=to_int(v[0:n]), inv=v[n], r=to_int(v[n + 1:2 * n + 1]), r2=to_int(v[2 * n + 1:3 * n + 1]))


def num_threads() -> int:
    return lib().orc_num_threads()


class CBackend(_plonk.Backend):
    """oracle.plonk.Backend whose transforms and MSM run in the C++ restatement; the SRS is held
    as a Montgomery array (``srs_arr``) and ``srs`` passed to commit() is ignored in favour of it."""

    def __init__(self, cv: Curve, srs_arr: Optional[np.ndarray] = None):
        super().__init__(cv)
        self.srs_arr = srs_arr

    def _ntt(self, n, vals, inverse, coset):
        log_n = n.bit_length() - 1
        assert 1 << log_n == n
        arr = fr_to_mont(self.cv, vals) if len(vals) else np.zeros((0, 4), dtype=np.uint64)
        return fr_from_mont(self.cv, ntt_mont(self.cv, log_n, inverse, coset, arr))

    def ifft(self, n, evals):
        return self._ntt(n, evals, True, False)

    def fft(self, n, coeffs):
        return self._ntt(n, coeffs, False, False)

    def coset_fft(self, n, coeffs):
        return self._ntt(n, coeffs, False, True)

    def coset_ifft(self, n, evals):
        return self._ntt(n, evals, True, True)

    def msm(self, bases, scalars):
        if len(scalars) == 0:
            return None
        if self.srs_arr is not None:
            b = self.srs_arr[:len(scalars)]
        else:
            b = points_to_mont(self.cv, bases)
        out, inf = msm_mont(self.cv, b, fr_to_mont(self.cv, scalars), True)
        return None if inf else points_from_mont(self.cv, out)[0]
