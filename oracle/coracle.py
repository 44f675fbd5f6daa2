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
