"""GpuDomain: host-side mirror of the reference's Domain seam.

``D: EvaluationDomain<F> + EvaluationDomainExt<F>`` (plonk-core/src/proof_system/prove.rs:70,
plonk-core/src/util.rs:27-59); the methods below are the ones the prover calls
(util.rs:71-139; prove.rs:77,441).  Vectors are numpy ``uint64`` arrays of shape (len, 4):
arkworks' in-memory Montgomery limbs, exactly what the Rust shim would hand over.
"""
from __future__ import annotations

import numpy as np

from ._lib import Context, ZktError


class GpuDomain:
    def __init__(self, ctx: Context, num_coeffs: int):
        """``D::new(n)``: smallest power-of-two domain holding ``num_coeffs``; raises
        ZktError(ZKT_ERR_INVALID_DOMAIN_SIZE) like Error::InvalidEvalDomainSize (prove.rs:77-81)."""
        self.ctx = ctx
        size = 1 if num_coeffs <= 1 else 1 << (num_coeffs - 1).bit_length()
        self._size = size
        self._log = size.bit_length() - 1
        self._gen = ctx.group_gen(self._log)  # also validates against TWO_ADICITY

    def size(self) -> int:
        return self._size

    def log_size_of_group(self) -> int:  # util.rs:42-50
        return self._log

    def group_gen(self) -> np.ndarray:   # util.rs:52-58
        return self._gen.copy()

    def fft(self, coeffs: np.ndarray) -> np.ndarray:          # util.rs:104-113
        return self.ctx.ntt(self._log, coeffs, inverse=False, coset=False)

    def ifft(self, evals: np.ndarray) -> np.ndarray:          # util.rs:63-86
        return self.ctx.ntt(self._log, evals, inverse=True, coset=False)

    def coset_fft(self, coeffs: np.ndarray) -> np.ndarray:    # util.rs:117-140
        return self.ctx.ntt(self._log, coeffs, inverse=False, coset=True)

    def coset_ifft(self, evals: np.ndarray) -> np.ndarray:    # util.rs:90-100
        return self.ctx.ntt(self._log, evals, inverse=True, coset=True)
