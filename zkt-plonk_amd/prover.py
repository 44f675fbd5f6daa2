"""Host-side mirror of the reference's prover entry points.

* ``seed_transcript``  ~ ``VerifierKey::seed_transcript`` (plonk-core/src/proof_system/keys/mod.rs:260-275)
* ``GpuProver.prove``  ~ ``ZKTPlonk::prove`` -> ``proof_system::prove`` (plonk-core/src/plonk.rs:94-111,
  plonk-core/src/proof_system/prove.rs:59-470) with the circuit already synthesised into wire values.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

from ._lib import Context, Transcript

PK_ORDER = ("q_m", "q_l", "q_r", "q_o", "q_c", "sigma1", "sigma2", "sigma3", "q_lookup", "q_table")
NUM_BLINDERS = 19  # a(2) b(2) c(2) h1(3) h2(2) z1(3) z2(3) b0 b1 -- prove.rs:125-127,170-171,225,244,296


def seed_transcript(tr: Transcript, n: int, vk_commits: Dict[str, Optional[tuple]]) -> Transcript:
    """keys/mod.rs:260-275; vk_commits maps the PK_ORDER names to affine points (canonical ints) or None."""
    tr.seed(n, [vk_commits[name] for name in PK_ORDER])
    return tr


class GpuKZG10:
    """Host-side mirror of the Commitment seam ``PC: HomomorphicCommitment<F>`` = KZG10 (plonk-core/src/commitment.rs:10-46):
    ``commit`` = kzg10::commit = MSM over the loaded powers (device), ``multi_scalar_mul`` on arbitrary commitments
    (host, commitment.rs:32-45).  Openings are produced inside ``GpuProver.prove``."""

    def __init__(self, ctx: Context, powers_of_g: np.ndarray = None):
        self.ctx = ctx
        if powers_of_g is not None:
            ctx.srs_load(powers_of_g)            # ck.powers_of_g, as PC::trim leaves them (plonk.rs:79-85)

    def commit(self, coeffs: np.ndarray):
        """-> (xy limbs, is_infinity); raises ZktError(5) = TooManyCoefficients beyond the loaded powers."""
        return self.ctx.msm(coeffs, 0, True)

    def multi_scalar_mul(self, commitments: np.ndarray, scalars: np.ndarray):
        from ._lib import g1_msm_host
        return g1_msm_host(self.ctx.curve, commitments, scalars, True)


class GpuProver:
    """Device-resident circuit + SRS; one instance per (circuit, context)."""

    def __init__(self, ctx: Context, log_n: int, pk_polys: Dict[str, np.ndarray] = None):
        self.ctx = ctx
        self.log_n = log_n
        self.n = 1 << log_n
        if pk_polys is not None:
            ctx.circuit_load(log_n, [pk_polys[k] for k in PK_ORDER])

    @classmethod
    def setup(cls, ctx: Context, log_n: int, evals: Dict[str, np.ndarray]):
        """proof_system::setup (plonk-core/src/proof_system/setup.rs:42-166) on the device, from the SetupComposer's
        ten evaluation vectors.  -> (prover, {name: (xy limbs, is_infinity)}) : the loaded prover and the VerifierKey
        commitments."""
        self = cls(ctx, log_n, None)
        pts, inf = ctx.circuit_setup(log_n, [evals[k] for k in PK_ORDER])
        return self, {k: (pts[i], bool(inf[i])) for i, k in enumerate(PK_ORDER)}

    def prove(self, a, b, c, table, public_inputs: Dict[int, np.ndarray], blinders, transcript: Transcript) -> bytes:
        pos = sorted(public_inputs.keys())
        vals = np.stack([np.asarray(public_inputs[p], dtype=np.uint64).reshape(4) for p in pos]) if pos else \
            np.zeros((0, 4), dtype=np.uint64)
        return self.ctx.prove(a, b, c, table, pos, vals, blinders, transcript)
