"""zkt-plonk_amd: MI355X (gfx950) prover hot path for ZKTLabs/zkt-plonk behind a C-ABI.

Host-side mirror of the reference's two generic seams:

* ``GpuDomain``        ~ ``D: EvaluationDomain<F> + EvaluationDomainExt<F>``  (plonk-core/src/util.rs:27-140)
* ``GpuKZG10``         ~ ``PC: HomomorphicCommitment<F>``: commit = G1 MSM over the loaded powers, multi_scalar_mul
                         (plonk-core/src/commitment.rs:10-46)
* ``GpuProver.prove``  ~ ``proof_system::prove``  (plonk-core/src/proof_system/prove.rs:59-470; the openings live inside it)
* ``PoseidonGadget``   ~ ``PoseidonRef<ConstraintSystem, PlonkSpecRef, ..>::hash`` as the proving composer sees it: the gadget's
                         variables, made on the device (plonk-hashing/src/hasher/poseidon/spec.rs:174-375)
* ``parallel``         ~ one proof or many across the GPUs of a node (communicators, SRS slices)

These are thin ctypes mirrors for the tests and ``bench.py``; the product is the C-ABI (include/zkt_plonk.h).

All numerics run in ``libzkt_plonk_hip.so`` (hand-written HIP kernels).  There is no CPU fallback:
importing works without a GPU (so the C-ABI can be inspected), creating a ``Context`` does not.
"""
from ._lib import (  # noqa: F401
    Context, ZktError, Transcript, lib, lib_path, CURVE_BN254, CURVE_BLS12_381, curve_id, declared_symbols,
)
from .domain import GpuDomain  # noqa: F401
from .prover import GpuProver, GpuKZG10, seed_transcript, PK_ORDER, NUM_BLINDERS  # noqa: F401
from .poseidon import PoseidonGadget  # noqa: F401

__all__ = ["Context", "ZktError", "Transcript", "GpuDomain", "GpuProver", "GpuKZG10", "PoseidonGadget", "seed_transcript", "PK_ORDER",
           "NUM_BLINDERS", "lib", "lib_path", "CURVE_BN254", "CURVE_BLS12_381", "curve_id", "declared_symbols"]
