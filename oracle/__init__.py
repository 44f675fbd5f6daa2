"""CPU oracle for the zkt-plonk prover hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker.  The product path
(``zkt-plonk_amd/`` + ``include/zkt_plonk.h``) never imports or links it.

What it restates (each function cites the reference file:line it follows):
the arkworks-0.3 conventions the reference's prover relies on (ark-ff
Montgomery fields, ark-poly radix-2 domains, ark-ec VariableBaseMSM,
ark-poly-commit SonicKZG10 commit/open, ark-serialize compressed points,
merlin 3.0 / STROBE-128) and the reference's own prover
(``plonk-core/src/proof_system/*``, ``permutation/mod.rs``, ``lookup/*``).

PARITY PINNING (see DESIGN.md "Oracle"):
* pinned by the reference's own literal KATs: ``combine_split``
  (plonk-core/src/lookup/multiset.rs:272-329) and the EthereumTranscript hex
  digests (gadgets/src/transcript.rs:101-127), which also pin Keccak-f[1600],
  the BN254 Fr modulus handling and big-endian scalar encoding;
* pinned by the reference's property tests restated here (z1/z2 grand-product
  identities, blinder invariance, K1/K2 coset checks, prove->verify round trip);
* NTT / MSM / proof *bytes*: the reference holds no golden vector for them
  ("parity unpinned" by the reference's tests).  They are pinned instead by
  canonical uniqueness (naive O(n^2) DFT, naive double-and-add, textbook
  affine formulas with Python big integers) and by published constants
  (field moduli, two-adic roots, generators, the merlin conformance vector).
"""
