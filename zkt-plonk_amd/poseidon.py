"""Host-side mirror of the reference's Poseidon hasher for the prover's witness.

``PoseidonGadget`` ~ ``PoseidonRef<ConstraintSystem, PlonkSpecRef, G, WIDTH>`` (plonk-hashing/src/hasher/poseidon/spec.rs:
223-375) seen from the PROVING composer: ``hash(cs, inputs)`` assigns ``vars_per_hash`` fresh variables per call
(constraint_system/arithmetic.rs:19,79 via spec.rs:174-219).  The caller keeps the bookkeeping the composer does -- where a
call's variables start in ``VariableMap::values`` and which variables it was fed -- and ``fill`` has the device compute all
of them at once (zkt_poseidon_gadget_witness_dev), straight into the variable map ``zkt_prove`` gathers its wires from.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from ._lib import Context

VARIABLE_ZERO = 0xFFFFFFFF   # ZKT_VARIABLE_ZERO = Variable::Zero (constraint_system/variable.rs:10-15)


class PoseidonGadget:
    """PoseidonConstants resident in HBM (zkt_poseidon_load) + the record of hash calls whose variables are to be made.
    All arrays are (count, 4) uint64 Montgomery limbs."""

    def __init__(self, ctx: Context, width: int, half_full_rounds: int, partial_rounds: int, round_constants, mds, domain_tag):
        self.ctx, self.width = ctx, width
        self._h = ctx.poseidon_load(width, half_full_rounds, partial_rounds, round_constants, mds, domain_tag)
        self.vars_per_hash = ctx.poseidon_gadget_vars_per_hash(self._h)
        # elements[1] after the last product_mds (spec.rs:315): the running sum j = 1, i = W - 1
        self.hash_var_offset = self.vars_per_hash - 1 - (width - 2) * width
        self.calls: List[Tuple[int, Sequence[int]]] = []
        self._staged = []

    def close(self):
        if self._h:
            self.unstage()
            self.ctx.poseidon_free(self._h)
            self._h = None

    def hash(self, first_variable: int, input_variables: Sequence[int]) -> int:
        """Records one FieldHasher::hash call whose variables occupy [first_variable, first_variable + vars_per_hash) and
        whose inputs are the plain variables `input_variables` (VARIABLE_ZERO allowed).  Returns the variable holding the
        hash."""
        if len(input_variables) > self.width - 1:
            raise ValueError("Poseidon Error: FullBuffer")      # spec.rs:253-255
        self.calls.append((first_variable, tuple(input_variables)))
        return first_variable + self.hash_var_offset

    def levels(self) -> List[List[int]]:
        """Calls grouped by dependency depth: a call whose input is a variable another call makes (the leaf hash of
        circuits/src/withdraw.rs:91-94 takes the commitment hash) must run in a later launch than that call."""
        import bisect
        order = sorted(range(len(self.calls)), key=lambda k: self.calls[k][0])
        bases = [self.calls[k][0] for k in order]
        depth = [-1] * len(self.calls)

        def maker(v):
            at = bisect.bisect_right(bases, v) - 1
            return order[at] if at >= 0 and v < bases[at] + self.vars_per_hash else None

        def depth_of(k, seen=()):
            if depth[k] < 0:
                if k in seen:
                    raise ValueError("Poseidon calls feed each other in a cycle")
                deps = [maker(v) for v in self.calls[k][1] if v != VARIABLE_ZERO]
                depth[k] = 1 + max([depth_of(d, seen + (k,)) for d in deps if d is not None], default=-1)
            return depth[k]

        out: List[List[int]] = []
        for k in range(len(self.calls)):
            d = depth_of(k)
            while len(out) <= d:
                out.append([])
            out[d].append(k)
        return out

    def stage(self):
        """Uploads the recorded calls' trace bases and input indices (structure of the circuit: the same for every witness).
        An absent input and an input that is Variable::Zero are the same LTVariable (Zero, 1, 0) (spec.rs:239-245 reset /
        variable.rs:62-64), so calls of every arity go out together, as launches of arity width - 1 padded with
        VARIABLE_ZERO: ONE launch per dependency level (`levels`), in order on the context's stream."""
        self.unstage()
        arity = self.width - 1
        for lvl in self.levels():
            bases = np.array([self.calls[k][0] for k in lvl], dtype=np.uint32)
            idx = np.full((len(lvl), max(arity, 1)), VARIABLE_ZERO, dtype=np.uint32)
            for row, k in enumerate(lvl):
                ins = self.calls[k][1]
                idx[row, :len(ins)] = ins
            d_base, d_idx = self.ctx.alloc(bases.nbytes), self.ctx.alloc(idx.nbytes)
            self.ctx.upload(d_base, bases)
            self.ctx.upload(d_idx, idx)
            self._staged.append((arity, len(lvl), d_base, d_idx))

    def unstage(self):
        for _, _, d_base, d_idx in getattr(self, "_staged", []):
            self.ctx.free(d_base)
            if d_idx:
                self.ctx.free(d_idx)
        self._staged = []

    def fill(self, d_variables: int, n_vars: int, check: bool = True) -> int:
        """Enqueues the gadget kernel for every recorded call on the context's stream: ONE launch per dependency level
        (`levels`: a call fed by another call's output runs in a later launch).  Inputs that no call makes must already be
        in the map at d_variables.  Returns the number of launches.  With check=True synchronises and raises when an index
        lay outside the map."""
        if not getattr(self, "_staged", None):
            self.stage()
        for arity, count, d_base, d_idx in self._staged:
            self.ctx.poseidon_gadget_witness_dev(self._h, count, arity, d_variables, n_vars, d_input_vars=d_idx if arity else 0,
                                                 d_trace_base=d_base)
        if check:
            self.ctx.poseidon_gadget_check(self._h)
        return len(self._staged)
