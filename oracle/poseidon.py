"""Native Poseidon permutation of the reference's hash gadget (oracle; test infrastructure only).

Restates plonk-hashing/src/hasher/poseidon/spec.rs: full_round :18-37, partial_round :39-54, add_round_constants
:56-71, product_mds :73-88 (result[j] = sum_i m[i][j] * state[i]), quintic_s_box :91-112, the input layout
:239-265 (state[0] = domain tag, inputs from position 1) and the round schedule of output_hash :267-316 (half_full
full rounds, the partial rounds, half_full full rounds; output = state[1]).  The reference generates its constants at
run time (constants.rs:27) and holds no known-answer vector for this hash: "parity unpinned"; the GPU kernel is
compared with this restatement on arbitrary constants."""
from __future__ import annotations

from typing import List, Sequence


def permute(p: int, width: int, half_full: int, partial: int, rc: Sequence[int], mds: Sequence[Sequence[int]], domain_tag: int,
            inputs: Sequence[int]):
    assert len(inputs) <= width - 1, "FullBuffer (spec.rs:253-257)"
    st = [domain_tag % p] + [x % p for x in inputs] + [0] * (width - 1 - len(inputs))
    trace: List[List[int]] = [list(st)]
    off = 0
    rounds = 2 * half_full + partial
    for r in range(rounds):
        full = r < half_full or r >= half_full + partial
        if full:
            st = [pow((x + rc[off + i]) % p, 5, p) for i, x in enumerate(st)]
        else:
            st = [(x + rc[off + i]) % p for i, x in enumerate(st)]
            st[0] = pow(st[0], 5, p)
        off += width
        st = [sum(mds[i][j] * st[i] for i in range(width)) % p for j in range(width)]
        trace.append(list(st))
    return st[1], trace
