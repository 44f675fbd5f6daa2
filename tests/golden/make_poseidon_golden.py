#!/usr/bin/env python3
"""Fixture generator for the Poseidon parameter sets the reference's workload runs on.  BUILD CONTAINER
ONLY: it reads /root/reference as text, which does not exist on the GPU box.

The withdraw circuit hashes with `Bn254x3 / x4 / x5` (gadgets/src/poseidon/bn254_x{3,4,5}.rs): FULL_ROUNDS = 8,
PARTIAL_ROUNDS = 55 / 56 / 56, WIDTH = 3 / 4 / 5, constants turned into field elements by `parse_vec`
(gadgets/src/poseidon/mod.rs:12-23), which is restated here: the FIRST TWO CHARACTERS of every 64-digit string are
dropped, the remaining 62 digits are 31 bytes, and those bytes are read LITTLE-endian.  What is stored is the result of
that function -- the field elements the reference computes with -- not the text of its files:

  tests/golden/poseidon_bn254.npz   rc_x3, mds_x3, rc_x4, ... : canonical values as (k, 4) little-endian uint64 limbs
  tests/golden/poseidon_bn254.json  rounds, input seeds and the SHA-256 digests of the hashes / per-round states that
                                     oracle/poseidon.py (the restated plonk-hashing spec.rs) produces on them

"Parity unpinned": the reference holds no known-answer vector for its hash; these digests pin the GPU kernel to the
oracle on the reference's parameters, and the oracle to itself over time."""
import hashlib
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/gadgets/src/poseidon"
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617   # BN254 Fr


def parse_vec(strings):
    """gadgets/src/poseidon/mod.rs:12-23: hex::decode(&x[2..]) -> BigUint::from_bytes_le -> F::from_repr."""
    out = []
    for x in strings:
        v = int.from_bytes(bytes.fromhex(x[2:]), "little")
        assert v < R, "from_repr would refuse it"
        out.append(v)
    return out


def read_parameter_file(path):
    text = open(path).read()
    num = lambda name: int(re.search(r"const %s: usize = (\d+);" % name, text).group(1))
    body = lambda name: re.search(r"const %s: [^=]*= &\[(.*?)\n\];" % name, text, re.S).group(1)
    rc = parse_vec(re.findall(r'"([0-9A-Fa-f]{64})"', body("ROUND_CONSTANTS")))
    rows = re.findall(r"&\[(.*?)\]", body("MDS_MATRIX"), re.S)
    mds = [parse_vec(re.findall(r'"([0-9A-Fa-f]{64})"', r)) for r in rows]
    return num("WIDTH"), num("FULL_ROUNDS"), num("PARTIAL_ROUNDS"), rc, mds


def limbs(vals):
    return np.array([[(v >> (64 * i)) & (2 ** 64 - 1) for i in range(4)] for v in vals], dtype=np.uint64)


def main():
    from oracle import poseidon as OP
    from helpers import field_elems, digest
    arrays, meta = {}, {}
    for w in (3, 4, 5):
        width, full, partial, rc, mds = read_parameter_file(os.path.join(REF, "bn254_x%d.rs" % w))
        assert width == w and len(mds) == w and all(len(r) == w for r in mds) and full % 2 == 0
        assert len(rc) >= w * (full + partial)                       # constants.rs:59-62
        rc = rc[:w * (full + partial)]
        arrays["rc_x%d" % w] = limbs(rc)
        arrays["mds_x%d" % w] = limbs([x for row in mds for x in row])
        tag = (1 << (w - 1)) - 1                                     # constants.rs:64-65
        batch, seed = 96, 4200 + w
        hashes, states = [], []
        for b in range(batch):
            ins = field_elems(R, seed + b, w - 1)
            h, trace = OP.permute(R, w, full // 2, partial, rc, mds, tag, ins)
            hashes.append(h)
            states.extend(x for row in trace for x in row)
        meta["x%d" % w] = {"width": w, "full_rounds": full, "partial_rounds": partial, "domain_tag": tag, "batch": batch,
                           "arity": w - 1, "input_seed": seed,
                           "params_sha256": hashlib.sha256(arrays["rc_x%d" % w].tobytes() + arrays["mds_x%d" % w].tobytes()).hexdigest(),
                           "hashes_sha256": digest(hashes), "states_sha256": digest(states), "hash0": "%x" % hashes[0]}
    np.savez_compressed(os.path.join(HERE, "poseidon_bn254.npz"), **arrays)
    with open(os.path.join(HERE, "poseidon_bn254.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
