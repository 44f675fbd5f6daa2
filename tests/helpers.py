"""Shared helpers for the test-suite: deterministic inputs (splitmix64) and golden decoding."""
import hashlib

M64 = (1 << 64) - 1


def splitmix64(seed):
    s = seed & M64
    while True:
        s = (s + 0x9E3779B97F4A7C15) & M64
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        yield z ^ (z >> 31)


def field_elems(p, seed, count):
    g = splitmix64(seed)
    out = []
    for _ in range(count):
        v = 0
        for i in range(4):
            v |= next(g) << (64 * i)
        out.append(v % p)
    return out


def digest(vals):
    return hashlib.sha256(b"".join(int(v).to_bytes(32, "little") for v in vals)).hexdigest()


def unhex_point(p):
    return None if p is None else (int(p[0], 16), int(p[1], 16))


def rand_fr(rng, n):
    """n valid field elements as (n, 4) uint64 Montgomery words: any value below 2^253 is below both
    scalar moduli, so random limbs with the top limb < 2^61 are canonical representatives."""
    import numpy as np
    x = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    x[:, 3] >>= np.uint64(3)
    return x
