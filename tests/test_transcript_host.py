"""CPU: the product's own host-side transcripts (C++ in libzkt_plonk_hip.so) against the published merlin
vector, the reference's EthereumTranscript KAT, and the oracle on random traffic."""
import random

import zkt_plonk_amd as z
from oracle import fields as F, transcript as T, curve as C


def test_merlin_conformance_vector():
    t = z.Transcript("merlin", "test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_ethereum_reference_kat():
    # gadgets/src/transcript.rs:101-127
    e = z.Transcript("ethereum", "test")
    e.append_u64("a", 1)
    assert e.challenge_scalar("a").to_bytes(32, "big").hex() == "0f9d11cec4f06b0d18060cde3db4196495ddfbb096108951446fc8a1d45f4b59"
    e.append_scalar("b", 2)
    assert e.challenge_scalar("b").to_bytes(32, "big").hex() == "0f4dccb919a5dba2dd010a562ba45b4551291f5e565706536e78b24ac8b5c64d"
    e.append_commitment("c", (3, 4))
    assert e.challenge_scalar("c").to_bytes(32, "big").hex() == "1b5bf46adfcd1dd4f9ac7166586cf83f261192bc4b83fdda30ddee22f9054c1f"


def test_random_traffic_matches_oracle():
    rnd = random.Random(3)
    for cv in (F.BN254, F.BLS12_381):
        kinds = ["merlin"] + (["ethereum"] if cv.name == "bn254" else [])
        for kind in kinds:
            mine = z.Transcript(kind, "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
            ref = T.MerlinTranscript(cv, "ZKT Plonk") if kind == "merlin" else T.EthereumTranscript(cv)
            G = C.generator(cv)
            for step in range(60):
                op = rnd.randrange(5)
                if op == 0:
                    v = rnd.randrange(1 << 64)
                    mine.append_u64("circuit_size", v); ref.append_u64("circuit_size", v)
                elif op == 1:
                    v = rnd.randrange(cv.fr.p)
                    mine.append_scalar("a_eval", v); ref.append_scalar("a_eval", v)
                elif op == 2:
                    vs = [rnd.randrange(cv.fr.p) for _ in range(rnd.randrange(0, 5))]
                    mine.append_scalars("pi", vs); ref.append_scalars("pi", vs)
                elif op == 3:
                    P = None if rnd.randrange(6) == 0 else C.scalar_mul(cv, rnd.randrange(1, 1 << 40), G)
                    mine.append_commitment("a_commit", P); ref.append_commitment("a_commit", P)
                else:
                    assert mine.challenge_scalar("beta") == ref.challenge_scalar("beta")
            assert mine.challenge_scalar("eta") == ref.challenge_scalar("eta")


def test_seed_in_one_call_equals_the_eleven_appends():
    """zkt_transcript_seed = VerifierKey::seed_transcript (keys/mod.rs:260-275): circuit_size and the ten commitments."""
    rnd = random.Random(8)
    for cv in (F.BN254, F.BLS12_381):
        kinds = ["merlin"] + (["ethereum"] if cv.name == "bn254" else [])
        g = (cv.gx, cv.gy)
        pts = {}
        for i, name in enumerate(z.PK_ORDER):
            pts[name] = None if i == 7 else C.scalar_mul(cv, rnd.randrange(1, cv.fr.p), g)   # one identity among them
        for kind in kinds:
            a = z.Transcript(kind, "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
            z.seed_transcript(a, 1 << 14, pts)
            b = z.Transcript(kind, "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
            b.append_u64("circuit_size", 1 << 14)
            for name in z.PK_ORDER:
                b.append_commitment(name + "_commit", pts[name])
            assert a.challenge_scalar("x") == b.challenge_scalar("x")
