"""Host verifier timing (zkt_verify = prepare + two pairing products), n = 256 proof made by the oracle.  usage: python tools/verify_timing.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import time, numpy as np, sys
sys.path.insert(0,'tests')
from oracle import fields as F, plonk as P, coracle as K, curve as C, pairing as PR
from helpers import field_elems
import zkt_plonk_amd as z
from zkt_plonk_amd import _lib
from test_pairing_host import g2_mont
for cv in (F.BN254, F.BLS12_381):
    T = PR.Tower(cv)
    cs = P.synthetic_circuit(cv, 150, 16, seed=77, n_public=3)
    n = cs.circuit_bound(); tau = 0xBEEFCAFE
    srs = K.srs_mont(cv, tau, n + 8); be = K.CBackend(cv, srs)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    proof = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk, "merlin"), field_elems(cv.fr.p, 8, P.NUM_BLINDERS)).serialize(cv)
    pis = [cs.pi[k] for k in sorted(cs.pi)]
    H = PR.G2_GENERATORS[cv.name]
    h, beta_h = g2_mont(cv, [H])[0], g2_mont(cv, [T.g2_mul(tau, H)])[0]
    commits = K.points_to_mont(cv, [vk.commits[k] for k in z.PK_ORDER]); inf = [vk.commits[k] is None for k in z.PK_ORDER]
    roots=K.fr_to_mont(cv, vk.pi_roots); pub=K.fr_to_mont(cv, pis)
    def tr():
        t = z.Transcript("merlin", "ZKT Plonk", fr_bits=cv.fr.bits, fq_bytes=cv.fq.limbs64 * 8)
        z.seed_transcript(t, vk.n, vk.commits); return t
    best=[1e9,1e9,1e9]
    for _ in range(8):
        t_=tr(); t=time.perf_counter(); ok=_lib.verify(cv.name, vk.n, commits, inf, roots, pub, proof, srs[0], h, beta_h, t_); best[0]=min(best[0],(time.perf_counter()-t)*1e3)
        t_=tr(); t=time.perf_counter(); pairs,infs=_lib.verify_prepare(cv.name, vk.n, commits, inf, roots, pub, proof, srs[0], t_); best[1]=min(best[1],(time.perf_counter()-t)*1e3)
        L=cv.fq.limbs64
        g1=np.stack([pairs[0], pairs[1]]); g2=np.stack([h,beta_h])
        t=time.perf_counter(); _lib.pairing_product_is_one(cv.name, g1, g2); best[2]=min(best[2],(time.perf_counter()-t)*1e3)
    print(cv.name, "verify %.2f  prepare %.2f  one 2-pairing product %.2f ms"%tuple(best), ok)
