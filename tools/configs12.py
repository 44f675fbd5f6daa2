"""BASELINE.json configs[1] (BLS12-381 radix-2 NTT over Fr at n = 2^20, bit-exact coefficients) and configs[2] (KZG10 G1 MSM,
2^20 random scalars / points, bit-exact commitment) as standalone lines: device time (HIP events, data resident in HBM), the
CPU oracle's time on the host cores of the same box (oracle/coracle.cpp: ports of ark-poly's radix-2 FFT and ark-ec's
Pippenger, OpenMP) and the bit-for-bit comparison of the two results.  Measurement aid (uses the oracle as the checker and
the reported-only CPU baseline, like bench.py's cpu_baseline leg).  usage (GPU box): python tools/configs12.py > gpurun_out/configs12.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import zkt_plonk_amd as z
from oracle import coracle as K, fields as F

dev = torch.device("cuda", 0)
rng = np.random.default_rng(2020)


def rand_fr(n):
    x = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    x[:, 3] >>= np.uint64(3)
    return x


def gpu_ms(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("host threads of the CPU oracle: %d" % K.num_threads())
log_n, n = 20, 1 << 20
for cv in (F.BLS12_381, F.BN254):
    ctx = z.Context(cv.name, 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    x = rand_fr(n)
    dx = torch.from_numpy(x.view(np.int64)).to(dev)
    dy = torch.empty_like(dx)
    for name, inv, cos in (("fft", 0, 0), ("ifft", 1, 0), ("coset_fft", 0, 1), ("coset_ifft", 1, 1)):
        ms = gpu_ms(lambda: ctx.ntt_dev(log_n, dx.data_ptr(), n, dy.data_ptr(), inverse=bool(inv), coset=bool(cos)), 20)
        got = dy.cpu().numpy().view(np.uint64)
        t = time.perf_counter()
        want = K.ntt_mont(cv, log_n, inv, cos, x)
        cpu_ms = 1e3 * (time.perf_counter() - t)
        print("configs[1] %-9s NTT %-10s n=2^20: GPU %.4f ms = %6.1f GB/s (64 N / t; %.2f %% of 8 TB/s), CPU oracle %8.1f ms (x%.0f), "
              "coefficients bit-exact: %s" % (cv.name, name, ms, 64.0 * n / ms / 1e6, 64.0 * n / ms / 1e6 / 80.0, cpu_ms, cpu_ms / ms,
                                              bool(np.array_equal(got, want))), flush=True)
    ctx.srs_generate(0x5EED, n)
    srs = ctx.srs_download(0, n)
    info = ctx.msm_info()
    sc = rand_fr(n)
    sc[::100] = 0
    sc[1::100] = K.fr_to_mont(cv, [1])[0]
    sc[2::1000] = K.fr_to_mont(cv, [cv.fr.p - 1])[0]
    ds = torch.from_numpy(sc.view(np.int64)).to(dev)
    ms = gpu_ms(lambda: ctx.msm_enqueue_dev(ds.data_ptr(), n), 10)
    out, inf = ctx.msm(sc)
    t = time.perf_counter()
    want, winf = K.msm_mont(cv, srs, sc)
    cpu_ms = 1e3 * (time.perf_counter() - t)
    c_ref = 15
    w_ref = -(-cv.fr.bits // c_ref)
    ref_adds = w_ref * n + 2 * w_ref * ((1 << c_ref) - 1)
    print("configs[2] %-9s MSM n=2^20 (1 %% zeros, 1 %% ones, 0.1 %% r-1): GPU %.3f ms per MSM back to back (digits c = %d, %d windows) = "
          "%.3e G1-adds/s by the reference-window formula (%.3e adds), %.1f Mpoints/s; CPU oracle %8.1f ms (x%.0f); commitment bit-exact: %s"
          % (cv.name, ms, info["window_bits"], info["windows"], ref_adds / ms * 1e3, ref_adds, n / ms / 1e3, cpu_ms, cpu_ms / ms,
             bool(inf == winf and np.array_equal(out, want))), flush=True)
    ctx.close()
