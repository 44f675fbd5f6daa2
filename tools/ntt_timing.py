"""Quick device-resident NTT timing (development aid; bench.py is the judged harness)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import zkt_plonk_amd as z

for curve in ("bls12_381", "bn254"):
    ctx = z.Context(curve, 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for log_n in (14, 16, 18, 20, 22, 24):
        n = 1 << log_n
        x = torch.randint(0, 1 << 61, (n, 4), dtype=torch.int64, device="cuda")
        y = torch.empty_like(x)
        for inv, cos in ((0, 0), (1, 1)):
            for _ in range(3):
                ctx.ntt_dev(log_n, x.data_ptr(), n, y.data_ptr(), inverse=bool(inv), coset=bool(cos))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                ctx.ntt_dev(log_n, x.data_ptr(), n, y.data_ptr(), inverse=bool(inv), coset=bool(cos))
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print("%s log_n=%d inv=%d coset=%d: %.3f ms  %.1f GB/s  %.2f Gbutterfly/s" % (
                curve, log_n, inv, cos, ms, 64.0 * n / ms / 1e6, (n / 2) * log_n / ms / 1e6), flush=True)
    ctx.close()
