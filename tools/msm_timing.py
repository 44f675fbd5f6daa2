"""Quick device-resident MSM timing (development aid)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import zkt_plonk_amd as z

for curve, logs in (("bn254", (14, 18, 20)), ("bls12_381", (20,))):
    ctx = z.Context(curve, 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for log_n in logs:
        n = 1 << log_n
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.srs_generate(12345, n); e1.record(); torch.cuda.synchronize()
        print("%s srs_generate+table 2^%d: %.1f ms %s" % (curve, log_n, e0.elapsed_time(e1), ctx.msm_info()), flush=True)
        x = torch.randint(0, 1 << 61, (n, 4), dtype=torch.int64, device="cuda")
        for _ in range(2):
            ctx.msm_enqueue_dev(x.data_ptr(), n)
        torch.cuda.synchronize()
        reps = 5
        e0.record()
        for _ in range(reps):
            ctx.msm_enqueue_dev(x.data_ptr(), n)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%s msm 2^%d: %.3f ms  %.2f Mpoints/s" % (curve, log_n, ms, n / ms / 1e3), flush=True)
    ctx.close()
