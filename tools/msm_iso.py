"""One MSM at a time (device synchronised between launches): per-kernel durations without the overlap of the previous
MSM's bucket reduction on the side stream.  usage: rocprofv3 --kernel-trace --stats -- python3 tools/msm_iso.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import zkt_plonk_amd as z
curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = z.Context(curve, 0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << log_n
ctx.srs_generate(12345, n + 8)
x = torch.randint(0, 1 << 61, (n, 4), dtype=torch.int64, device="cuda")
for _ in range(8):
    ctx.msm_enqueue_dev(x.data_ptr(), n)
    torch.cuda.synchronize()
ctx.close()
