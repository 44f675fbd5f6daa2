import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import zkt_plonk_amd as z
curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 22
ctx = z.Context(curve, 0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n = 1 << log_n
x = torch.randint(0, 1 << 61, (n, 4), dtype=torch.int64, device="cuda")
y = torch.empty_like(x)
for _ in range(3):
    ctx.ntt_dev(log_n, x.data_ptr(), n // 4 + 3, y.data_ptr(), inverse=False, coset=True)
torch.cuda.synchronize()
ctx.close()
