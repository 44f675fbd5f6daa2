"""Multi-GPU layer: one process per GPU, independent proofs sharded across ranks (SURVEY.md section 8e,
"the MSMs of one proof / of concurrent proofs are independent units").  There is no data-path collective:
torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only for the barrier and for the
max-over-ranks timing that bench.py reports."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) share of `total` independent units for `rank`; sizes differ by at most one."""
    assert 0 <= rank < world
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def barrier(dist) -> None:
    if dist is not None and dist.is_initialized():
        dist.barrier()


def max_over_ranks(dist, value: float, device=None) -> float:
    """Whole-job time = the slowest rank's time."""
    if dist is None or not dist.is_initialized():
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_proofs(dist, proofs: Sequence[bytes], device=None) -> List[bytes]:
    """Collect every rank's proofs on all ranks, in global unit order (fixed-size records)."""
    if dist is None or not dist.is_initialized():
        return list(proofs)
    import torch
    world = dist.get_world_size()
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([len(proofs)], dtype=torch.int64, device=device))
    size = len(proofs[0]) if proofs else 0
    sz = torch.tensor([size], dtype=torch.int64, device=device)
    dist.all_reduce(sz, op=dist.ReduceOp.MAX)
    size = int(sz.item())
    mx = max(int(c.item()) for c in counts)
    buf = torch.zeros(mx * size, dtype=torch.uint8, device=device)
    flat = b"".join(proofs)
    if flat:
        buf[:len(flat)] = torch.frombuffer(bytearray(flat), dtype=torch.uint8).to(buf.device)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out: List[bytes] = []
    for r in range(world):
        raw = bytes(bufs[r].cpu().numpy().tobytes())
        for k in range(int(counts[r].item())):
            out.append(raw[k * size:(k + 1) * size])
    return out
