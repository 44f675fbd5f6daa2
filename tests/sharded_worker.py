"""One rank of the two-process sharded-proof rehearsal (started by tests/test_gpu_sharded.py through
torch.distributed.run; not a test module).  Prints `SHARDED <rank> OK|BAD <sha256>`."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    import zkt_plonk_amd as z
    from zkt_plonk_amd import parallel as par
    from oracle import fields as F, plonk as P, coracle as K
    from helpers import field_elems
    backend = os.environ.get("ZKT_DIST_BACKEND", "gloo")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)
    if len(sys.argv) > 1 and sys.argv[1] == "devcomm":
        return devcomm(torch, dist, z, par, dev)
    cv = F.BN254
    cs = P.synthetic_circuit(cv, 3000, 64, seed=12, value_seed=4)
    n = cs.circuit_bound()
    srs_arr = K.srs_mont(cv, 0x7A57E, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    blinders = field_elems(cv.fr.p, 44, P.NUM_BLINDERS)
    want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders).serialize(cv)
    ctx = z.Context(cv.name, dev)
    ctx.set_comm(par.TorchComm(dist, torch.device("cuda", dev)))
    lo, hi = par.shard_range(n + 8, rank, world)
    ctx.srs_load_slice(srs_arr[lo:hi], lo, n + 8)
    z.GpuProver(ctx, n.bit_length() - 1, {k: K.fr_to_mont(cv, pk.polys[k]) for k in z.PK_ORDER})
    a, b, c = (K.fr_to_mont(cv, w) for w in cs.wire_evals(cs.n_gates))
    pos = sorted(cs.pi)
    tr = z.seed_transcript(z.Transcript("merlin", "ZKT Plonk"), vk.n, vk.commits)
    got = ctx.prove(a, b, c, K.fr_to_mont(cv, cs.table), pos, K.fr_to_mont(cv, [cs.pi[i] for i in pos]),
                    K.fr_to_mont(cv, blinders), tr)
    print("SHARDED %d %s %s" % (rank, "OK" if got == want else "BAD", hashlib.sha256(got).hexdigest()), flush=True)
    ctx.close()
    dist.destroy_process_group()
    sys.exit(0 if got == want else 1)


def devcomm(torch, dist, z, par, dev):
    """TorchComm's device branch with a world of one (RCCL): the library's HBM buffer dressed as a tensor, gathered in
    place, must come back intact; the host branch bounces through a device tensor."""
    device = torch.device("cuda", dev)
    comm = par.TorchComm(dist, device)
    assert comm.vt.device_buffers == 1
    ctx = z.Context("bn254", dev)
    data = np.arange(4096, dtype=np.uint64)
    d = ctx.alloc(data.nbytes)
    ctx.upload(d, data)
    assert comm._all_gather(None, d, d, data.nbytes, 1, None) == 0          # in place, as the prover calls it
    assert np.array_equal(ctx.download(d, data.shape), data)
    seen = torch.as_tensor(par._DevBuf(d, data.nbytes), device=device).cpu().numpy().view(np.uint64)
    assert np.array_equal(seen, data)
    host = (np.arange(192, dtype=np.uint8) * 3).astype(np.uint8)
    recv = np.zeros(192, dtype=np.uint8)
    assert comm._all_gather(None, host.ctypes.data, recv.ctypes.data, 192, 0, None) == 0
    assert np.array_equal(recv, host)
    ctx.free(d)
    ctx.close()
    dist.destroy_process_group()
    print("DEVCOMM OK", flush=True)


if __name__ == "__main__":
    main()
