"""The RCCL transport (libzkt_comm_rccl.so) on the GPU box, in a process of its own (started by tests/test_gpu_sharded.py;
not a test module): no torch in the process, RCCL and the HIP runtime are /opt/rocm's, as in a Rust host.
  direct   the vtable's all_gather called the way the library calls it, world of one
  sharded  ONE proof over 2 thread-ranks on this GPU whose communicators report device_buffers = 1: every exchange of the
           sharded prover goes through capi.hip comm_all_gather_dev's device branch into a real ncclAllGather"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def direct():
    import zkt_plonk_amd as z
    from zkt_plonk_amd import parallel as par
    assert "torch" not in sys.modules
    ctx = z.Context("bn254", 0)
    comm = par.RcclComm(par.RcclComm.unique_id(), 0, 1, 0)
    assert (comm.vt.rank, comm.vt.world, comm.vt.device_buffers) == (0, 1, 1)
    data = np.arange(1 << 16, dtype=np.uint64)
    d_a, d_b = ctx.alloc(data.nbytes), ctx.alloc(data.nbytes)
    ctx.upload(d_a, data)
    comm.all_gather(d_a, d_a, data.nbytes, True)                      # RCCL's in-place form (send == recv + rank * bytes)
    assert np.array_equal(ctx.download(d_a, data.shape), data)
    ctx.upload(d_b, np.zeros_like(data))
    comm.all_gather(d_a, d_b, data.nbytes, True)                      # out of place
    assert np.array_equal(ctx.download(d_b, data.shape), data)
    half = data.nbytes // 2
    comm.all_gather(d_a + 64, d_a, half, True)                        # overlapping, not in place: staged
    assert np.array_equal(ctx.download(d_a, (half // 8,)), data[8:8 + half // 8])
    host = (np.arange(4096, dtype=np.uint32) * 7).astype(np.uint32)
    recv = np.zeros_like(host)
    comm.all_gather(host.ctypes.data, recv.ctypes.data, host.nbytes, False)   # host buffers (the partial sums' path)
    assert np.array_equal(recv, host)
    # through the C-ABI's own plumbing check
    from zkt_plonk_amd import _lib
    assert _lib.comm_selftest(comm.vt, b"zkt-rccl") == b"zkt-rccl"
    comm.close()
    for d in (d_a, d_b):
        ctx.free(d)
    ctx.close()
    print("RCCL DIRECT OK", flush=True)


def sharded():
    import zkt_plonk_amd as z
    from zkt_plonk_amd import parallel as par
    from oracle import fields as F, plonk as P, coracle as K
    from helpers import field_elems, sharded_rank_job, sharded_exchange_bytes
    import threading
    cv = F.BN254
    cs = P.synthetic_circuit(cv, 3000, 64, seed=12, value_seed=4)
    n = cs.circuit_bound()
    srs_arr = K.srs_mont(cv, 0x7A57E, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    blinders = field_elems(cv.fr.p, 44, P.NUM_BLINDERS)
    want = P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders).serialize(cv)
    evals = {"evals": {k: K.fr_to_mont(cv, v) for k, v in P.setup_evals(be, cs).items()}}
    a, b, c = (K.fr_to_mont(cv, w) for w in cs.wire_evals(cs.n_gates))
    pos = sorted(cs.pi)
    job = (a, b, c, K.fr_to_mont(cv, cs.table), pos, K.fr_to_mont(cv, [cs.pi[i] for i in pos]), K.fr_to_mont(cv, blinders))
    for world, use_async in ((2, False), (2, True), (4, True)):
        # use_async: the communicator offers zkt_comm_vtable::all_gather_async, so the quotient exchange goes out in
        # ZKT_QUOTIENT_CHUNKS pieces on the library's communication stream (same bytes, more calls)
        group = par.RcclLocalGroup(world, 0, use_async=use_async)
        comms = [group.comm(r) for r in range(world)]
        out = [None] * world
        ths = [threading.Thread(target=sharded_rank_job, args=(z, par, comms[r], cv, n, srs_arr, evals, vk, [job], out, r, world, True))
               for r in range(world)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=600)
        for r in range(world):
            assert not isinstance(out[r], BaseException) and out[r] is not None, out[r]
            proofs, (calls, sent), (setup_calls, setup_sent) = out[r]
            assert proofs == [want], (world, r)
            pieces = 4 if use_async else 1                          # ZKT_QUOTIENT_CHUNKS
            assert calls - setup_calls == 4 + pieces and sent - setup_sent == sharded_exchange_bytes(cv, n, world, 1)
            assert comms[r].device_calls >= pieces, "the quotient exchange must have taken the device branch"
            assert comms[r].async_calls == (pieces if use_async else 0)
        for cm in comms:
            cm.close()
    print("RCCL SHARDED OK", flush=True)


if __name__ == "__main__":
    {"direct": direct, "sharded": sharded}[sys.argv[1]]()
