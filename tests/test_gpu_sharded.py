"""One proof across several GPUs (BASELINE.json configs[4]) rehearsed on ONE GPU: the ranks are
contexts of one process (threads, parallel.LocalGroup) or processes over gloo; every rank's proof bytes must equal the
oracle's (= the single-GPU bytes).  The class transform the sharding rests on is pinned to the oracle on its own."""
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import fields as F, plonk as P, coracle as K
from helpers import field_elems, rand_fr, run_sharded_ranks, sharded_exchange_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CURVES = [F.BN254, F.BLS12_381]


@pytest.mark.parametrize("cv", CURVES, ids=lambda c: c.name)
def test_class_transform_is_the_decimated_coset_transform(cv):
    """zkt_ntt_class: out[i] = p(g w_big^(cls + G i)) -- every class of every split of the coset transform, ragged and
    over-long (folded) inputs, single-workgroup and multi-pass sizes."""
    import zkt_plonk_amd as z
    ctx = z.Context(cv.name, 0)
    rng = np.random.default_rng(5)
    for log_big, lens in ((7, (1, 40, 32 + 8)), (12, (1000, 1024 + 8)), (14, (4096 + 8,)), (18, (65536 + 8,))):
        for in_len in lens:
            x = rand_fr(rng, in_len)
            full = K.ntt_mont(cv, log_big, False, True, x)
            for lg in (1, 2, 3):
                G = 1 << lg
                classes = range(G) if log_big <= 14 else (0, G - 1)
                for cls in classes:
                    got = ctx.ntt_class(log_big - lg, log_big, cls, x)
                    assert np.array_equal(got, full[cls::G]), (log_big, in_len, G, cls)
    with pytest.raises(z.ZktError):
        ctx.ntt_class(5, 7, 4, rand_fr(rng, 8))        # class index outside the split
    # the committed fixtures (tests/golden/vectors_r02.json)
    import json
    from helpers import digest
    with open(os.path.join(ROOT, "tests", "golden", "vectors_r02.json")) as f:
        fx = json.load(f)[cv.name]["ntt_class"]
    for e in fx:
        lg = e["G"].bit_length() - 1
        x = K.fr_to_mont(cv, field_elems(cv.fr.p, e["seed"], e["in_len"]))
        assert digest(K.fr_from_mont(cv, ctx.ntt_class(e["log_big"] - lg, e["log_big"], e["cls"], x))) == e["sha256"]
    ctx.close()


def _setup_oracle(cv, cs, tau):
    n = cs.circuit_bound()
    srs_arr = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    return n, srs_arr, be, pk, epk, vk


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("cvname,gates,table_size", [("bn254", 60, 16), ("bn254", 4000, 1024), ("bls12_381", 1000, 64),
                                                     ("bn254", 60000, 1024)])
def test_sharded_proof_bytes_equal_the_oracle(world, cvname, gates, table_size):
    """`world` contexts on one GPU play the ranks (threads + in-process all-gather): SRS slices, sharded setup (the
    VerifierKey commitments through the partial-sum exchange), two chained proofs with distinct witnesses.  Bytes on
    every rank == oracle (oracle.plonk up to n = 4096, the array prover beyond)."""
    import zkt_plonk_amd as z
    from zkt_plonk_amd import parallel as par
    from oracle import fastplonk as FP
    cv = F.CURVES[cvname]
    css = [P.synthetic_circuit(cv, gates, table_size, seed=gates + 3, value_seed=k, n_public=3 if gates < 100 else 20)
           for k in (1, 2)]
    cs = css[0]
    tau = 0xABC0 + world + gates
    n = cs.circuit_bound()
    srs_arr = K.srs_mont(cv, tau, n + 8)
    be = K.CBackend(cv, srs_arr)
    mont = lambda v: K.fr_to_mont(cv, v) if len(v) else np.zeros((0, 4), dtype=np.uint64)
    evals = {k: mont(v) for k, v in P.setup_evals(be, cs).items()}
    blinders = [field_elems(cv.fr.p, 900 + k, P.NUM_BLINDERS) for k in range(2)]
    if n <= 4096:
        pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
        want = [P.prove(be, [None] * (n + 8), pk, epk, vk, c_, P.new_seeded_transcript(cv, vk), b).serialize(cv)
                for c_, b in zip(css, blinders)]
    else:
        keys = FP.setup(cv, srs_arr, n.bit_length() - 1, evals)
        vk = keys.verifier_key(cv, cs.pi.keys())
        want = [FP.prove(cv, srs_arr, keys, *[mont(w) for w in c_.wire_evals(c_.n_gates)], mont(c_.table), c_.pi,
                         P.new_seeded_transcript(cv, vk), b) for c_, b in zip(css, blinders)]
    jobs = []
    for c_, b in zip(css, blinders):
        a, b_, c = (mont(w) for w in c_.wire_evals(c_.n_gates))
        pos = sorted(c_.pi)
        jobs.append((a, b_, c, mont(c_.table), pos, mont([c_.pi[i] for i in pos]), mont(b)))
    out = run_sharded_ranks(z, cv, n, srs_arr, {"evals": evals}, vk, jobs, world)
    xyzz = 4 * cv.fq.limbs64 * 8
    for r in range(world):
        proofs, (calls, sent), (setup_calls, setup_sent) = out[r]
        assert proofs == want, r
        # setup: the ten VerifierKey commitments in 2 exchanges; per proof: 4 partial-sum exchanges + the quotient exchange,
        # which moves exactly 4n * 32 / world bytes per rank
        assert (setup_calls, setup_sent) == (2, 10 * xyzz)
        assert calls == 2 + 5 * len(jobs)
        assert sent - setup_sent == sharded_exchange_bytes(cv, n, world, len(jobs))


def test_two_rank_rehearsal_over_gloo_processes():
    """The same path with real processes and torch.distributed (gloo) between them: tests/sharded_worker.py is started
    twice by torch.distributed.run; both ranks print the digest of their proof, which must be the oracle's."""
    port = socket.socket()
    port.bind(("127.0.0.1", 0))
    pnum = port.getsockname()[1]
    port.close()
    env = dict(os.environ, ZKT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(pnum), os.path.join(ROOT, "tests", "sharded_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import re
    found = re.findall(r"SHARDED (\d) (OK|BAD) ([0-9a-f]{64})", r.stdout)      # the two ranks may share a line
    assert sorted(f[0] for f in found) == ["0", "1"] and all(f[1] == "OK" for f in found), r.stdout[-2000:]
    assert found[0][2] == found[1][2]


def test_torch_comm_device_path_on_one_rank():
    """parallel.TorchComm's device branch (raw HBM pointers dressed as tensors, in-place all_gather_into_tensor over
    RCCL) cannot meet a second GPU here; with a world of one it must at least leave the library's buffer intact and
    see the bytes the library wrote.  Runs in its own process: torch's HIP runtime has to come up before the library's
    (as in bench.py), which the pytest process can no longer guarantee at this point."""
    port = socket.socket()
    port.bind(("127.0.0.1", 0))
    pnum = port.getsockname()[1]
    port.close()
    env = dict(os.environ, ZKT_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(pnum), os.path.join(ROOT, "tests", "sharded_worker.py"), "devcomm"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DEVCOMM OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("leg", ["direct", "sharded"])
def test_rccl_transport_of_the_c_abi(leg):
    """libzkt_comm_rccl.so (include/zkt_comm_rccl.h), the transport a host without torch links: `direct` calls the vtable's
    all_gather with a world of one the way the library does (in place, out of place, overlapping, host buffers); `sharded`
    proves ONE proof over 2 and 4 thread-ranks on this GPU whose communicators have device_buffers = 1, so that every exchange
    of the sharded prover passes device pointers through capi.hip's device branch into a real ncclAllGather on the context's
    stream (bytes == the CPU oracle's proof, exchange counts and sizes as designed).  In a process of its own, without
    torch: RCCL and HIP are /opt/rocm's, as in a Rust host.  More than one RCCL rank remains UNVERIFIED ON HARDWARE."""
    from conftest import rccl_transport_or_skip
    rccl_transport_or_skip()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ZKT_SYSTEM_ROCM="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), leg], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "RCCL %s OK" % leg.upper() in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_sharded_mode_refuses_a_key_that_is_not_the_ranks_slice():
    """With a communicator attached, the whole key (zkt_srs_load) or another rank's slice would turn every combined
    commitment into a multiple of the right one: setup and prove refuse before any collective is entered."""
    import zkt_plonk_amd as z
    from zkt_plonk_amd import parallel as par
    cv = F.BN254
    cs = P.synthetic_circuit(cv, 100, 16, seed=2)
    n = cs.circuit_bound()
    srs_arr = K.srs_mont(cv, 77, n + 8)
    be = K.CBackend(cv, srs_arr)
    evals = {k: K.fr_to_mont(cv, v) for k, v in P.setup_evals(be, cs).items()}
    ctx = z.Context(cv.name, 0)
    comm = par.LocalGroup(2).comm(0)          # the partner never shows up: no collective may be reached
    ctx.set_comm(comm)
    ctx.srs_load(srs_arr)                                        # the whole key
    with pytest.raises(z.ZktError) as e:
        z.GpuProver.setup(ctx, n.bit_length() - 1, evals)
    assert e.value.code == 1 and comm.calls == 0
    lo, hi = par.shard_range(n + 8, 1, 2)
    ctx.srs_load_slice(srs_arr[lo:hi], lo, n + 8)                # rank 1's slice on rank 0
    with pytest.raises(z.ZktError) as e:
        z.GpuProver.setup(ctx, n.bit_length() - 1, evals)
    assert e.value.code == 1 and comm.calls == 0
    ctx.set_comm(None)                                           # detached: an ordinary single-GPU context again
    ctx.srs_load(srs_arr)
    pk, epk, vk = P.setup(be, [None] * (n + 8), cs, True)
    z.GpuProver.setup(ctx, n.bit_length() - 1, evals)
    a, b, c = (K.fr_to_mont(cv, w) for w in cs.wire_evals(cs.n_gates))
    pos = sorted(cs.pi)
    blinders = field_elems(cv.fr.p, 1, P.NUM_BLINDERS)
    tr = z.seed_transcript(z.Transcript("merlin", "ZKT Plonk"), vk.n, vk.commits)
    got = ctx.prove(a, b, c, K.fr_to_mont(cv, cs.table), pos, K.fr_to_mont(cv, [cs.pi[i] for i in pos]),
                    K.fr_to_mont(cv, blinders), tr)
    assert got == P.prove(be, [None] * (n + 8), pk, epk, vk, cs, P.new_seeded_transcript(cv, vk), blinders).serialize(cv)
    ctx.close()


def test_library_and_torch_share_one_hip_runtime_in_either_import_order():
    """PyTorch ships its own libamdhip64; loaded second it used to find no device.  The binding maps torch's copy first,
    so a context created BEFORE torch touches the GPU leaves torch fully functional (fresh process)."""
    code = "\n".join([
        "import sys; sys.path.insert(0, %r)" % ROOT,
        "import numpy as np",
        "import zkt_plonk_amd as z",
        "ctx = z.Context('bn254', 0)",
        "x = ctx.ntt(4, np.ones((16, 4), dtype=np.uint64))",
        "import torch",
        "torch.cuda.set_device(0)",
        "t = torch.arange(8, device='cuda').sum().item()",
        "y = ctx.ntt(4, np.ones((16, 4), dtype=np.uint64))",
        "assert t == 28 and np.array_equal(x, y)",
        "print('SHARED OK')"])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARED OK" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
