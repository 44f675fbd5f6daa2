"""Multi-GPU layer: one process per GPU, two ways to use the GPUs of a node.
(a) Independent proofs sharded across ranks (`shard_range`, `gather_proofs`): no data-path collective at all;
    torch.distributed (RCCL on GPUs, gloo in the CPU tests) only carries the barrier and the max-over-ranks timing that
    bench.py reports.
(b) ONE proof across the ranks (`TorchComm`, `LocalGroup`: the communicator behind zkt_ctx_set_comm): index-range MSMs
    whose partial sums are all-gathered as raw bytes (a collective cannot reduce curve points) and ONE all-gather of the
    quotient evaluations (4n x 32 B in total) per proof.  The device branch of TorchComm
    (all_gather_into_tensor on the library's HBM buffer over RCCL, the send slice copied out first) has run with a world of one only: the pool gives
    one GPU per box, so the multi-GPU device transport is UNVERIFIED ON HARDWARE; bench.py fails (exit 4) if its
    bytes ever differ from the single-GPU proof's.
`sharded_msm_combine` is the one exchange step of an MSM whose points are split by index range."""
from __future__ import annotations

import ctypes
import sys
import traceback
from typing import List, Sequence, Tuple


class _DevBuf:
    """A raw device pointer dressed up for torch.as_tensor (no copy)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class TorchComm:
    """zkt_comm_vtable over torch.distributed: the all-gather a proof sharded across the GPUs of a node needs
    (include/zkt_plonk.h "one proof across the GPUs of a node").  backend "nccl" = RCCL over xGMI: the quotient exchange
    moves device to device; the round's partial commitments (a few hundred bytes) bounce through a device tensor.
    backend "gloo" (CPU tests, several ranks rehearsing on one GPU): host buffers only, the library stages the device
    exchange through pinned memory."""

    def __init__(self, dist, device=None):
        import torch
        from ._lib import CommVtable, ALL_GATHER_CB
        self.dist = dist
        self.torch = torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.nccl = dist.get_backend() == "nccl"
        self.device = device
        self.calls = 0
        self._cb = ALL_GATHER_CB(self._all_gather)
        self.vt = CommVtable(None, self.rank, self.world, 1 if self.nccl else 0, self._cb)

    def _host(self, ptr, nbytes):
        return self.torch.frombuffer((ctypes.c_uint8 * nbytes).from_address(ptr), dtype=self.torch.uint8)

    def _all_gather(self, user, send, recv, nbytes, on_device, stream):
        try:
            torch, dist = self.torch, self.dist
            self.calls += 1
            if on_device:
                s = torch.as_tensor(_DevBuf(send, nbytes), device=self.device)
                r = torch.as_tensor(_DevBuf(recv, nbytes * self.world), device=self.device)
                if recv <= send < recv + nbytes * self.world:
                    # the library sends from inside its receive buffer (RCCL's in-place form).  Whether
                    # torch.distributed accepts aliased tensors there has never been seen on more than one GPU: give
                    # the send data a buffer of its own (one device-to-device copy of 4n * 32 / world bytes)
                    s = s.clone()
                dist.all_gather_into_tensor(r, s)
                torch.cuda.synchronize(self.device)          # contract: complete when the callback returns
                return 0
            s, r = self._host(send, nbytes), self._host(recv, nbytes * self.world)
            if self.nccl:
                rd = torch.empty(nbytes * self.world, dtype=torch.uint8, device=self.device)
                dist.all_gather_into_tensor(rd, s.to(self.device))
                r.copy_(rd.cpu())
            else:
                dist.all_gather([r[i * nbytes:(i + 1) * nbytes] for i in range(self.world)], s.clone())
            return 0
        except Exception:                                     # never let an exception cross the C boundary
            traceback.print_exc(file=sys.stderr)
            return 1


class LocalGroup:
    """Communicators for `world` contexts driven by threads of ONE process (one thread per GPU, or several contexts
    rehearsing on one GPU in the tests): an in-process all-gather over host buffers; the library stages the device
    exchange through pinned memory."""

    def __init__(self, world: int):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [b""] * world

    def comm(self, rank: int) -> "LocalComm":
        return LocalComm(self, rank)


class LocalComm:
    def __init__(self, group: LocalGroup, rank: int):
        from ._lib import CommVtable, ALL_GATHER_CB
        self.group, self.rank, self.world = group, rank, group.world
        self.calls = 0
        self._cb = ALL_GATHER_CB(self._all_gather)
        self.vt = CommVtable(None, rank, group.world, 0, self._cb)

    def _all_gather(self, user, send, recv, nbytes, on_device, stream):
        try:
            g = self.group
            self.calls += 1
            g.slots[self.rank] = ctypes.string_at(send, nbytes)
            g.barrier.wait(timeout=120)
            ctypes.memmove(recv, b"".join(g.slots), nbytes * self.world)
            g.barrier.wait(timeout=120)          # nobody overwrites a slot before everyone has read it
            return 0
        except Exception:
            traceback.print_exc(file=sys.stderr)
            return 1


_rccl_lib = None


def rccl_lib():
    """libzkt_comm_rccl.so (include/zkt_comm_rccl.h): the transport the C-ABI owns.  One copy of RCCL per process: when
    PyTorch is installed its librccl.so.1 is mapped first (by path, without importing torch), like its HIP runtime in
    _lib._share_torch_hip_runtime, so that this library, the prover and torch.distributed all resolve to the same one."""
    global _rccl_lib
    if _rccl_lib is None:
        import importlib.util
        import os
        from . import _lib
        _lib.lib()                                              # the HIP runtime first (torch's copy if there is one)
        try:
            spec = importlib.util.find_spec("torch")
        except (ImportError, ValueError):
            spec = None
        if spec is not None and spec.submodule_search_locations and not os.environ.get("ZKT_SYSTEM_ROCM"):
            path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
            if os.path.exists(path):
                try:
                    ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
                except OSError:
                    pass
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libzkt_comm_rccl.so")
        if not os.path.exists(path):
            raise ImportError("libzkt_comm_rccl.so is missing (%s): build it with __graft_entry__.build()" % path)
        L = ctypes.CDLL(path)
        L.zkt_comm_rccl_unique_id.argtypes = [ctypes.c_char_p]
        L.zkt_comm_rccl_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.zkt_comm_rccl_vtable.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.zkt_comm_rccl_destroy.argtypes = [ctypes.c_void_p]
        L.zkt_comm_rccl_destroy.restype = None
        L.zkt_comm_rccl_last_error.argtypes = [ctypes.c_void_p]
        L.zkt_comm_rccl_last_error.restype = ctypes.c_char_p
        _rccl_lib = L
    return _rccl_lib


class RcclComm:
    """zkt_comm_rccl (include/zkt_comm_rccl.h): ncclAllGather on the proving context's stream, no Python in the data path.
    `unique_id()` on one rank, the 128 bytes to every rank (any channel), then RcclComm(id, rank, world, device) on each --
    collective.  More than one rank has never run on hardware (one GPU per box in the development pool)."""
    UNIQUE_ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(RcclComm.UNIQUE_ID_BYTES)
        if rccl_lib().zkt_comm_rccl_unique_id(buf):
            raise RuntimeError("ncclGetUniqueId failed")
        return buf.raw

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int = 0):
        from ._lib import CommVtable
        assert len(unique_id) == self.UNIQUE_ID_BYTES
        L = rccl_lib()
        h = ctypes.c_void_p()
        rc = L.zkt_comm_rccl_create(unique_id, rank, world, device, ctypes.byref(h))
        if rc:
            raise RuntimeError("zkt_comm_rccl_create failed with status %d" % rc)
        self._h, self.rank, self.world = h, rank, world
        self.vt = CommVtable()
        if L.zkt_comm_rccl_vtable(self._h, ctypes.byref(self.vt)):
            raise RuntimeError("zkt_comm_rccl_vtable failed")

    def all_gather(self, send: int, recv: int, nbytes: int, on_device: bool, stream: int = 0) -> None:
        """The vtable's callback, called the way the library calls it (raw pointers)."""
        if self.vt.all_gather(self.vt.user, send, recv, nbytes, 1 if on_device else 0, stream):
            raise RuntimeError("rccl all_gather: %s" % rccl_lib().zkt_comm_rccl_last_error(self._h).decode())

    def close(self):
        if self._h:
            rccl_lib().zkt_comm_rccl_destroy(self._h)
            self._h = None


class RcclLocalGroup:
    """Rehearsal on ONE GPU of the device path a multi-GPU job takes: `world` contexts driven by threads of one process,
    communicator with device_buffers = 1.  Each rank's bytes enter the exchange through its OWN world-of-one RCCL
    communicator (ncclAllGather on the context's stream, device pointers straight from the library: capi.hip
    comm_all_gather_dev's device branch); what RCCL would carry over xGMI between the ranks is then copied device to device
    (the ranks share the GPU).  Not a substitute for a multi-GPU run: it checks the pointers, sizes, in-place rule and
    stream ordering the library hands a device transport, with a real RCCL call on them."""

    def __init__(self, world: int, device: int = 0, use_async: bool = True):
        import threading
        self.world, self.device, self.use_async = world, device, use_async
        self.barrier = threading.Barrier(world)
        self.recv = [0] * world
        self.host_slots = [b""] * world

    def comm(self, rank: int) -> "RcclLocalComm":
        return RcclLocalComm(self, rank)


class RcclLocalComm:
    def __init__(self, group: RcclLocalGroup, rank: int):
        from ._lib import CommVtable, ALL_GATHER_CB, ALL_GATHER_ASYNC_CB, lib
        self.group, self.rank, self.world = group, rank, group.world
        self.inner = RcclComm(RcclComm.unique_id(), 0, 1, group.device)
        self.calls = self.device_calls = self.async_calls = 0
        self._L = lib()
        self._cb = ALL_GATHER_CB(self._all_gather)
        # the stream-ordered entry (zkt_comm_vtable::all_gather_async): the library then sends the quotient exchange in
        # pieces on its communication stream.  The rehearsal cannot be asynchronous (the ranks meet at a host barrier);
        # what it checks is the pieces' pointers, sizes and order, and that the bytes still come out right
        self._cb_async = ALL_GATHER_ASYNC_CB(self._all_gather_async) if group.use_async else ALL_GATHER_ASYNC_CB()
        self.vt = CommVtable(None, rank, group.world, 1, self._cb, self._cb_async)

    def _all_gather_async(self, user, send, recv, nbytes, stream):
        self.async_calls += 1
        return self._all_gather(user, send, recv, nbytes, 1, stream)

    def _all_gather(self, user, send, recv, nbytes, on_device, stream):
        try:
            g = self.group
            self.calls += 1
            if not on_device:
                g.host_slots[self.rank] = ctypes.string_at(send, nbytes)
                g.barrier.wait(timeout=120)
                ctypes.memmove(recv, b"".join(g.host_slots), nbytes * self.world)
                g.barrier.wait(timeout=120)
                return 0
            self.device_calls += 1
            # my share into my slot of my receive buffer: a real ncclAllGather (world of one) on the context's stream
            self.inner.all_gather(send, recv + self.rank * nbytes, nbytes, True, stream or 0)
            g.recv[self.rank] = recv
            g.barrier.wait(timeout=120)
            hip = ctypes.CDLL("libamdhip64.so.7")                # by soname: the process's one HIP runtime, already mapped
            hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            for q in range(self.world):
                if q != self.rank:
                    rc = hip.hipMemcpy(recv + q * nbytes, g.recv[q] + q * nbytes, nbytes, 3)   # hipMemcpyDeviceToDevice
                    if rc:
                        raise RuntimeError("hipMemcpy D2D failed: %d" % rc)
            g.barrier.wait(timeout=120)
            return 0
        except Exception:
            traceback.print_exc(file=sys.stderr)
            try:
                self.group.barrier.abort()
            except Exception:
                pass
            return 1

    def close(self):
        self.inner.close()


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


def sharded_msm_combine(dist, curve, partial_xy, partial_is_inf: bool, device=None):
    """All-gather every rank's partial MSM result (affine, Montgomery limbs; identity = (0, 0)) and add them on the
    host: every rank returns the full sum -> (xy limbs, is_infinity).  SURVEY.md section 8e: rank g holds the SRS slice
    [g n/G, (g+1) n/G) and computes sum_i s_i P_i over it; 64-96 bytes per rank cross xGMI."""
    import numpy as np
    from ._lib import g1_sum_host
    mine = np.zeros_like(np.asarray(partial_xy, dtype=np.uint64)) if partial_is_inf else np.asarray(partial_xy, dtype=np.uint64)
    if dist is None or not dist.is_initialized():
        return g1_sum_host(curve, mine.reshape(1, -1))
    import torch
    t = torch.from_numpy(mine.view(np.int64).copy()).to(device) if device is not None else torch.from_numpy(mine.view(np.int64).copy())
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    pts = np.stack([p.cpu().numpy().view(np.uint64) for p in parts])
    return g1_sum_host(curve, pts)
