"""ctypes binding of libzkt_plonk_hip.so (the C-ABI declared in include/zkt_plonk.h)."""
from __future__ import annotations

import ctypes
import os
import re
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZKT_LIB_PATH: load another build of the same C-ABI (the host-side sanitizer build of tests/test_host_sanitize.py)
_LIB = os.environ.get("ZKT_LIB_PATH") or os.path.join(_HERE, "libzkt_plonk_hip.so")
_HEADER = os.path.join(_HERE, "..", "include", "zkt_plonk.h")

CURVE_BN254 = 0
CURVE_BLS12_381 = 1
_CURVES = {"bn254": 0, "bls12_381": 1, "bls12-381": 1, 0: 0, 1: 1}

_lib = None


class ZktError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("zkt error %d: %s" % (code, msg))
        self.code = code


def curve_id(curve) -> int:
    return _CURVES[curve]


def lib_path() -> str:
    return _LIB


def declared_symbols():
    """Every function the public header declares (used by the CPU test that checks the exports)."""
    text = open(_HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zkt_[a-z0-9_]+)\s*\(", text)))


def _share_torch_hip_runtime():
    """PyTorch's wheel carries its own libamdhip64.  Two HIP runtimes in one process do not share the device: whichever
    initialises second finds no GPU.  When torch is installed, its copy is mapped first (by path, without importing
    torch), so that this library and torch resolve to the same runtime whatever the import order."""
    import importlib.util
    if os.environ.get("ZKT_SYSTEM_ROCM"):       # a process that will never load torch (tests of the torch-free transport)
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Loads the HIP library.  Raises loudly when it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.environ.get("ZKT_LIB_PATH"):
            _share_torch_hip_runtime()
        if not os.path.exists(_LIB):
            raise ImportError(
                "libzkt_plonk_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'`; the HIP extension is mandatory, there is no CPU fallback." % _LIB)
        L = ctypes.CDLL(_LIB)
        vp, u64p, u32p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)
        L.zkt_version.restype = ctypes.c_char_p
        L.zkt_last_error.restype = ctypes.c_char_p
        L.zkt_last_error.argtypes = [vp]
        L.zkt_ctx_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
        L.zkt_ctx_destroy.argtypes = [vp]
        L.zkt_ctx_destroy.restype = None
        L.zkt_ctx_set_stream.argtypes = [vp, vp]
        L.zkt_ctx_synchronize.argtypes = [vp]
        L.zkt_profile_enable.argtypes = [vp, ctypes.c_int]
        L.zkt_profile_get.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double)]
        L.zkt_dev_alloc.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
        L.zkt_dev_free.argtypes = [vp, vp]
        L.zkt_dev_upload.argtypes = [vp, vp, vp, ctypes.c_size_t]
        L.zkt_dev_download.argtypes = [vp, vp, vp, ctypes.c_size_t]
        L.zkt_ntt.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, u64p, ctypes.c_size_t, u64p]
        L.zkt_ntt_dev.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]
        L.zkt_domain_group_gen.argtypes = [vp, ctypes.c_int, u64p]
        L.zkt_debug_params.argtypes = [vp, ctypes.c_int, u32p, ctypes.c_size_t]
        L.zkt_debug_fr_mul.argtypes = [vp, u64p, u64p, ctypes.c_size_t, u64p]
        L.zkt_debug_quotient.argtypes = [vp, u64p, ctypes.POINTER(ctypes.c_void_p), u64p, u64p, ctypes.c_size_t, u64p]
        _bind_optional(L)
        _bind_prover(L)
        _lib = L
    return _lib


def _bind_optional(L):
    vp, u64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)
    ip = ctypes.POINTER(ctypes.c_int)
    if hasattr(L, "zkt_srs_load"):
        L.zkt_srs_load.argtypes = [vp, u64p, ctypes.c_size_t]
        L.zkt_srs_load_dev.argtypes = [vp, vp, ctypes.c_size_t]
        L.zkt_srs_generate.argtypes = [vp, u64p, ctypes.c_size_t]
        L.zkt_srs_download.argtypes = [vp, ctypes.c_size_t, ctypes.c_size_t, u64p]
        L.zkt_msm_g1.argtypes = [vp, u64p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, u64p, ip]
        L.zkt_msm_g1_dev.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, vp]
        L.zkt_msm_enqueue_dev.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]
        L.zkt_msm_info.argtypes = [vp, ip, ip, ctypes.POINTER(ctypes.c_size_t)]


ALL_GATHER_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                 ctypes.c_int, ctypes.c_void_p)


ALL_GATHER_ASYNC_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


class CommVtable(ctypes.Structure):
    """zkt_comm_vtable (include/zkt_plonk.h): the caller's all-gather for a proof sharded across GPUs; all_gather_async
    (optional) is its stream-ordered form for device buffers."""
    _fields_ = [("user", ctypes.c_void_p), ("rank", ctypes.c_int), ("world", ctypes.c_int),
                ("device_buffers", ctypes.c_int), ("all_gather", ALL_GATHER_CB), ("all_gather_async", ALL_GATHER_ASYNC_CB)]


def shard_range(total: int, rank: int, world: int):
    lo, hi = ctypes.c_size_t(0), ctypes.c_size_t(0)
    L = lib()
    L.zkt_shard_range.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t),
                                  ctypes.POINTER(ctypes.c_size_t)]
    if L.zkt_shard_range(total, rank, world, ctypes.byref(lo), ctypes.byref(hi)):
        raise ValueError("zkt_shard_range(%d, %d, %d)" % (total, rank, world))
    return lo.value, hi.value


def comm_selftest(vt: CommVtable, send: bytes) -> bytes:
    """Gathers `send` from every rank through the vtable (host buffers): plumbing check, no GPU needed."""
    L = lib()
    L.zkt_comm_selftest.argtypes = [ctypes.POINTER(CommVtable), ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]
    recv = ctypes.create_string_buffer(len(send) * vt.world)
    rc = L.zkt_comm_selftest(ctypes.byref(vt), send, recv, len(send))
    if rc:
        raise ZktError(rc, "zkt_comm_selftest")
    return recv.raw


def keyfile_committer_key(path: str, curve, max_powers: int = 0) -> np.ndarray:
    """--ck file of the reference CLI -> powers_of_g as (count, 2*fq_limbs) Montgomery limbs (host only)."""
    L = lib()
    cid = curve_id(curve)
    words = 8 if cid == CURVE_BN254 else 12
    L.zkt_keyfile_committer_key.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64),
                                            ctypes.POINTER(ctypes.c_size_t)]
    n = ctypes.c_size_t(0)
    rc = L.zkt_keyfile_committer_key(path.encode(), cid, max_powers, None, ctypes.byref(n))
    if rc:
        raise ZktError(rc, "zkt_keyfile_committer_key(%s)" % path)
    out = np.zeros((n.value, words), dtype=np.uint64)
    rc = L.zkt_keyfile_committer_key(path.encode(), cid, max_powers, u64p(out) if out.size else None, ctypes.byref(n))
    if rc:
        raise ZktError(rc, "zkt_keyfile_committer_key(%s)" % path)
    return out


def keyfile_prover_key(path: str, curve):
    """--pk file -> the ten coefficient arrays ((len_k, 4) Montgomery limbs) in zkt_circuit_load order (host only)."""
    L = lib()
    cid = curve_id(curve)
    P64 = ctypes.POINTER(ctypes.c_uint64)
    L.zkt_keyfile_prover_key.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(P64), ctypes.POINTER(ctypes.c_size_t)]
    lens = (ctypes.c_size_t * 10)()
    rc = L.zkt_keyfile_prover_key(path.encode(), cid, None, lens)
    if rc:
        raise ZktError(rc, "zkt_keyfile_prover_key(%s)" % path)
    arrs = [np.zeros((max(lens[k], 1), 4), dtype=np.uint64) for k in range(10)]
    ptrs = (P64 * 10)(*[u64p(a) for a in arrs])
    rc = L.zkt_keyfile_prover_key(path.encode(), cid, ptrs, lens)
    if rc:
        raise ZktError(rc, "zkt_keyfile_prover_key(%s)" % path)
    return [a[:lens[k]] for k, a in enumerate(arrs)]


def keyfile_extended_prover_key(path: str, curve, which: int = -1):
    """--epk file -> the seventeen vector lengths, and vector `which` as (len, 4) Montgomery limbs (host only, streamed)."""
    L = lib()
    cid = curve_id(curve)
    P64 = ctypes.POINTER(ctypes.c_uint64)
    L.zkt_keyfile_extended_prover_key.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, P64, ctypes.c_size_t,
                                                  ctypes.POINTER(ctypes.c_size_t)]
    lens = (ctypes.c_size_t * 17)()
    rc = L.zkt_keyfile_extended_prover_key(path.encode(), cid, -1, None, 0, lens)
    if rc:
        raise ZktError(rc, "zkt_keyfile_extended_prover_key(%s)" % path)
    if which < 0:
        return list(lens), None
    out = np.zeros((max(lens[which], 1), 4), dtype=np.uint64)
    rc = L.zkt_keyfile_extended_prover_key(path.encode(), cid, which, u64p(out), lens[which], lens)
    if rc:
        raise ZktError(rc, "zkt_keyfile_extended_prover_key(%s, %d)" % (path, which))
    return list(lens), out[:lens[which]]


def keyfile_verifier_key(path: str, curve):
    """--vk file -> (n, pi_roots (k, 4), commitments (10, 2*fq_limbs), is_infinity (10,)); Montgomery limbs (host only)."""
    L = lib()
    cid = curve_id(curve)
    words = 8 if cid == CURVE_BN254 else 12
    P64 = ctypes.POINTER(ctypes.c_uint64)
    L.zkt_keyfile_verifier_key.argtypes = [ctypes.c_char_p, ctypes.c_int, P64, P64, ctypes.c_size_t,
                                           ctypes.POINTER(ctypes.c_size_t), P64, ctypes.POINTER(ctypes.c_int)]
    n, k = ctypes.c_uint64(0), ctypes.c_size_t(0)
    rc = L.zkt_keyfile_verifier_key(path.encode(), cid, ctypes.byref(n), None, 0, ctypes.byref(k), None, None)
    if rc:
        raise ZktError(rc, "zkt_keyfile_verifier_key(%s)" % path)
    roots = np.zeros((max(k.value, 1), 4), dtype=np.uint64)
    commits = np.zeros((10, words), dtype=np.uint64)
    inf = (ctypes.c_int * 10)()
    rc = L.zkt_keyfile_verifier_key(path.encode(), cid, ctypes.byref(n), u64p(roots), roots.shape[0], ctypes.byref(k),
                                    u64p(commits), inf)
    if rc:
        raise ZktError(rc, "zkt_keyfile_verifier_key(%s)" % path)
    return n.value, roots[:k.value], commits, np.array([bool(x) for x in inf])


class VerifyInputs(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint64), ("vk_commitments", ctypes.POINTER(ctypes.c_uint64)),
                ("vk_is_infinity", ctypes.POINTER(ctypes.c_int)), ("pi_roots", ctypes.POINTER(ctypes.c_uint64)),
                ("pub_inputs", ctypes.POINTER(ctypes.c_uint64)), ("n_pi", ctypes.c_size_t),
                ("proof", ctypes.c_char_p), ("proof_len", ctypes.c_size_t), ("g", ctypes.POINTER(ctypes.c_uint64))]


def verify_prepare(curve, n: int, vk_commits: np.ndarray, vk_inf, pi_roots: np.ndarray, pub_inputs: np.ndarray, proof: bytes,
                   g_xy: np.ndarray, transcript):
    """proof_system/proof.rs:285-503 without the pairings (zkt_verify_prepare; host only).  Arrays are Montgomery limbs;
    `transcript` is a seeded Transcript.  -> (pairs (4, 2*fq_limbs) = L1, W1, L2, W2; is_infinity (4,))."""
    L = lib()
    cid = curve_id(curve)
    words = 8 if cid == CURVE_BN254 else 12
    vk_commits = np.ascontiguousarray(vk_commits, dtype=np.uint64).reshape(10, words)
    pi_roots = np.ascontiguousarray(pi_roots, dtype=np.uint64).reshape(-1, 4)
    pub_inputs = np.ascontiguousarray(pub_inputs, dtype=np.uint64).reshape(-1, 4)
    assert pi_roots.shape == pub_inputs.shape
    g_xy = np.ascontiguousarray(g_xy, dtype=np.uint64).reshape(words)
    inf = (ctypes.c_int * 10)(*[int(bool(x)) for x in vk_inf])
    null = ctypes.POINTER(ctypes.c_uint64)()
    inp = VerifyInputs(n, u64p(vk_commits), inf, u64p(pi_roots) if pi_roots.size else null,
                       u64p(pub_inputs) if pub_inputs.size else null, pi_roots.shape[0], proof, len(proof), u64p(g_xy))
    out = np.zeros((4, words), dtype=np.uint64)
    oinf = (ctypes.c_int * 4)()
    L.zkt_verify_prepare.argtypes = [ctypes.c_int, ctypes.POINTER(VerifyInputs), ctypes.c_void_p,
                                     ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]
    rc = L.zkt_verify_prepare(cid, ctypes.byref(inp), transcript.handle, u64p(out), oinf)
    if rc:
        raise ZktError(rc, "zkt_verify_prepare")
    return out, np.array([bool(x) for x in oinf])


def pairing_product_is_one(curve, g1_points: np.ndarray, g2_points: np.ndarray) -> bool:
    """prod_i e(P_i, Q_i) == 1 on the host (zkt_pairing_product_is_one).  g1: (n, 2*fq_limbs); g2: (n, 4*fq_limbs) =
    x.c0, x.c1, y.c0, y.c1; Montgomery limbs."""
    L = lib()
    cid = curve_id(curve)
    words = 4 if cid == CURVE_BN254 else 6
    g1 = np.ascontiguousarray(g1_points, dtype=np.uint64).reshape(-1, 2 * words)
    g2 = np.ascontiguousarray(g2_points, dtype=np.uint64).reshape(-1, 4 * words)
    assert g1.shape[0] == g2.shape[0]
    P64 = ctypes.POINTER(ctypes.c_uint64)
    L.zkt_pairing_product_is_one.argtypes = [ctypes.c_int, P64, P64, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
    one = ctypes.c_int(0)
    rc = L.zkt_pairing_product_is_one(cid, u64p(g1) if g1.size else None, u64p(g2) if g2.size else None, g1.shape[0],
                                      ctypes.byref(one))
    if rc:
        raise ZktError(rc, "zkt_pairing_product_is_one")
    return bool(one.value)


def verify(curve, n: int, vk_commits, vk_inf, pi_roots, pub_inputs, proof: bytes, g_xy, h_g2, beta_h_g2, transcript) -> bool:
    """Proof::verify (proof_system/proof.rs:285-503) on the host, pairings included (zkt_verify)."""
    L = lib()
    cid = curve_id(curve)
    words = 8 if cid == CURVE_BN254 else 12
    vk_commits = np.ascontiguousarray(vk_commits, dtype=np.uint64).reshape(10, words)
    pi_roots = np.ascontiguousarray(pi_roots, dtype=np.uint64).reshape(-1, 4)
    pub_inputs = np.ascontiguousarray(pub_inputs, dtype=np.uint64).reshape(-1, 4)
    g_xy = np.ascontiguousarray(g_xy, dtype=np.uint64).reshape(words)
    h = np.ascontiguousarray(h_g2, dtype=np.uint64).reshape(2 * words)
    bh = np.ascontiguousarray(beta_h_g2, dtype=np.uint64).reshape(2 * words)
    inf = (ctypes.c_int * 10)(*[int(bool(x)) for x in vk_inf])
    null = ctypes.POINTER(ctypes.c_uint64)()
    inp = VerifyInputs(n, u64p(vk_commits), inf, u64p(pi_roots) if pi_roots.size else null,
                       u64p(pub_inputs) if pub_inputs.size else null, pi_roots.shape[0], proof, len(proof), u64p(g_xy))
    P64 = ctypes.POINTER(ctypes.c_uint64)
    L.zkt_verify.argtypes = [ctypes.c_int, ctypes.POINTER(VerifyInputs), ctypes.c_void_p, P64, P64, ctypes.POINTER(ctypes.c_int)]
    ok = ctypes.c_int(0)
    rc = L.zkt_verify(cid, ctypes.byref(inp), transcript.handle, u64p(h), u64p(bh), ctypes.byref(ok))
    if rc:
        raise ZktError(rc, "zkt_verify")
    return bool(ok.value)


def verify_batch(curve, items, h_g2, beta_h_g2) -> bool:
    """zkt_verify_batch: `items` = [(n, vk_commits, vk_inf, pi_roots, pub_inputs, proof bytes, g_xy, seeded Transcript), ...];
    True iff every proof verifies (one product of two pairings for the whole batch)."""
    L = lib()
    cid = curve_id(curve)
    words = 8 if cid == CURVE_BN254 else 12
    h = np.ascontiguousarray(h_g2, dtype=np.uint64).reshape(2 * words)
    bh = np.ascontiguousarray(beta_h_g2, dtype=np.uint64).reshape(2 * words)
    null = ctypes.POINTER(ctypes.c_uint64)()
    k = len(items)
    ins = (VerifyInputs * max(k, 1))()
    trs = (ctypes.c_void_p * max(k, 1))()
    keep = []
    for i, (n, vk_commits, vk_inf, pi_roots, pub_inputs, proof, g_xy, transcript) in enumerate(items):
        vk_commits = np.ascontiguousarray(vk_commits, dtype=np.uint64).reshape(10, words)
        pi_roots = np.ascontiguousarray(pi_roots, dtype=np.uint64).reshape(-1, 4)
        pub_inputs = np.ascontiguousarray(pub_inputs, dtype=np.uint64).reshape(-1, 4)
        assert pi_roots.shape == pub_inputs.shape
        g_xy = np.ascontiguousarray(g_xy, dtype=np.uint64).reshape(words)
        inf = (ctypes.c_int * 10)(*[int(bool(x)) for x in vk_inf])
        keep.append((vk_commits, pi_roots, pub_inputs, g_xy, inf, proof))
        ins[i] = VerifyInputs(n, u64p(vk_commits), inf, u64p(pi_roots) if pi_roots.size else null,
                              u64p(pub_inputs) if pub_inputs.size else null, pi_roots.shape[0], proof, len(proof), u64p(g_xy))
        trs[i] = transcript.handle
    P64 = ctypes.POINTER(ctypes.c_uint64)
    L.zkt_verify_batch.argtypes = [ctypes.c_int, ctypes.POINTER(VerifyInputs), ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t,
                                   P64, P64, ctypes.POINTER(ctypes.c_int)]
    ok = ctypes.c_int(0)
    rc = L.zkt_verify_batch(cid, ins, trs, k, u64p(h), u64p(bh), ctypes.byref(ok))
    if rc:
        raise ZktError(rc, "zkt_verify_batch")
    return bool(ok.value)


class ProveInputs(ctypes.Structure):
    _fields_ = [("a_evals", ctypes.POINTER(ctypes.c_uint64)), ("b_evals", ctypes.POINTER(ctypes.c_uint64)),
                ("c_evals", ctypes.POINTER(ctypes.c_uint64)), ("n_rows", ctypes.c_size_t),
                ("table", ctypes.POINTER(ctypes.c_uint64)), ("table_len", ctypes.c_size_t),
                ("pi_pos", ctypes.POINTER(ctypes.c_size_t)), ("pi_vals", ctypes.POINTER(ctypes.c_uint64)),
                ("n_pi", ctypes.c_size_t), ("blinders", ctypes.POINTER(ctypes.c_uint64)),
                ("wires_on_device", ctypes.c_int),
                ("variables", ctypes.POINTER(ctypes.c_uint64)), ("n_vars", ctypes.c_size_t),
                ("w_l", ctypes.POINTER(ctypes.c_uint32)), ("w_r", ctypes.POINTER(ctypes.c_uint32)),
                ("w_o", ctypes.POINTER(ctypes.c_uint32))]


class PreparedInputs:
    """A zkt_prove_inputs struct together with the arrays it points into."""

    def __init__(self, struct, keep):
        self.struct = struct
        self._keep = keep


def _bind_prover(L):
    vp, u64p_, u8p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint8)
    L.zkt_transcript_new.restype = vp
    L.zkt_transcript_new.argtypes = [ctypes.c_int, ctypes.c_char_p]
    L.zkt_transcript_free.argtypes = [vp]
    L.zkt_transcript_free.restype = None
    L.zkt_transcript_append_u64.argtypes = [vp, ctypes.c_char_p, ctypes.c_uint64]
    L.zkt_transcript_append_u64.restype = None
    L.zkt_transcript_append_scalars.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    L.zkt_transcript_append_scalars.restype = None
    L.zkt_transcript_append_commitment.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                                   ctypes.c_size_t, ctypes.c_int]
    L.zkt_transcript_append_commitment.restype = None
    L.zkt_transcript_challenge_scalar.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, u8p]
    L.zkt_transcript_challenge_scalar.restype = None
    L.zkt_transcript_seed.argtypes = [vp, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    L.zkt_transcript_seed.restype = None
    L.zkt_transcript_append_message.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    L.zkt_transcript_challenge_bytes.argtypes = [vp, ctypes.c_char_p, u8p, ctypes.c_size_t]
    L.zkt_circuit_load.argtypes = [vp, ctypes.c_int, ctypes.POINTER(u64p_), ctypes.POINTER(ctypes.c_size_t)]
    L.zkt_circuit_setup.argtypes = [vp, ctypes.c_int, ctypes.POINTER(u64p_), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int,
                                    u64p_, ctypes.POINTER(ctypes.c_int)]
    L.zkt_prove.argtypes = [vp, ctypes.POINTER(ProveInputs), vp, u8p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.zkt_prove_set_next.argtypes = [vp, ctypes.POINTER(ProveInputs)]


def g1_sum_host(curve, points) -> tuple:
    """Host sum of affine G1 points ((count, 2*fq_limbs) Montgomery limbs, (0,0) = identity) -> (xy limbs, is_infinity).
    The combine step of an index-range-sharded MSM; needs no GPU context."""
    L = lib()
    cid = CURVE_BN254 if curve in ("bn254", CURVE_BN254) else CURVE_BLS12_381
    limbs = 4 if cid == CURVE_BN254 else 6
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 2 * limbs)
    out = np.zeros(2 * limbs, dtype=np.uint64)
    inf = ctypes.c_int(0)
    L.zkt_g1_sum_host.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t,
                                  ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]
    rc = L.zkt_g1_sum_host(cid, u64p(pts) if pts.size else ctypes.POINTER(ctypes.c_uint64)(), pts.shape[0], u64p(out),
                           ctypes.byref(inf))
    if rc:
        raise ZktError(rc, "zkt_g1_sum_host")
    return out, bool(inf.value)


def srs_generate_g2(curve, tau: int):
    """The G2 half of the test SRS (zkt_srs_generate_g2): -> (h, beta_h = tau h), each (4 * fq_limbs,) Montgomery limbs in
    arkworks' Fp2 layout -- SonicKZG10 VerifierKey::h / ::beta_h for zkt_verify."""
    L = lib()
    cid = curve_id(curve)
    words = 8 if cid == CURVE_BN254 else 12
    t = np.array([(int(tau) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    h = np.zeros(2 * words, dtype=np.uint64)
    bh = np.zeros(2 * words, dtype=np.uint64)
    P64 = ctypes.POINTER(ctypes.c_uint64)
    L.zkt_srs_generate_g2.argtypes = [ctypes.c_int, P64, P64, P64]
    rc = L.zkt_srs_generate_g2(cid, u64p(t), u64p(h), u64p(bh))
    if rc:
        raise ZktError(rc, "zkt_srs_generate_g2")
    return h, bh


def g1_msm_host(curve, points, scalars, montgomery: bool = True) -> tuple:
    """HomomorphicCommitment::multi_scalar_mul (commitment.rs:32-45) on arbitrary points, on the host ->
    (xy limbs, is_infinity)."""
    L = lib()
    cid = curve_id(curve)
    words = 8 if cid == CURVE_BN254 else 12
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, words)
    sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    assert pts.shape[0] == sc.shape[0]
    out = np.zeros(words, dtype=np.uint64)
    inf = ctypes.c_int(0)
    P64 = ctypes.POINTER(ctypes.c_uint64)
    L.zkt_g1_msm_host.argtypes = [ctypes.c_int, P64, P64, ctypes.c_size_t, ctypes.c_int, P64, ctypes.POINTER(ctypes.c_int)]
    rc = L.zkt_g1_msm_host(cid, u64p(pts) if pts.size else None, u64p(sc) if sc.size else None, pts.shape[0], int(montgomery),
                           u64p(out), ctypes.byref(inf))
    if rc:
        raise ZktError(rc, "zkt_g1_msm_host")
    return out, bool(inf.value)


class Transcript:
    """Built-in host transcript (T: TranscriptProtocol): kind 'merlin' (plonk-core/src/transcript.rs:46-109)
    or 'ethereum' (gadgets/src/transcript.rs:8-90).  Scalars / coordinates are canonical integers."""

    def __init__(self, kind: str = "merlin", label: str = "ZKT Plonk", fr_bits: int = 254, fq_bytes: int = 32):
        self._L = lib()
        self._h = ctypes.c_void_p(self._L.zkt_transcript_new(0 if kind == "merlin" else 1, label.encode()))
        self.fr_bits = fr_bits
        self.fq_bytes = fq_bytes

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.zkt_transcript_free(self._h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def append_u64(self, label: str, v: int):
        self._L.zkt_transcript_append_u64(self._h, label.encode(), v)

    def append_scalar(self, label: str, v: int):
        self._L.zkt_transcript_append_scalars(self._h, label.encode(), int(v).to_bytes(32, "little"), 1, 1)

    def append_scalars(self, label: str, vals):
        vals = list(vals)
        self._L.zkt_transcript_append_scalars(self._h, label.encode(),
                                              b"".join(int(v).to_bytes(32, "little") for v in vals), len(vals), 0)

    def append_commitment(self, label: str, point):
        nb = self.fq_bytes
        if point is None:
            self._L.zkt_transcript_append_commitment(self._h, label.encode(), bytes(nb), bytes(nb), nb, 1)
        else:
            self._L.zkt_transcript_append_commitment(self._h, label.encode(), int(point[0]).to_bytes(nb, "little"),
                                                     int(point[1]).to_bytes(nb, "little"), nb, 0)

    def seed(self, circuit_size: int, points):
        """VerifierKey::seed_transcript (keys/mod.rs:260-275) in one call; points: the ten commitments in ProverKey
        order, affine canonical ints or None (identity)."""
        nb = self.fq_bytes
        buf = b"".join(bytes(2 * nb) if p is None else int(p[0]).to_bytes(nb, "little") + int(p[1]).to_bytes(nb, "little")
                       for p in points)
        inf = bytes(1 if p is None else 0 for p in points)
        self._L.zkt_transcript_seed(self._h, circuit_size, buf, inf, nb)

    def challenge_scalar(self, label: str) -> int:
        out = (ctypes.c_uint8 * 32)()
        self._L.zkt_transcript_challenge_scalar(self._h, label.encode(), self.fr_bits, out)
        return int.from_bytes(bytes(out), "little")

    def append_message(self, label: bytes, msg: bytes):
        assert self._L.zkt_transcript_append_message(self._h, label, msg, len(msg)) == 0

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        out = (ctypes.c_uint8 * n)()
        assert self._L.zkt_transcript_challenge_bytes(self._h, label, out, n) == 0
        return bytes(out)


def u64p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], "need a C-contiguous uint64 array"
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


class Context:
    """zkt_ctx wrapper: one per caller thread / proof stream."""

    def __init__(self, curve="bn254", device: int = 0):
        self._L = lib()
        self.curve = curve_id(curve)
        self.fq_limbs = 4 if self.curve == CURVE_BN254 else 6
        h = ctypes.c_void_p()
        rc = self._L.zkt_ctx_create(self.curve, device, ctypes.byref(h))
        if rc:
            raise ZktError(rc, "zkt_ctx_create failed (no GPU? there is no CPU fallback)")
        self._h = h

    def fork(self) -> "Context":
        """zkt_ctx_fork: a context sharing this one's key / circuit / twiddle tables, with its own stream and work buffers."""
        self._L.zkt_ctx_fork.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        h = ctypes.c_void_p()
        self.check(self._L.zkt_ctx_fork(self._h, ctypes.byref(h)))
        other = Context.__new__(Context)
        other.__dict__.update(self.__dict__)
        other._h = h
        return other

    def close(self):
        if getattr(self, "_h", None):
            self._L.zkt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        if rc:
            raise ZktError(rc, self._L.zkt_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    def set_stream(self, stream_ptr: int):
        self.check(self._L.zkt_ctx_set_stream(self._h, ctypes.c_void_p(stream_ptr)))

    def synchronize(self):
        self.check(self._L.zkt_ctx_synchronize(self._h))

    def profile_enable(self, on: bool = True):
        self.check(self._L.zkt_profile_enable(self._h, int(on)))

    def profile_get(self, name: str):
        """-> (launch count, total milliseconds) measured with HIP events on the context's stream."""
        calls, ms = ctypes.c_uint64(0), ctypes.c_double(0.0)
        self.check(self._L.zkt_profile_get(self._h, name.encode(), ctypes.byref(calls), ctypes.byref(ms)))
        return calls.value, ms.value

    # -- one proof across several GPUs --------------------------------------------------------------
    def set_comm(self, comm):
        """Attach a communicator (an object with a `.vt` CommVtable, e.g. parallel.TorchComm) or detach with None.
        Drops whatever SRS / circuit was loaded: keys are laid out for the rank's share."""
        L = self._L
        L.zkt_ctx_set_comm.argtypes = [ctypes.c_void_p, ctypes.POINTER(CommVtable)]
        self._comm = comm                       # keeps the callback alive
        self.check(L.zkt_ctx_set_comm(self._h, ctypes.byref(comm.vt) if comm is not None else None))

    def comm_stats(self):
        """-> (collective calls, bytes sent by this rank) since the communicator was attached."""
        L = self._L
        L.zkt_comm_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self.check(L.zkt_comm_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def srs_load_slice(self, pts_slice: np.ndarray, offset: int, total: int):
        pts = np.ascontiguousarray(pts_slice, dtype=np.uint64).reshape(-1, 2 * self.fq_limbs)
        L = self._L
        L.zkt_srs_load_slice.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t, ctypes.c_size_t,
                                         ctypes.c_size_t]
        self.check(L.zkt_srs_load_slice(self._h, u64p(pts), offset, pts.shape[0], total))

    def srs_generate_slice(self, tau: int, offset: int, count: int, total: int):
        t = np.array([(tau >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
        L = self._L
        L.zkt_srs_generate_slice.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t,
                                             ctypes.c_size_t, ctypes.c_size_t]
        self.check(L.zkt_srs_generate_slice(self._h, u64p(t), offset, count, total))

    def ntt_class(self, log_n: int, log_big: int, cls: int, arr: np.ndarray) -> np.ndarray:
        """One GPU's share (output indices = cls mod 2^(log_big - log_n)) of the forward coset transform of size
        2^log_big; `arr` may be longer than 2^log_n (folded)."""
        arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
        out = np.empty((1 << log_n, 4), dtype=np.uint64)
        L = self._L
        L.zkt_ntt_class.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64),
                                    ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
        self.check(L.zkt_ntt_class(self._h, log_n, log_big, cls, u64p(arr), arr.shape[0], u64p(out)))
        return out

    # -- batched Poseidon (witness synthesis) ------------------------------------------------------------
    def poseidon_hash_batch(self, width, half_full, partial, round_constants, mds, domain_tag, inputs, trace=False):
        """plonk-hashing's Poseidon permutation for a batch of inputs ((batch, arity, 4) Montgomery limbs) ->
        hashes (batch, 4) [, states (batch, rounds + 1, width, 4)]."""
        class Params(ctypes.Structure):
            _fields_ = [("width", ctypes.c_int), ("half_full_rounds", ctypes.c_int), ("partial_rounds", ctypes.c_int),
                        ("round_constants", ctypes.POINTER(ctypes.c_uint64)), ("mds", ctypes.POINTER(ctypes.c_uint64)),
                        ("domain_tag", ctypes.POINTER(ctypes.c_uint64))]
        rc = np.ascontiguousarray(round_constants, dtype=np.uint64).reshape(-1, 4)
        m = np.ascontiguousarray(mds, dtype=np.uint64).reshape(width * width, 4)
        tag = np.ascontiguousarray(domain_tag, dtype=np.uint64).reshape(4)
        inp = np.ascontiguousarray(inputs, dtype=np.uint64)
        batch, arity = inp.shape[0], (inp.shape[1] if inp.ndim == 3 else 0)
        rounds = 2 * half_full + partial
        assert rc.shape[0] == rounds * width
        out = np.zeros((batch, 4), dtype=np.uint64)
        st = np.zeros((batch, rounds + 1, width, 4), dtype=np.uint64) if trace else None
        prm = Params(width, half_full, partial, u64p(rc), u64p(m), u64p(tag))
        L = self._L
        L.zkt_poseidon_hash_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(Params), ctypes.POINTER(ctypes.c_uint64),
                                              ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64),
                                              ctypes.POINTER(ctypes.c_uint64)]
        self.check(L.zkt_poseidon_hash_batch(self._h, ctypes.byref(prm), u64p(inp.reshape(-1, 4)) if inp.size else None, batch,
                                             arity, u64p(out), u64p(st.reshape(-1, 4)) if trace else None))
        return (out, st) if trace else out

    def poseidon_load(self, width, half_full, partial, round_constants, mds, domain_tag) -> int:
        """zkt_poseidon_load: PoseidonConstants resident in HBM -> opaque handle (free with poseidon_free)."""
        class Params(ctypes.Structure):
            _fields_ = [("width", ctypes.c_int), ("half_full_rounds", ctypes.c_int), ("partial_rounds", ctypes.c_int),
                        ("round_constants", ctypes.POINTER(ctypes.c_uint64)), ("mds", ctypes.POINTER(ctypes.c_uint64)),
                        ("domain_tag", ctypes.POINTER(ctypes.c_uint64))]
        rc = np.ascontiguousarray(round_constants, dtype=np.uint64).reshape(-1, 4)
        m = np.ascontiguousarray(mds, dtype=np.uint64).reshape(width * width, 4)
        tag = np.ascontiguousarray(domain_tag, dtype=np.uint64).reshape(4)
        assert rc.shape[0] == (2 * half_full + partial) * width
        prm = Params(width, half_full, partial, u64p(rc), u64p(m), u64p(tag))
        L = self._L
        L.zkt_poseidon_load.argtypes = [ctypes.c_void_p, ctypes.POINTER(Params), ctypes.POINTER(ctypes.c_void_p)]
        h = ctypes.c_void_p()
        self.check(L.zkt_poseidon_load(self._h, ctypes.byref(prm), ctypes.byref(h)))
        return h.value

    def poseidon_free(self, handle: int):
        L = self._L
        L.zkt_poseidon_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.zkt_poseidon_free.restype = None
        L.zkt_poseidon_free(self._h, ctypes.c_void_p(handle))

    def poseidon_hash_batch_dev(self, handle: int, d_inputs: int, batch: int, arity: int, d_hashes: int, d_states: int = 0):
        """zkt_poseidon_hash_batch_dev: device pointers in and out, enqueue only (no allocation, no synchronisation)."""
        L = self._L
        vp = ctypes.c_void_p
        L.zkt_poseidon_hash_batch_dev.argtypes = [vp, vp, vp, ctypes.c_size_t, ctypes.c_int, vp, vp]
        self.check(L.zkt_poseidon_hash_batch_dev(self._h, vp(handle), vp(d_inputs), batch, arity, vp(d_hashes),
                                                 vp(d_states) if d_states else None))

    def poseidon_gadget_vars_per_hash(self, handle: int) -> int:
        L = self._L
        L.zkt_poseidon_gadget_vars_per_hash.argtypes = [ctypes.c_void_p]
        L.zkt_poseidon_gadget_vars_per_hash.restype = ctypes.c_size_t
        return int(L.zkt_poseidon_gadget_vars_per_hash(ctypes.c_void_p(handle)))

    def poseidon_gadget_witness_dev(self, handle: int, batch: int, arity: int, d_variables: int, n_vars: int, d_inputs: int = 0,
                                    d_input_vars: int = 0, d_trace_base: int = 0, trace_base0: int = 0, d_out_hashes: int = 0,
                                    kernel: int = 0, validate_only: bool = False):
        """zkt_poseidon_gadget_witness_dev: the variables PlonkSpecRef's gadget allocates for `batch` independent hashes,
        written in allocation order into the variable map at d_variables (device pointers; enqueue only).
        validate_only: zkt_poseidon_gadget_validate on the same arguments instead (disjoint traces, no input made by the
        same launch; synchronises, launches nothing)."""
        class Args(ctypes.Structure):
            _fields_ = [("batch", ctypes.c_size_t), ("arity", ctypes.c_int), ("d_inputs", ctypes.c_void_p),
                        ("d_input_vars", ctypes.c_void_p), ("d_variables", ctypes.c_void_p), ("n_vars", ctypes.c_size_t),
                        ("d_trace_base", ctypes.c_void_p), ("trace_base0", ctypes.c_size_t), ("d_out_hashes", ctypes.c_void_p),
                        ("kernel", ctypes.c_int)]
        a = Args(batch, arity, d_inputs or None, d_input_vars or None, d_variables or None, n_vars, d_trace_base or None,
                 trace_base0, d_out_hashes or None, kernel)
        L = self._L
        fn = L.zkt_poseidon_gadget_validate if validate_only else L.zkt_poseidon_gadget_witness_dev
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(Args)]
        self.check(fn(self._h, ctypes.c_void_p(handle), ctypes.byref(a)))

    def poseidon_gadget_check(self, handle: int):
        """zkt_poseidon_gadget_check: synchronises; raises if a launch skipped a hash (index outside the variable map)."""
        L = self._L
        L.zkt_poseidon_gadget_check.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.check(L.zkt_poseidon_gadget_check(self._h, ctypes.c_void_p(handle)))

    # -- device memory ------------------------------------------------------------------------
    def alloc(self, nbytes: int) -> int:
        p = ctypes.c_void_p()
        self.check(self._L.zkt_dev_alloc(self._h, nbytes, ctypes.byref(p)))
        return p.value

    def free(self, dptr: int):
        self.check(self._L.zkt_dev_free(self._h, ctypes.c_void_p(dptr)))

    def upload(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self.check(self._L.zkt_dev_upload(self._h, ctypes.c_void_p(dptr), arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))

    def download(self, dptr: int, shape, dtype=np.uint64) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self.check(self._L.zkt_dev_download(self._h, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(dptr), out.nbytes))
        return out

    # -- Domain seam ------------------------------------------------------------------------------
    def ntt(self, log_n: int, arr: np.ndarray, inverse=False, coset=False) -> np.ndarray:
        arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
        out = np.empty((1 << max(log_n, 0), 4), dtype=np.uint64)
        self.check(self._L.zkt_ntt(self._h, log_n, int(inverse), int(coset), u64p(arr), arr.shape[0], u64p(out)))
        return out

    def ntt_dev(self, log_n: int, d_in: int, in_len: int, d_out: int, inverse=False, coset=False):
        self.check(self._L.zkt_ntt_dev(self._h, log_n, int(inverse), int(coset), ctypes.c_void_p(d_in), in_len,
                                       ctypes.c_void_p(d_out)))

    def group_gen(self, log_n: int) -> np.ndarray:
        out = np.zeros(4, dtype=np.uint64)
        self.check(self._L.zkt_domain_group_gen(self._h, log_n, u64p(out)))
        return out

    # -- Commitment seam ----------------------------------------------------------------------------
    def srs_load(self, pts: np.ndarray):
        pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, 2 * self.fq_limbs)
        self.check(self._L.zkt_srs_load(self._h, u64p(pts), pts.shape[0]))

    def srs_load_file(self, ck_path: str, max_powers: int = 0):
        """zkt_srs_load from the reference CLI's --ck file (CommitterKey.powers_of_g)."""
        L = self._L
        L.zkt_srs_load_file.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        self.check(L.zkt_srs_load_file(self._h, ck_path.encode(), max_powers))

    def circuit_load_file(self, pk_path: str, log_n: int):
        """zkt_circuit_load from the reference CLI's --pk file (ProverKey<F>)."""
        L = self._L
        L.zkt_circuit_load_file.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        self.check(L.zkt_circuit_load_file(self._h, pk_path.encode(), log_n))

    def srs_generate(self, tau: int, count: int):
        t = np.array([(tau >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
        self.check(self._L.zkt_srs_generate(self._h, u64p(t), count))

    def srs_download(self, offset: int, count: int) -> np.ndarray:
        out = np.empty((count, 2 * self.fq_limbs), dtype=np.uint64)
        self.check(self._L.zkt_srs_download(self._h, offset, count, u64p(out)))
        return out

    def msm(self, scalars: np.ndarray, base_offset: int = 0, montgomery: bool = True):
        """-> (xy Montgomery limbs (2*fq_limbs,), is_infinity)"""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(2 * self.fq_limbs, dtype=np.uint64)
        inf = ctypes.c_int(0)
        self.check(self._L.zkt_msm_g1(self._h, u64p(scalars), scalars.shape[0], base_offset, int(montgomery),
                                      u64p(out), ctypes.byref(inf)))
        return out, bool(inf.value)

    def msm_dev(self, d_scalars: int, n: int, base_offset: int = 0, montgomery: bool = True) -> np.ndarray:
        out = np.zeros(2 * self.fq_limbs, dtype=np.uint64)
        self.check(self._L.zkt_msm_g1_dev(self._h, ctypes.c_void_p(d_scalars), n, base_offset, int(montgomery),
                                          out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def msm_enqueue_dev(self, d_scalars: int, n: int, base_offset: int = 0, montgomery: bool = True):
        self.check(self._L.zkt_msm_enqueue_dev(self._h, ctypes.c_void_p(d_scalars), n, base_offset, int(montgomery)))

    def msm_info(self):
        c, w, n = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_size_t(0)
        self.check(self._L.zkt_msm_info(self._h, ctypes.byref(c), ctypes.byref(w), ctypes.byref(n)))
        return dict(window_bits=c.value, windows=w.value, srs_count=n.value)

    # -- commitments of evaluation vectors (Lagrange-basis key, include/zkt_plonk.h) ----------------------
    def set_lagrange(self, on: bool = True):
        self._L.zkt_ctx_set_lagrange.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.check(self._L.zkt_ctx_set_lagrange(self._h, int(on)))

    def lagrange_info(self):
        self._L.zkt_lagrange_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_size_t)]
        lg, nb = ctypes.c_int(0), ctypes.c_size_t(0)
        self.check(self._L.zkt_lagrange_info(self._h, ctypes.byref(lg), ctypes.byref(nb)))
        return dict(log_n=lg.value, bases=nb.value)

    def commit_evals_dev(self, d_evals: int, blinders=None, path: int = 1):
        """PC::commit(poly_from_evals(evals) + blinders): path 0 through the coefficients, 1 through the Lagrange basis.
        -> (xy Montgomery limbs, is_infinity); the domain is the loaded circuit's."""
        self._L.zkt_commit_evals_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int,
                                                 ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]
        k = 0 if blinders is None else int(np.asarray(blinders).reshape(-1, 4).shape[0])
        bl = np.ascontiguousarray(blinders, dtype=np.uint64).reshape(-1, 4) if k else None
        out = np.zeros(2 * self.fq_limbs, dtype=np.uint64)
        inf = ctypes.c_int(0)
        self.check(self._L.zkt_commit_evals_dev(self._h, ctypes.c_void_p(d_evals), u64p(bl) if k else None, k, int(path),
                                                u64p(out), ctypes.byref(inf)))
        return out, bool(inf.value)

    # -- prover -----------------------------------------------------------------------------------------
    def circuit_load(self, log_n: int, pk_polys):
        """pk_polys: 10 arrays (len_k, 4) in the order q_m q_l q_r q_o q_c sigma1 sigma2 sigma3 q_lookup q_table."""
        arrs = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) for p in pk_polys]
        assert len(arrs) == 10
        ptrs = (ctypes.POINTER(ctypes.c_uint64) * 10)(*[u64p(a) if a.size else ctypes.POINTER(ctypes.c_uint64)() for a in arrs])
        lens = (ctypes.c_size_t * 10)(*[a.shape[0] for a in arrs])
        self.check(self._L.zkt_circuit_load(self._h, log_n, ptrs, lens))

    def circuit_setup(self, log_n: int, evals):
        """proof_system::setup (setup.rs:42-166) on the device.  evals: the 10 evaluation vectors (len_k <= n, 4) in
        ProverKey order, Montgomery limbs.  Leaves the circuit loaded; -> (commitments (10, 2*fq_limbs) Montgomery
        limbs, is_infinity (10,) bool) = the VerifierKey commitments in the same order."""
        arrs = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) for p in evals]
        assert len(arrs) == 10
        ptrs = (ctypes.POINTER(ctypes.c_uint64) * 10)(*[u64p(a) if a.size else ctypes.POINTER(ctypes.c_uint64)() for a in arrs])
        lens = (ctypes.c_size_t * 10)(*[a.shape[0] for a in arrs])
        out = np.zeros((10, 2 * self.fq_limbs), dtype=np.uint64)
        inf = (ctypes.c_int * 10)()
        self.check(self._L.zkt_circuit_setup(self._h, log_n, ptrs, lens, 0, u64p(out), inf))
        return out, np.array([bool(x) for x in inf])

    def _prepare(self, wires, n_rows, table, pi_pos, pi_vals, blinders, on_device, variables=None, idx=None, keep=()):
        table = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, 4)
        pi_vals = np.ascontiguousarray(pi_vals, dtype=np.uint64).reshape(-1, 4)
        blinders = np.ascontiguousarray(blinders, dtype=np.uint64).reshape(19, 4)
        pos = (ctypes.c_size_t * max(1, len(pi_pos)))(*pi_pos)
        null = ctypes.POINTER(ctypes.c_uint64)()
        null32 = ctypes.POINTER(ctypes.c_uint32)()
        inp = ProveInputs(wires[0], wires[1], wires[2], n_rows, u64p(table) if table.size else null, table.shape[0],
                          pos, u64p(pi_vals) if pi_vals.size else null, len(pi_pos), u64p(blinders), int(on_device),
                          variables[0] if variables else null, variables[1] if variables else 0,
                          idx[0] if idx else null32, idx[1] if idx else null32, idx[2] if idx else null32)
        return PreparedInputs(inp, (table, pi_vals, blinders, pos) + tuple(keep))

    def prepare_dev(self, d_a: int, d_b: int, d_c: int, n_rows: int, table, pi_pos, pi_vals, blinders) -> "PreparedInputs":
        """zkt_prove_inputs for wire vectors already resident in HBM (device pointers), built once and reusable: the
        prefetch of zkt_prove_set_next recognises the next proof by the identity of these pointers."""
        cast = lambda p: ctypes.cast(ctypes.c_void_p(p), ctypes.POINTER(ctypes.c_uint64))
        return self._prepare((cast(d_a), cast(d_b), cast(d_c)), n_rows, table, pi_pos, pi_vals, blinders, True)

    def prepare_host(self, a, b, c, table, pi_pos, pi_vals, blinders) -> "PreparedInputs":
        """zkt_prove_inputs for wire vectors in host memory ((n_rows, 4) Montgomery uint64 arrays, kept alive here)."""
        a, b, c = (np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in (a, b, c))
        null = ctypes.POINTER(ctypes.c_uint64)()
        return self._prepare(tuple(u64p(x) if x.size else null for x in (a, b, c)), a.shape[0], table, pi_pos, pi_vals,
                             blinders, False, keep=(a, b, c))

    def prepare_vars(self, variables, w_l, w_r, w_o, table, pi_pos, pi_vals, blinders) -> "PreparedInputs":
        """zkt_prove_inputs in the composer's own witness layout (host memory): `variables` (n_vars, 4) Montgomery
        values and three uint32 index vectors (0xFFFFFFFF = Variable::Zero); prove.rs:49-55 runs on the device."""
        variables = np.ascontiguousarray(variables, dtype=np.uint64).reshape(-1, 4)
        idx = [np.ascontiguousarray(w, dtype=np.uint32).reshape(-1) for w in (w_l, w_r, w_o)]
        assert idx[0].shape == idx[1].shape == idx[2].shape
        null = ctypes.POINTER(ctypes.c_uint64)()
        u32p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        return self._prepare((null, null, null), idx[0].shape[0], table, pi_pos, pi_vals, blinders, False,
                             variables=(u64p(variables) if variables.size else null, variables.shape[0]),
                             idx=[u32p(x) for x in idx], keep=(variables, idx))

    def prepare_vars_dev(self, d_variables: int, n_vars: int, d_w_l: int, d_w_r: int, d_w_o: int, n_rows: int, table, pi_pos,
                         pi_vals, blinders) -> "PreparedInputs":
        """The composer's witness layout with everything resident in HBM (wires_on_device = 1): `d_variables` may be a
        buffer a device kernel filled, e.g. the states of zkt_poseidon_hash_batch_dev."""
        c64 = lambda p: ctypes.cast(ctypes.c_void_p(p), ctypes.POINTER(ctypes.c_uint64))
        c32 = lambda p: ctypes.cast(ctypes.c_void_p(p), ctypes.POINTER(ctypes.c_uint32))
        null = ctypes.POINTER(ctypes.c_uint64)()
        return self._prepare((null, null, null), n_rows, table, pi_pos, pi_vals, blinders, True,
                             variables=(c64(d_variables), n_vars), idx=[c32(d_w_l), c32(d_w_r), c32(d_w_o)])

    def prove_prepared(self, prep: "PreparedInputs", transcript, next_prep: "PreparedInputs" = None) -> bytes:
        """zkt_prove on prepared inputs.  next_prep announces the proof that follows (zkt_prove_set_next): its
        challenge-free rounds 1 and 2 are issued behind this proof's last commitments."""
        self.check(self._L.zkt_prove_set_next(self._h, ctypes.byref(next_prep.struct) if next_prep is not None else None))
        out = (ctypes.c_uint8 * 2048)()
        n = ctypes.c_size_t(0)
        self.check(self._L.zkt_prove(self._h, ctypes.byref(prep.struct), transcript.handle, out, 2048, ctypes.byref(n)))
        return bytes(out[:n.value])

    def prove_dev(self, d_a: int, d_b: int, d_c: int, n_rows: int, table, pi_pos, pi_vals, blinders, transcript) -> bytes:
        """Same as prove() with the three wire vectors already resident in HBM (device pointers)."""
        return self.prove_prepared(self.prepare_dev(d_a, d_b, d_c, n_rows, table, pi_pos, pi_vals, blinders), transcript)

    def prove(self, a, b, c, table, pi_pos, pi_vals, blinders, transcript) -> bytes:
        """proof_system::prove (prove.rs:59-470); all arrays are (len, 4) Montgomery uint64."""
        return self.prove_prepared(self.prepare_host(a, b, c, table, pi_pos, pi_vals, blinders), transcript)

    def prove_vars(self, variables, w_l, w_r, w_o, table, pi_pos, pi_vals, blinders, transcript) -> bytes:
        """proof_system::prove from the composer's own witness layout (see prepare_vars)."""
        return self.prove_prepared(self.prepare_vars(variables, w_l, w_r, w_o, table, pi_pos, pi_vals, blinders), transcript)

    # -- debug hooks ----------------------------------------------------------------------------------
    def debug_params(self, which: int):
        buf = (ctypes.c_uint32 * 64)()
        n = self._L.zkt_debug_params(self._h, which, buf, 64)
        w = list(buf)
        to_int = lambda l: sum(int(x) << (32 * i) for i, x in enumerate(l))
        return dict(p=to_int(w[0:n]), inv32=int(w[n]), r=to_int(w[n + 1:2 * n + 1]), r2=to_int(w[2 * n + 1:3 * n + 1]))

    def debug_fr_mul(self, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
        out = np.empty_like(a)
        self.check(self._L.zkt_debug_fr_mul(self._h, u64p(a), u64p(b), a.shape[0], u64p(out)))
        return out

    def debug_grand_products(self, n: int, challenges, vectors):
        """z1, z2 evaluation vectors (zkt_debug_grand_products): challenges (4, 4) beta gamma delta epsilon; vectors: seven
        (n, 4) host arrays a b c f t h1 h2."""
        ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(4, 4)
        keep = [np.ascontiguousarray(w, dtype=np.uint64).reshape(n, 4) for w in vectors]
        assert len(keep) == 7
        ptrs = (ctypes.c_void_p * 7)(*[w.ctypes.data for w in keep])
        z1, z2 = np.empty((n, 4), dtype=np.uint64), np.empty((n, 4), dtype=np.uint64)
        self._L.zkt_debug_grand_products.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_void_p),
                                                     ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        self.check(self._L.zkt_debug_grand_products(self._h, u64p(ch), ptrs, u64p(z1), u64p(z2)))
        return z1, z2

    def check_epk_file(self, path: str):
        """zkt_circuit_check_epk_file: None when every vector of the reference CLI's --epk file equals the loaded circuit's
        extended key as the device derives it, else (vector 0..16, first differing element or -1 for a wrong length)."""
        self._L.zkt_circuit_check_epk_file.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int),
                                                       ctypes.POINTER(ctypes.c_size_t)]
        vec, at = ctypes.c_int(0), ctypes.c_size_t(0)
        self.check(self._L.zkt_circuit_check_epk_file(self._h, path.encode(), ctypes.byref(vec), ctypes.byref(at)))
        if vec.value < 0:
            return None
        return vec.value, (-1 if at.value == ctypes.c_size_t(-1).value else at.value)

    def debug_eval_lincomb(self, polys, points, scalars, out_len: int):
        """zkt_debug_eval_lincomb: polys: list of (len_j, 4) arrays, points / scalars (k, 4); returns (evals (k, 4), lincomb
        (out_len, 4)).  Montgomery words throughout."""
        keep = [np.ascontiguousarray(p_, dtype=np.uint64).reshape(-1, 4) for p_ in polys]
        k = len(keep)
        P64 = ctypes.POINTER(ctypes.c_uint64)
        ptrs = (P64 * k)(*[u64p(a) for a in keep])
        lens = (ctypes.c_size_t * k)(*[a.shape[0] for a in keep])
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(k, 4)
        scs = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(k, 4)
        ev = np.empty((k, 4), dtype=np.uint64)
        lc = np.empty((out_len, 4), dtype=np.uint64)
        self._L.zkt_debug_eval_lincomb.argtypes = [ctypes.c_void_p, ctypes.POINTER(P64), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int,
                                                   P64, P64, P64, P64, ctypes.c_size_t]
        self.check(self._L.zkt_debug_eval_lincomb(self._h, ptrs, lens, k, u64p(pts), u64p(scs), u64p(ev), u64p(lc), out_len))
        return ev, lc

    def debug_open_witness(self, coeffs, z) -> np.ndarray:
        """(p(X) - p(z)) / (X - z) (zkt_debug_open_witness): coeffs (len, 4), z (4,), Montgomery words."""
        p_ = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        zz = np.ascontiguousarray(z, dtype=np.uint64).reshape(4)
        out = np.empty((p_.shape[0] - 1, 4), dtype=np.uint64)
        self._L.zkt_debug_open_witness.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t,
                                                   ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        self.check(self._L.zkt_debug_open_witness(self._h, u64p(p_), p_.shape[0], u64p(zz), u64p(out)))
        return out

    def debug_quotient(self, n: int, challenges, wit, pi_pos=(), pi_vals=None) -> np.ndarray:
        """The quotient kernel alone over the loaded circuit (zkt_debug_quotient): challenges (5, 4) alpha beta gamma
        delta epsilon; wit: nine (4n, 4) host arrays a b c pi z1 z2 t h1 h2 (wit[3] may be None with public inputs
        given as positions + values).  Montgomery words throughout."""
        ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(5, 4)
        keep = [None if w is None else np.ascontiguousarray(w, dtype=np.uint64).reshape(4 * n, 4) for w in wit]
        assert len(keep) == 9
        ptrs = (ctypes.c_void_p * 9)(*[None if w is None else w.ctypes.data for w in keep])
        pos = np.ascontiguousarray(list(pi_pos), dtype=np.uint64)
        vals = np.ascontiguousarray(pi_vals if len(pos) else np.zeros((0, 4)), dtype=np.uint64).reshape(-1, 4)
        out = np.empty((4 * n, 4), dtype=np.uint64)
        self.check(self._L.zkt_debug_quotient(self._h, u64p(ch), ptrs, u64p(pos) if len(pos) else None,
                                              u64p(vals) if len(pos) else None, len(pos), u64p(out)))
        return out
