"""Builds libzkt_plonk_hip.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only authoring container.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libzkt_plonk_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-psabi", "-ffp-contract=off"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") or f.endswith(".cpp"))


def _headers_mtime():
    m = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith((".hpp", ".h")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _compile(src, hdr_m, force, extra, obj_dir=None):
    obj = os.path.join(obj_dir or OBJ, src + ".o")
    sp = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(sp), hdr_m):
        return obj, False
    cmd = ["hipcc"] + FLAGS + extra + ["-c", sp, "-o", obj]
    if src.endswith(".cpp"):
        cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "-Wall", "-x", "c++", "-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force: bool = False, extra_flags=None, jobs: int = 4, lib: str = None, obj_dir: str = None) -> str:
    lib = lib or LIB
    obj_dir = obj_dir or OBJ
    os.makedirs(obj_dir, exist_ok=True)
    hdr_m = _headers_mtime()
    extra = list(extra_flags or [])
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(lambda s: _compile(s, hdr_m, force, extra, obj_dir), srcs))
    objs = [o for o, _ in res]
    # relink whenever an object is newer than the library (an object compiled by hand counts too)
    stale = not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs)
    if force or stale or any(ch for _, ch in res):
        cmd = ["hipcc", "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return lib


COMM_SRC = os.path.join(HERE, "csrc_comm", "comm_rccl.cpp")
COMM_LIB = os.path.join(HERE, "libzkt_comm_rccl.so")


def build_comm(force: bool = False) -> str:
    """libzkt_comm_rccl.so (include/zkt_comm_rccl.h): the optional RCCL transport; host code, links librccl."""
    hdr = os.path.join(HERE, "..", "include", "zkt_comm_rccl.h")
    src_m = max(os.path.getmtime(COMM_SRC), os.path.getmtime(hdr), _headers_mtime())
    if force or not os.path.exists(COMM_LIB) or os.path.getmtime(COMM_LIB) < src_m:
        cmd = ["hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", COMM_SRC, "-o", COMM_LIB, "-L/opt/rocm/lib", "-lrccl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building the RCCL transport failed:\n%s\n%s" % (r.stdout, r.stderr))
    return COMM_LIB


def build_experiments(force: bool = False, extra_flags=None) -> str:
    """The A/B variant: same sources with -DZKT_EXPERIMENTS (environment knobs honoured, csrc/ctx.hpp exp_env) into
    _ab/libzkt_exp.so; run it with ZKT_LIB_PATH.  Never loaded by default."""
    ab = os.path.join(HERE, "..", "_ab")
    os.makedirs(ab, exist_ok=True)
    return build(force, ["-DZKT_EXPERIMENTS"] + list(extra_flags or []), lib=os.path.join(ab, "libzkt_exp.so"),
                 obj_dir=os.path.join(HERE, "_obj_exp"))


# ---- host-side sanitizer build ------------------------------------------------------------------------------------
# Every source compiled for the HOST only (--cuda-host-only: kernels become stubs that cannot launch) with
# AddressSanitizer + UndefinedBehaviorSanitizer.  The entry points that never touch a device -- transcripts
# (transcript.cpp), host inversion and field / curve arithmetic (hostinv.hpp, fp.hpp, fx.hpp, ec.hpp through
# zkt_host_field_op / zkt_g1_sum_host), the communicator plumbing -- then run under the CPU tests
# (tests/test_host_sanitize.py).  GPU-side sanitizers are not available on this pool.
ASAN_OBJ = os.path.join(HERE, "_obj_asan")
ASAN_LIB = os.path.join(HERE, "libzkt_plonk_host_asan.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
ASAN_FLAGS = ["-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
              "-fno-omit-frame-pointer", "-Wno-psabi", "-Wno-unused-function", "-Wno-unused-variable", "-ffp-contract=off"]


def asan_runtime() -> str:
    r = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    return r.stdout.strip()


def _compile_asan(src, hdr_m, force):
    obj = os.path.join(ASAN_OBJ, src + ".o")
    sp = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(sp), hdr_m):
        return obj, False
    lang = ["-x", "hip", "--cuda-host-only", "-I/opt/rocm/include"] if src.endswith(".hip") else ["-x", "c++"]
    r = subprocess.run([CLANG] + lang + ASAN_FLAGS + ["-c", sp, "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("sanitizer build failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj, True


def build_host_sanitized(force: bool = False, jobs: int = 4) -> str:
    os.makedirs(ASAN_OBJ, exist_ok=True)
    hdr_m = _headers_mtime()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(lambda s: _compile_asan(s, hdr_m, force), _sources()))
    if force or any(ch for _, ch in res) or not os.path.exists(ASAN_LIB):
        # a host-only object still registers "its" device code: give every such reference an empty offload bundle
        # (magic + zero entries), so that the module constructors have something well-formed to hand to the runtime
        syms = set()
        for o, _ in res:
            nm = subprocess.run(["nm", "-u", o], capture_output=True, text=True).stdout
            syms.update(l.split()[-1] for l in nm.splitlines() if "__hip_fatbin_" in l and "wrapper" not in l)
        stub = os.path.join(ASAN_OBJ, "fatbin_stub.c")
        with open(stub, "w") as f:
            for sname in sorted(syms):
                f.write('const unsigned char %s[32] __attribute__((aligned(4096))) = "__CLANG_OFFLOAD_BUNDLE__";\n' % sname)
        stub_o = stub + ".o"
        subprocess.check_call(["gcc", "-fPIC", "-c", stub, "-o", stub_o])
        cmd = [CLANG, "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", "-o", ASAN_LIB] + \
              [o for o, _ in res] + [stub_o, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("sanitizer link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return ASAN_LIB


if __name__ == "__main__":
    print(build_experiments(force="--force" in sys.argv) if "--exp" in sys.argv else build(force="--force" in sys.argv))
