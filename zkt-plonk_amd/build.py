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


def _compile(src, hdr_m, force, extra):
    obj = os.path.join(OBJ, src + ".o")
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


def build(force: bool = False, extra_flags=None, jobs: int = 4) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = _headers_mtime()
    extra = list(extra_flags or [])
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(lambda s: _compile(s, hdr_m, force, extra), srcs))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = ["hipcc", "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
