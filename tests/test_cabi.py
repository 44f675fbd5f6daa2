"""CPU checks of the drop-in boundary: the library loads and exports every symbol the public header
declares.  No compute calls (no GPU here)."""
import ctypes
import os

import pytest

import zkt_plonk_amd as z


def test_library_exports_every_declared_symbol():
    L = z.lib()
    syms = z.declared_symbols()
    assert "zkt_ntt" in syms and "zkt_ctx_create" in syms
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_version_and_no_cpu_fallback():
    L = z.lib()
    assert b"gfx950" in L.zkt_version()
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(z.ZktError) as e:
            z.Context("bn254")
        assert e.value.code == 4  # ZKT_ERR_NO_DEVICE: the product never computes on the CPU


def test_product_does_not_import_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "zkt-plonk_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "oracle/" not in text, f


def test_public_header_is_plain_c():
    """The drop-in boundary is a C ABI: include/zkt_plonk.h must compile as C99 on its own (no C++, no torch types)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "zkt_plonk.h")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import re
    code = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)      # comments may mention torch.distributed
    assert "torch" not in code and "std::" not in code and "class " not in code


def test_rccl_transport_library_exports_its_header():
    """include/zkt_comm_rccl.h (the optional transport a non-Python host links): plain C, every declared function exported by
    libzkt_comm_rccl.so, which links RCCL and the HIP runtime and nothing of PyTorch; the main library links no transport."""
    import re
    import subprocess
    from conftest import rccl_transport_or_skip
    rccl_transport_or_skip()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "zkt_comm_rccl.h")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(zkt_comm_rccl_[a-z0-9_]+)\s*\(", text)))
    assert declared == ["zkt_comm_rccl_create", "zkt_comm_rccl_destroy", "zkt_comm_rccl_last_error", "zkt_comm_rccl_unique_id",
                        "zkt_comm_rccl_vtable"]
    from zkt_plonk_amd import parallel as par
    L = par.rccl_lib()
    assert all(hasattr(L, s) for s in declared)
    so = os.path.join(root, "zkt-plonk_amd", "libzkt_comm_rccl.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "librccl.so" in needed and "libamdhip64.so" in needed and "torch" not in needed
    main_needed = subprocess.run(["readelf", "-d", z.lib_path()], capture_output=True, text=True).stdout
    assert "rccl" not in main_needed and "mpi" not in main_needed
