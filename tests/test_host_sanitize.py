"""Host-side hygiene: the library built for the HOST only with AddressSanitizer and
UndefinedBehaviorSanitizer (zkt-plonk_amd/build.py build_host_sanitized) runs the CPU tests of everything that never
touches a device -- transcripts, host inversion, field and curve arithmetic, the communicator plumbing, the C-ABI
export check.  A sanitizer report aborts the child process."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_entry_points_under_asan_and_ubsan():
    spec = importlib.util.spec_from_file_location("zkt_build", os.path.join(ROOT, "zkt-plonk_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    lib = b.build_host_sanitized()
    rt = b.asan_runtime()
    assert os.path.exists(lib) and os.path.exists(rt)
    env = dict(os.environ, ZKT_LIB_PATH=lib, LD_PRELOAD=rt, ZKT_SKIP_SLOW_ORACLE="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    tests = ["tests/test_field_host.py", "tests/test_transcript_host.py", "tests/test_cabi.py", "tests/test_keyfile_host.py", "tests/test_verify_host.py", "tests/test_pairing_host.py::test_pairing_smoke",
             "tests/test_pairing_host.py::test_batch_verifier_folds_proofs_of_different_circuits_into_one_pairing_product",
             "tests/test_parallel_gloo.py::test_g1_sum_host_matches_oracle",
             "tests/test_parallel_gloo.py::test_shard_range_covers_everything"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-6000:]
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-6000:]
    assert " passed" in r.stdout
