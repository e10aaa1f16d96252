"""CPU-only: the caching device allocator behind flashe_dev_alloc / flashe_dev_free (flashe_amd/csrc/blockpool.h) against a mock
backend, built with AddressSanitizer + UBSan (tests/host_blockpool_check.cpp)."""
import os
import subprocess

from conftest import ROOT


def test_blockpool_invariants_under_sanitizers(tmp_path):
    exe = tmp_path / "blockpool_check"
    src = os.path.join(ROOT, "tests", "host_blockpool_check.cpp")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-I", os.path.join(ROOT, "flashe_amd", "csrc"), src, "-o", str(exe)])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "BLOCKPOOL_OK" in r.stdout, r.stdout + r.stderr[-3000:]
