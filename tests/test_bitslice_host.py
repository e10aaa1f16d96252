"""CPU-only: the generated bit-sliced AES device code, compiled for the host with software models of
v_bitop3_b32 / v_perm_b32, must match libcrypto AES-256 (tests/host_bitslice_check.cpp)."""
import os
import subprocess

from conftest import ROOT


def test_generated_bitslice_code_on_host(tmp_path):
    exe = tmp_path / "bs_check"
    src = os.path.join(ROOT, "tests", "host_bitslice_check.cpp")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "flashe_amd", "csrc"), src,
                           "-o", str(exe), "-lcrypto"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
