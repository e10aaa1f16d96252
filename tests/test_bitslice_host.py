"""CPU-only: the generated bit-sliced AES device code, compiled for the host with software models of
v_bitop3_b32 / v_perm_b32, must match libcrypto AES-256 (tests/host_bitslice_check.cpp)."""
import os
import subprocess
import sys

from conftest import ROOT


def test_generated_bitslice_code_on_host(tmp_path):
    exe = tmp_path / "bs_check"
    src = os.path.join(ROOT, "tests", "host_bitslice_check.cpp")
    # the generated header is not tracked: run the generator (it verifies its own netlists first)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "bitslice", "gen_bitslice.py"), str(tmp_path / "aes_bitslice_gen.h")],
                          stdout=subprocess.DEVNULL)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", str(tmp_path), "-I", os.path.join(ROOT, "flashe_amd", "csrc", "bitslice"), src,
                           "-o", str(exe), "-lcrypto"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
