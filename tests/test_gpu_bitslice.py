"""The bit-sliced AES PRF backends (v_bitop3_b32 netlists, tools/bitslice/) are measured alternatives, not what runs: they are built
only by `make -C flashe_amd/csrc bitslice` into libflashe_hip_bitslice.so.  When that library is present, the backend-parametrised
parity tests of tests/test_gpu_parity.py run against it here, in a process that loads it instead of the product library."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bitsliced_backends_against_the_oracle():
    lib = os.path.join(ROOT, "flashe_amd", "libflashe_hip_bitslice.so")
    if not os.path.exists(lib):
        pytest.skip("libflashe_hip_bitslice.so not built (make -C flashe_amd/csrc bitslice)")
    env = dict(os.environ, FLASHE_LIB_NAME="libflashe_hip_bitslice.so", FLASHE_TEST_BITSLICE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        "-k", "bitsliced_prf_backend or counter_window_across"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
