#!/bin/bash
# CPU-only sanitizer pass over the host-side C of the repository (GPU AddressSanitizer is not available on the pool):
# the oracle (oracle/flashe_oracle.c) and the object-array converter (flashe_amd/csrc/pyconv.c) are rebuilt with
# -fsanitize=address,undefined and driven by the CPU test-suite files that exercise them; the normal builds are restored afterwards.
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
ASAN=$(gcc -print-file-name=libasan.so)
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
trap 'make -s -C oracle clean all; make -s -C flashe_amd/csrc ../_pyconv.so -B >/dev/null' EXIT
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -fopenmp -std=gnu11 -ffp-contract=off -maes -msse4.1 \
    -shared -o oracle/libflashe_oracle.so oracle/flashe_oracle.c -lm
gcc -O1 -g -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -I"$PYINC" -o flashe_amd/_pyconv.so flashe_amd/csrc/pyconv.c
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1 OMP_NUM_THREADS=2 \
    python3 -m pytest -x -q tests/test_oracle_golden.py tests/test_cipher_host_logic.py tests/test_properties.py -m "not gpu" -p no:cacheprovider
# the block pools behind flashe_dev_alloc / the host-pointer twins' staging (flashe_amd/csrc/blockpool.h) with a mock device
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -Iflashe_amd/csrc tests/host_blockpool_check.cpp -o /tmp/flashe_blockpool_check
ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1 /tmp/flashe_blockpool_check
echo "ASAN_CPU_OK"
