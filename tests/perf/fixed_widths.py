#!/usr/bin/env python3
"""The compact layout (uint32 arrays, int_bits <= 32) at every width in FLASHE_FIXED32_WIDTHS (csrc/kernels.hip): bench.py's config-2
round with the compile-time-width kernel and with the run-time-width one (FLASHE_SMALL_FIXED=0, tuning library only).
usage: fixed_widths.py [widths, comma separated]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(bits, fixed):
    env = dict(os.environ, FLASHE_LIB_NAME="libflashe_hip_tuning.so", FLASHE_SMALL_FIXED=str(fixed))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bits", str(bits), "--layout", "u32", "--no-cpu-baseline", "--no-e2e",
                        "--steps", "30", "--warmup", "10"], capture_output=True, text=True, env=env, timeout=600)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            d = json.loads(line)
            return d["ms_per_step"], d.get("ms_per_step_two_launch")
    raise RuntimeError(r.stdout[-500:] + r.stderr[-1500:])


def main():
    widths = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 20, 23, 24, 32]
    for b in widths:
        a, a2 = run(b, 1)
        g, g2 = run(b, 0)
        print(f"b={b:2d} m={128 // b:2d}  fixed {a:.4f} ms (two launches {a2:.4f})   run-time width {g:.4f} ms ({g2:.4f})   {100 * (a / g - 1):+.1f} %", flush=True)


if __name__ == "__main__":
    main()
