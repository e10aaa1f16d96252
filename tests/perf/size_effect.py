#!/usr/bin/env python3
"""Why is the ten-client chained encrypt 9-12 % slower per element at n >= 2e7 than at 1e7?  Separates sustained-clock effects
from footprint effects: (A) back-to-back launches at n = 1e7, per-launch times; (B) one n = 4e7 launch; (C) the same 4e7 vectors
as four range launches of 1e7; (D) n = 1e7 launches with idle gaps."""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flashe_amd.engine import Engine

KEY = bytes(range(32))
C, J = 10, 16


def main():
    eng = Engine(KEY, 128, device=0)
    N = 40_000_000
    pts = [eng.alloc(N * 8) for _ in range(C)]
    cts = [eng.alloc(N * 16) for _ in range(C)]
    evs = [eng.event() for _ in range(130)]

    def launch(n, first, cnt):
        eng.prf_jobs_dev(1, n, J, [(c, c + 1, first, cnt, pts[c].ptr + first * 8, 1, cts[c].ptr + first * 16) for c in range(C)])

    def series(label, n, ranges, reps, gap=0.0):
        for _ in range(3):
            launch(n, *ranges[0])
        eng.sync()
        k = 0
        eng.record(evs[0])
        for r in range(reps):
            for (f, c) in ranges:
                launch(n, f, c)
                k += 1
                eng.record(evs[k])
            if gap:
                eng.sync(); time.sleep(gap); k += 1; eng.record(evs[k])
        eng.sync()
        ts = [eng.elapsed_ms(evs[i], evs[i + 1]) for i in range(k)]
        if gap:
            ts = ts[0::2]
        per = [t / (c / 1e6) * 1e3 for t, (f, c) in zip(ts, ranges * reps)]
        print(f"{label:58s} us per M elements: first {per[0]:.1f}  " + " ".join(f"{p:.0f}" for p in per[1:]), flush=True)

    if len(sys.argv) > 1 and sys.argv[1] == "short":
        series("A  n=1e7, 12 launches back to back", 10_000_000, [(0, 10_000_000)], 12)
        series("B  n=4e7, one launch x 4", N, [(0, N)], 4)
        series("E  n=4e7 as eight ranges of 5e6", N, [(q * 5_000_000, 5_000_000) for q in range(8)], 1)
        return
    series("A  n=1e7, 40 launches back to back", 10_000_000, [(0, 10_000_000)], 40)
    series("D  n=1e7, 12 launches, 20 ms idle between", 10_000_000, [(0, 10_000_000)], 12, gap=0.02)
    series("B  n=4e7, one launch x 6", N, [(0, N)], 6)
    series("C  n=4e7 as four ranges of 1e7 x 3", N, [(q * 10_000_000, 10_000_000) for q in range(4)], 3)
    series("E  n=4e7 as eight ranges of 5e6 x 2", N, [(q * 5_000_000, 5_000_000) for q in range(8)], 2)
    series("F  n=2e7, one launch x 8", 20_000_000, [(0, 20_000_000)], 8)
    series("A' n=1e7 again", 10_000_000, [(0, 10_000_000)], 12)


main()
