"""TEST WRAPPER (tests/test_gpu_parity.py): runs bench.py's own N > 1 flow -- process spawn, watchdog, sequential round first,
calibration of the overlapped schedules, the one JSON line -- with several ranks on ONE GPU.  bench.main() is called with its two
test seams: a file-based double of RcclComm (tests/shm_comm.py, directory from BENCH_SHM_DIR) instead of RCCL, and device 0 for
every rank.  The printed figure is meaningless and labelled so.

BENCH_SHM_INJECT=raise:R | hang:R makes rank R raise / block forever inside the first overlapped (fused) round -- what a first
multi-GPU run may meet -- so that the tests can check the fallback: a valid sequential line within the deadline, exit code 0.
raise_elements:R | hang_elements:R do the same inside the element-sharded phase that follows the main line.
BENCH_SHM_DEVICES=N: the preflight of every rank sees N visible devices (needs no GPU: it fails before an engine exists)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bench
    from shm_comm import ShmComm
    inject = os.environ.get("BENCH_SHM_INJECT")
    if inject and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        kind, r = inject.split(":")
        if int(os.environ.get("RANK", "0")) == int(r):
            from flashe_amd.dist import ShardedRound

            def broken(self, *a, **k):
                if kind == "raise":
                    raise RuntimeError("injected failure of an optional schedule")
                time.sleep(10 ** 6)
            if kind.endswith("_elements"):                       # the element-sharded phase, timed AFTER the main line exists
                kind = kind[:-len("_elements")]
                ShardedRound.run_elements = broken
            else:
                ShardedRound.run_fused = broken
    fake = os.environ.get("BENCH_SHM_DEVICES")                     # the preflight's view of the machine: "2 devices visible" with 3 ranks
    bench.main(comm_factory=lambda rank, world: ShmComm(rank, world, os.environ["BENCH_SHM_DIR"]), device_override=0,
               devices_override=int(fake) if fake else None)


if __name__ == "__main__":
    main()
