"""A communicator double for -m gpu tests of the N > 1 path on ONE GPU: several processes share device 0, every rank runs the real
HIP kernels through HipOps, and the exchange -- where production uses RCCL (flashe_rccl_*) -- goes through files in a shared
directory (device -> host -> file -> host -> device).  Same method surface as flashe_amd.dist.RcclComm.  Test infrastructure."""
import os
import time

import numpy as np


class ShmComm:
    MAX, MIN, SUM = 0, 1, 2
    IS_TEST_DOUBLE = True
    LABEL = "TEST DOUBLE: files, all ranks on one GPU (figures meaningless)"

    def __init__(self, rank, world, directory):
        self.rank, self.world, self.dir = rank, world, directory
        self.seq = 0

    # ---- file plumbing ----
    def _put(self, name, data):
        tmp = os.path.join(self.dir, f".{name}.{self.rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, os.path.join(self.dir, name))
        self.__dict__.setdefault("_mine", []).append((self.seq, name))

    def _sweep(self):
        """Drop this rank's files of collectives <= seq - 2: a rank that has finished collective s - 1 has read every peer's s - 1 file, so
        every peer had started s - 1, i.e. finished reading s - 2 (full-size vectors would otherwise pile up gigabytes in the directory)."""
        keep = []
        for seq, name in self.__dict__.get("_mine", []):
            if seq <= self.seq - 2:
                try:
                    os.unlink(os.path.join(self.dir, name))
                except OSError:
                    pass
            else:
                keep.append((seq, name))
        self._mine = keep

    def _take(self, name):
        """A file only this rank reads (its piece of an all-to-all): read, then removed."""
        data = self._get(name)
        try:
            os.unlink(os.path.join(self.dir, name))
        except OSError:
            pass
        return data

    def _get(self, name, timeout=60.0):
        path = os.path.join(self.dir, name)
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > timeout:
                raise TimeoutError(f"rank {self.rank}: waited {timeout} s for {name}")
            time.sleep(0.0005)
        with open(path, "rb") as f:
            return f.read()

    @staticmethod
    def _d2h(engine, ptr, nbytes):
        out = np.empty(nbytes, dtype=np.uint8)
        engine._check(engine._lib.flashe_memcpy_d2h(engine._h, out.ctypes.data, ptr, nbytes))
        return out

    @staticmethod
    def _h2d(engine, ptr, arr):
        arr = np.ascontiguousarray(arr)
        engine._check(engine._lib.flashe_memcpy_h2d(engine._h, ptr, arr.ctypes.data, arr.nbytes))

    # ---- RcclComm surface ----
    def all_to_all(self, engine, send_ptr, send_stride, recv_ptr, recv_stride, nbytes):
        s = self.seq = self.seq + 1
        self._sweep()
        for p in range(self.world):
            self._put(f"a2a_{s}_{self.rank}_{p}", self._d2h(engine, send_ptr + p * send_stride, nbytes).tobytes())
        self._mine = [(q, nm) for q, nm in self._mine if not nm.startswith(f"a2a_{s}_")]      # (the receivers remove these)
        for p in range(self.world):
            self._h2d(engine, recv_ptr + p * recv_stride, np.frombuffer(self._take(f"a2a_{s}_{p}_{self.rank}"), dtype=np.uint8))

    def all_gather(self, engine, send_ptr, recv_ptr, nbytes):
        s = self.seq = self.seq + 1
        self._sweep()
        self._put(f"ag_{s}_{self.rank}", self._d2h(engine, send_ptr, nbytes).tobytes())
        for p in range(self.world):
            self._h2d(engine, recv_ptr + p * nbytes, np.frombuffer(self._get(f"ag_{s}_{p}"), dtype=np.uint8))

    def allreduce_modadd(self, engine, ptr, count):
        s = self.seq = self.seq + 1
        self._sweep()
        self._put(f"arm_{s}_{self.rank}", self._d2h(engine, ptr, 8 * count).tobytes())
        tot = np.zeros(count, dtype=np.uint64)
        for p in range(self.world):
            tot += np.frombuffer(self._get(f"arm_{s}_{p}"), dtype=np.uint64)
        b = engine.int_bits
        if b < 64:
            tot &= np.uint64((1 << b) - 1)
        self._h2d(engine, ptr, tot)

    def allreduce(self, engine, value, op=0):
        engine.sync()
        s = self.seq = self.seq + 1
        self._put(f"ar_{s}_{self.rank}", np.float64(value).tobytes())
        vals = [float(np.frombuffer(self._get(f"ar_{s}_{p}"), dtype=np.float64)[0]) for p in range(self.world)]
        return max(vals) if op == 0 else min(vals) if op == 1 else sum(vals)

    def barrier(self, engine):
        self.allreduce(engine, 0.0, 2)

    def close(self):
        pass
