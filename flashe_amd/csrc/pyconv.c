/* Host-side converter between the reference's data convention -- 1-D numpy object arrays of Python ints
 * (jzf_flashe.py:480-481 computes on those) -- and the engine's little-endian uint64 limb arrays.  NumPy's own
 * astype() goes through the generic number protocol (~65 ns per element); reading the ints directly is 4-6x faster,
 * which is what the class-level API (FlasheCipher.encrypt / decrypt / aggregate on object arrays) is bound by.
 *
 * Built as flashe_amd/_pyconv.so and loaded with ctypes.PyDLL (the GIL is held during the calls); it links nothing:
 * the CPython symbols come from the running interpreter.  No arithmetic of the cipher happens here. */
#define _GNU_SOURCE
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

#define FAST_DIGITS 5          /* 30-bit digits read / written in place: values below 2^150 */

/* Threads of the parallel read below: the CPUs this process may really use -- affinity mask capped by the cgroup CPU quota -- and at
 * most 16.  OpenMP's own default is one thread per hardware thread of the HOST: a container with a 16-CPU quota on a 256-thread
 * machine then runs 256 threads in 16 CPUs' worth of time slices, and a 262,144-element call that takes 2 ms took 10-100 ms
 * (tests/perf/notebook_table2.py on the GPU box).  FLASHE_PYCONV_THREADS overrides. */
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
static int pyconv_threads(void)
{
    static int cached = 0;
    if (cached) return cached;
    int n = 1;
    const char *e = getenv("FLASHE_PYCONV_THREADS");
    if (e && atoi(e) > 0) { cached = atoi(e); return cached; }
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");                       /* cgroup v2: "<quota|max> <period>" */
    if (f) {
        char q[32]; long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && period > 0 && strcmp(q, "max") != 0) {
            const long c = atol(q) / period;
            if (c >= 1 && c < n) n = (int)c;
        }
        fclose(f);
    } else {
        long quota = 0, period = 0;                                         /* cgroup v1 */
        FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fp && fscanf(fq, "%ld", &quota) == 1 && fscanf(fp, "%ld", &period) == 1 && quota > 0 && period > 0) {
            const long c = quota / period;
            if (c >= 1 && c < n) n = (int)c;
        }
        if (fq) fclose(fq);
        if (fp) fclose(fp);
    }
    if (n > 16) n = 16;
    if (n < 1) n = 1;
    cached = n;
    return n;
}

/* value mod 2^(64 * limbs) of any Python int (negative values wrap like Python's `&`), two's complement */
static int int_to_limbs(PyObject *o, int limbs, uint64_t *out)
{
    if (!PyLong_Check(o)) {                       /* numpy integer scalars held in the object array, bools, ... */
        PyObject *idx = PyNumber_Index(o);
        if (!idx) return -1;
        int rc = int_to_limbs(idx, limbs, out);
        Py_DECREF(idx);
        return rc;
    }
    int overflow = 0;
    long long v = PyLong_AsLongLongAndOverflow(o, &overflow);
    if (!overflow) {
        if (v == -1 && PyErr_Occurred()) return -1;
        out[0] = (uint64_t)v;
        if (limbs == 2) out[1] = v < 0 ? ~0ull : 0ull;
        return 0;
    }
    /* beyond int64: the low 64 * limbs bits of the two's complement representation */
#if PY_VERSION_HEX >= 0x030D0000
    /* CPython 3.13 changed the private _PyLong_AsByteArray / _PyLong_NumBits signatures and added a public call that does exactly
     * this: the low 16 bytes of the two's complement value, little endian, whatever the magnitude */
    {
        unsigned char nb[16];
        if (PyLong_AsNativeBytes(o, nb, 16, Py_ASNATIVEBYTES_LITTLE_ENDIAN) < 0) return -1;
        memcpy(&out[0], nb, 8);
        if (limbs == 2) memcpy(&out[1], nb + 8, 8);
        return 0;
    }
#else
    unsigned char buf[17];
    size_t nbits = _PyLong_NumBits(o);
    if (nbits == (size_t)-1 && PyErr_Occurred()) return -1;
    if (nbits < 128) {
        if (_PyLong_AsByteArray((PyLongObject *)o, buf, 17, 1, 1) < 0) return -1;
        memcpy(&out[0], buf, 8);
        if (limbs == 2) memcpy(&out[1], buf + 8, 8);
        return 0;
    }
    /* wider than the modulus: mask in Python, then convert */
    PyObject *mask = PyLong_FromString(limbs == 2 ? "ffffffffffffffffffffffffffffffff" : "ffffffffffffffff", NULL, 16);
    if (!mask) return -1;
    PyObject *low = PyNumber_And(o, mask);
    Py_DECREF(mask);
    if (!low) return -1;
    int rc = _PyLong_AsByteArray((PyLongObject *)low, buf, 17, 1, 0);
    Py_DECREF(low);
    if (rc < 0) return -1;
    memcpy(&out[0], buf, 8);
    if (limbs == 2) memcpy(&out[1], buf + 8, 8);
    return 0;
#endif
}

/* objs: the data of a 1-D numpy object array (n borrowed references); out: [n][limbs].  0 = ok, -1 = Python error set.
 *
 * Large arrays are read by several threads: the common case -- exact ints below 2^60 -- needs no CPython call at all (type pointer,
 * size and digits are plain memory reads of immutable objects that the array keeps alive, and the calling thread holds the GIL for
 * the whole call, so no Python code runs meanwhile); the workers never touch the interpreter.  The loop is one cache miss per element
 * (the ints live wherever the allocator put them), which is exactly what more threads overlap.  Whatever is not such an int (wider
 * values, negative ones, NumPy scalars) is left to a serial second pass through the C API. */
int flashe_pyconv_ints_to_limbs(PyObject **objs, Py_ssize_t n, int limbs, uint64_t *out)
{
    if (limbs != 1 && limbs != 2) { PyErr_SetString(PyExc_ValueError, "limbs must be 1 or 2"); return -1; }
    int leftovers = 0;
#if PY_VERSION_HEX < 0x030C0000
    /* (the in-place digit read relies on the PyLongObject layout of CPython <= 3.11: ob_size + ob_digit; 3.12 moved it to
     * long_value.lv_tag, where the generic path below takes everything) */
    const int nthreads = n >= 131072 ? pyconv_threads() : 1;
#pragma omp parallel for schedule(static) reduction(| : leftovers) num_threads(nthreads) if (nthreads > 1)
    for (Py_ssize_t i = 0; i < n; i++) {
        if (i + 16 < n) __builtin_prefetch(objs[i + 16], 0, 0);
        PyObject *o = objs[i];
        uint64_t *dst = out + (size_t)i * limbs;
        if (PyLong_CheckExact(o)) {
            /* non-negative ints of up to FAST_DIGITS digits (30-bit digits: < 2^150, i.e. every 128-bit ciphertext): read the digits in
             * place, reduced mod 2^(64 limbs) by the truncation of the accumulator */
            const Py_ssize_t sz = Py_SIZE(o);
            const digit *d = ((PyLongObject *)o)->ob_digit;
            if (sz >= 0 && sz <= FAST_DIGITS) {
                unsigned __int128 v = 0;
                for (Py_ssize_t j = 0; j < sz; j++) v |= (unsigned __int128)d[j] << (PyLong_SHIFT * j);
                dst[0] = (uint64_t)v;
                if (limbs == 2) dst[1] = (uint64_t)(v >> 64);
                continue;
            }
        }
        leftovers = 1;
    }
    if (!leftovers) return 0;
#else
    leftovers = 1;
#endif
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject *o = objs[i];
#if PY_VERSION_HEX < 0x030C0000
        if (PyLong_CheckExact(o)) { const Py_ssize_t sz = Py_SIZE(o); if (sz >= 0 && sz <= FAST_DIGITS) continue; }      /* done above */
#endif
        if (int_to_limbs(o, limbs, out + (size_t)i * limbs) < 0) return -1;
    }
    return 0;
}

/* in: [n][limbs]; objs: the data of a 1-D numpy object array whose n slots hold owned references (np.empty(n, object) holds None):
 * each slot is replaced by a new int. */
int flashe_pyconv_limbs_to_ints(const uint64_t *in, Py_ssize_t n, int limbs, PyObject **objs)
{
    if (limbs != 1 && limbs != 2) { PyErr_SetString(PyExc_ValueError, "limbs must be 1 or 2"); return -1; }
    for (Py_ssize_t i = 0; i < n; i++) {
        const uint64_t *p = in + (size_t)i * limbs;
        PyObject *v;
        if (limbs == 1 || p[1] == 0) v = PyLong_FromUnsignedLongLong(p[0]);
#if PY_VERSION_HEX < 0x030C0000 && PYLONG_BITS_IN_DIGIT == 30
        else {
            /* a 65..128-bit value: allocate the int with its digit count and write the 30-bit digits directly (the generic byte-array
             * constructor costs ~4x as much) */
            const unsigned __int128 x = ((unsigned __int128)p[1] << 64) | p[0];
            const int bits = 128 - __builtin_clzll(p[1]);
            const int nd = (bits + PyLong_SHIFT - 1) / PyLong_SHIFT;
            PyLongObject *lv = _PyLong_New(nd);
            if (lv) for (int j = 0; j < nd; j++) lv->ob_digit[j] = (digit)((x >> (PyLong_SHIFT * j)) & PyLong_MASK);
            v = (PyObject *)lv;
        }
#elif PY_VERSION_HEX >= 0x030D0000
        else v = PyLong_FromUnsignedNativeBytes(p, 16, Py_ASNATIVEBYTES_LITTLE_ENDIAN);
#else
        else v = _PyLong_FromByteArray((const unsigned char *)p, 16, 1, 0);
#endif
        if (!v) return -1;
        PyObject *old = objs[i];
        objs[i] = v;
        Py_XDECREF(old);
    }
    return 0;
}
