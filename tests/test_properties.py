"""Property tests of the cipher algebra (round trips for every scheme x int_bits x
chunking x dropout pattern).  The CPU variant exercises the oracle, the GPU variant the HIP engine through
the C ABI; both check the same statement: for ANY set of uploading clients,

    decrypt(aggregate({encrypt_i(pt_i)}), telescoped prefixes of the uploaded set) == sum_i pt_i  mod 2^b

and linearity of the mask stream in the prefix list."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

KEY = bytes(range(32))


def L(b):
    return 2 if b > 64 else 1


def limbs(rng, n, b):
    out = rng.integers(0, 2 ** 64, size=(n, L(b)), dtype=np.uint64)
    if b > 64 and b < 128:
        out[:, 1] &= np.uint64((1 << (b - 64)) - 1)
    elif b < 64:
        out[:, 0] &= np.uint64((1 << b) - 1)
    return out


def to_int(arr):
    return [int(r[0]) | (int(r[1]) << 64 if len(r) == 2 else 0) for r in arr]


case = st.tuples(st.integers(1, 128), st.integers(0, 300), st.integers(1, 20), st.integers(0, 2 ** 32 - 1),
                 st.lists(st.integers(0, 12), min_size=1, max_size=8), st.sampled_from(["double", "single"]), st.integers(0, 2 ** 31))


def check(enc, agg, dec, telescope, b, n, J, it, uploaded, scheme, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    clients = sorted(set(uploaded))
    pts = {c: limbs(rng, n, b) for c in clients}
    cts = {c: enc(it, c, scheme, J, b, pts[c]) for c in clients}
    total = agg([cts[c] for c in uploaded], b) if n else np.zeros((0, L(b)), dtype=np.uint64)
    if scheme == "double":
        add_idx, minus_idx = telescope(list(uploaded))
    else:
        add_idx, minus_idx = [], list(uploaded)
    out = dec(it, add_idx, minus_idx, J, b, total)
    mod = 1 << b
    want = [sum(v) % mod for v in zip(*[to_int(pts[c]) for c in uploaded])] if n else []
    assert to_int(out) == want


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(case)
def test_round_trip_oracle(oracle, c):
    b, n, J, it, uploaded, scheme, seed = c
    check(lambda it, i, s, J, b, pt: oracle.encrypt(KEY, it, i, s, J, b, pt), oracle.aggregate_elem,
          lambda it, a, m, J, b, ct: oracle.decrypt(KEY, it, a, m, J, b, ct), oracle.telescope, b, n, J, it, uploaded, scheme, seed)


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(st.integers(1, 128), st.integers(1, 200), st.integers(1, 16), st.lists(st.integers(0, 2 ** 32 - 1), min_size=0, max_size=5),
       st.lists(st.integers(0, 2 ** 32 - 1), min_size=0, max_size=5))
def test_mask_sum_is_linear_oracle(oracle, b, n, J, la, lb):
    a = oracle.mask_sum(KEY, 7, la, n, J, b)
    c = oracle.mask_sum(KEY, 7, lb, n, J, b)
    both = oracle.mask_sum(KEY, 7, la + lb, n, J, b)
    assert np.array_equal(both, oracle.combine(b, a, c, None))


@pytest.mark.gpu
@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(case)
def test_round_trip_engine(c):
    from flashe_amd import engine as E
    b, n, J, it, uploaded, scheme, seed = c
    eng = E.Engine(KEY, b)
    sch = {"double": E.SCHEME_DOUBLE, "single": E.SCHEME_SINGLE}
    check(lambda it, i, s, J, b, pt: eng.encrypt(it, i, sch[s], J, pt), lambda cts, b: eng.aggregate_elem(cts),
          lambda it, a, m, J, b, ct: eng.decrypt(it, a, m, J, ct), E.telescope, b, n, J, it, uploaded, scheme, seed)


@pytest.mark.gpu
@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(st.integers(1, 128), st.integers(1, 3000), st.integers(1, 16), st.integers(0, 2 ** 32 - 2), st.integers(0, 2 ** 31))
def test_engine_equals_oracle_random_shapes(oracle, b, n, J, idx, seed):
    from flashe_amd import engine as E
    rng = np.random.Generator(np.random.PCG64(seed))
    eng = E.Engine(KEY, b)
    pt = limbs(rng, n, b)
    assert np.array_equal(eng.encrypt(seed, idx, E.SCHEME_DOUBLE, J, pt), oracle.encrypt(KEY, seed, idx, "double", J, b, pt))
    x = limbs(rng, n, b)
    p = eng.pack(x)
    assert np.array_equal(p, oracle.pack(x, b)) and np.array_equal(eng.unpack(p, n), x)
