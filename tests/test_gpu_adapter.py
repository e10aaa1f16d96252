"""GPU parity tests of the rows either side of the cipher: the quantiser orchestration and its fusion into encrypt / decrypt
(SURVEY.md 8 f-1), and the adapter that drives the cipher (a-16, f-4).  Expected values: fixtures recorded from the unmodified
reference (tests/golden/quantclient.json, block.json, config1.npz) and the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, unhex

pytestmark = pytest.mark.gpu

KEY = bytes(range(32))


@pytest.fixture(scope="module")
def E():
    from flashe_amd import engine
    return engine


class _Weights:
    def __init__(self, layers):
        self.walking_order = sorted(layers)
        self._weights = dict(layers)


def _arr(hexstr, dtype, shape=None):
    a = np.frombuffer(bytes.fromhex(hexstr), dtype=dtype).copy()
    return a if shape is None else a.reshape(shape)


# ------------------------------------------------------------------------------------------------ f-1: orchestration
def test_quantizing_client_against_reference_fixture():
    """QuantizingClient.normalize -> quantize -> (arbiter) -> unquantize -> unnormalize, two rounds, every intermediate equal to
    what the reference produced with the same NumPy draws: bit-exact for everything element-wise, floats compared as bytes; the
    per-layer mean / std are np.mean / np.std of the host array (the reference's own calls: bit-exact against the same calls on this
    machine, 1e-12 against the generating machine's -- NumPy's summation order depends on the CPU)."""
    from flashe_amd.quantize import QuantizingClient
    g = load_golden("quantclient.json")
    assert len(g["clients"]) == 4
    for case in g["clients"]:
        dt = np.dtype(case["dtype"])
        qc = QuantizingClient(case["int_bits"], None, None, case["batch"], case["element_bits"], True, True)
        qc.num_clients = case["num_clients"]
        for rd_i, rd in enumerate(case["rounds"]):
            layers = {k: _arr(v, dt, case["shapes"][k]) for k, v in rd["layers"].items()}
            w = _Weights({k: v.copy() for k, v in layers.items()})
            if rd_i == 0:
                qc.set_layer_size_list(w)
            # every stage is checked from the reference's own state of the previous stage (the statistics carried between
            # rounds agree only to rounding, and one ulp of alpha may move a value across a rounding boundary)
            # (types as the reference leaves them: Python floats before the first round, np.float64 scalars -- np.mean / np.std
            # results -- afterwards; the type decides the dtype NumPy's next `-=` and clip arithmetic run in)
            cast = np.float64 if rd_i else float
            qc.past_layer_mean_list = [cast(float.fromhex(v)) for v in rd["past_mean"]]
            qc.past_layer_std_list = [cast(float.fromhex(v)) for v in rd["past_std"]]
            w = qc.normalize(w)
            for k in w.walking_order:
                assert np.asarray(w._weights[k]).tobytes() == bytes.fromhex(rd["normalized"][k]), (case["int_bits"], rd_i, k, "normalize")
            seed_state = None
            # the reference drew np.random.random(size) layer by layer inside quantize(): feed the same stream
            draws = np.concatenate([_arr(rd["uniforms"][k], np.float64) for k in w.walking_order])
            pos = [0]

            def fake_random(shape):
                size = int(np.prod(shape))
                out = draws[pos[0]:pos[0] + size].reshape(shape)
                pos[0] += size
                return out
            orig = np.random.random
            np.random.random = fake_random
            try:
                w = qc.quantize(w)
            finally:
                np.random.random = orig
            assert [float(a).hex() for a in qc.alpha_list] == rd["alpha"]
            for k in w.walking_order:
                assert [int(v) for v in np.asarray(w._weights[k]).flatten()] == unhex(rd["quantized"][k]), (case["int_bits"], rd_i, k, "quantize")
            shapes = case["shapes"]
            agg = {k: np.array(unhex(rd["aggregate"][k]), dtype=object) for k in w.walking_order}
            if not case["batch"]:
                agg = {k: v.reshape(shapes[k]) for k, v in agg.items()}
            w2 = qc.unquantize(_Weights(agg))
            for k in w2.walking_order:
                assert np.asarray(w2._weights[k], dtype=np.float64).tobytes() == bytes.fromhex(rd["unquantized"][k]), (rd_i, k, "unquantize")
            w2 = qc.unnormalize(w2)
            for k in w2.walking_order:
                assert np.asarray(w2._weights[k], dtype=np.float64).tobytes() == bytes.fromhex(rd["unnormalized"][k]), (rd_i, k, "unnormalize")
            # mean / std: np.mean / np.std of the layer just produced -- the reference's own calls.  NumPy's pairwise summation is
            # unrolled differently per CPU (its SIMD dispatch), so the last bit depends on the HOST: bit-exact against the same calls
            # made here on the fixture's array (what the reference would record on this machine), 1e-12 against the values recorded
            # on the machine that generated the fixture
            for i, k in enumerate(w2.walking_order):
                ref_arr = np.frombuffer(bytes.fromhex(rd["unnormalized"][k]), dtype=np.float64)
                assert float(qc.past_layer_mean_list[i]).hex() == float(np.mean(ref_arr)).hex(), (rd_i, k, "mean")
                assert float(qc.past_layer_std_list[i]).hex() == float(np.std(ref_arr)).hex(), (rd_i, k, "std")
                assert qc.past_layer_mean_list[i] == pytest.approx(float.fromhex(rd["new_mean"][i]), rel=1e-12, abs=1e-15)
                assert qc.past_layer_std_list[i] == pytest.approx(float.fromhex(rd["new_std"][i]), rel=1e-12)
            assert all(isinstance(v, np.floating) for v in qc.past_layer_mean_list + qc.past_layer_std_list)
            del seed_state


def test_mean_std_large_vector_vs_numpy(E):
    """flashe_mean_std_dev at ResNet-50 size, float32 and float64, against NumPy's float64 statistics (tolerance 1e-10 relative)."""
    eng = E.Engine(KEY, 64, device=0)
    for dt in (np.float32, np.float64):
        x = (np.random.RandomState(3).standard_normal(25_557_032) * 0.37 + 0.011).astype(dt)
        d = eng.upload(x)
        mean, std = eng.mean_std_dev(x.size, d, dt == np.float64)
        x64 = x.astype(np.float64)
        assert mean == pytest.approx(float(np.mean(x64)), rel=1e-10, abs=1e-14)
        assert std == pytest.approx(float(np.std(x64)), rel=1e-10)


# ------------------------------------------------------------------------------------------------ f-1: fusion
@pytest.mark.parametrize("b,scheme,n,J,dtype,eb", [(64, "single", 10_000, 8, np.float32, 32), (128, "double", 70_001, 16, np.float32, 16),
                                                    (128, "double", 1_300_003, 16, np.float64, 32), (20, "double", 9_999, 7, np.float32, 16),
                                                    (100, "single", 4_099, 3, np.float64, 20), (64, "double", 250_000, 16, np.float32, 8)])
def test_fused_quantize_encrypt_and_decrypt_unquantize(E, oracle, b, scheme, n, J, dtype, eb):
    """quantise -> encrypt in ONE launch and decrypt -> unquantise in ONE launch, bit-identical to the separate calls and to
    the oracle's quantise + encrypt; prefix lists of every shape on the decrypt side (none, one pair, dropouts, > 96 entries)."""
    from flashe_amd.quantize import ACIQ
    eng = E.Engine(KEY, b, device=0)
    L = 2 if b > 64 else 1
    rng = np.random.RandomState(n)
    x = (rng.standard_normal(n) * 1.3).astype(dtype)
    x[:3] = [0.0, 1e9, -1e9]
    u = rng.random_sample(n)
    alpha = ACIQ(eb).get_alpha_gaus_direct(1.0)
    sch = E.SCHEME_DOUBLE if scheme == "double" else E.SCHEME_SINGLE
    dx, du, dct = eng.upload(x), eng.upload(u), eng.alloc_vec(n)
    eng.quantize_encrypt_dev(5, 3, sch, n, J, dx, dtype == np.float64, alpha, eb, du, dct)
    ct = dct.download(np.uint64, n * L).reshape(n, L)
    q = oracle.quantize(x, alpha, eb, u)
    assert np.array_equal(q, eng.quantize(x, alpha, eb, u))
    assert np.array_equal(ct, oracle.encrypt(KEY, 5, 3, scheme, J, b, q)), "fused quantise + encrypt"
    # decrypt side: an aggregate of C such uploads, every list shape
    C = 4
    agg = np.zeros((n, L), dtype=np.uint64)
    agg[:, 0] = rng.randint(0, 2 ** 31, n).astype(np.uint64) * np.uint64(C)
    dagg, dout = eng.upload(agg), eng.alloc(8 * n)
    for add_idx, minus_idx in [([C], [0]), ([], list(range(C))), ([2, 4], [0, 3]), ([], []), (list(range(200, 330)), list(range(5, 105)))]:
        eng.decrypt_unquantize_dev(5, add_idx, minus_idx, n, J, dagg, alpha, eb, C, dout)
        got = dout.download(np.float64, n)
        dec = oracle.decrypt(KEY, 5, add_idx, minus_idx, J, b, agg)
        want = oracle.unquantize(dec, alpha, eb, C)
        assert got.tobytes() == want.tobytes(), (b, len(add_idx), len(minus_idx))
        assert np.array_equal(dagg.download(np.uint64, n * L).reshape(n, L), agg), "the fused decrypt must not modify its input"


def test_config1_one_launch_per_side(E):
    """BASELINE config 1 end to end with ONE launch on each side of the arbiter: fp32 -> [quantise + encrypt] -> aggregate ->
    [decrypt + unquantise], every stage equal to the fixture recorded from the reference (same MT19937 draws)."""
    z = np.load(os.path.join(GOLDEN, "config1.npz"))
    n, b, J, alpha = 10000, 64, 8, float(z["alpha"])
    eng = E.Engine(KEY, b, device=0)
    dcts = []
    for c in range(2):
        np.random.seed(7 + c)
        u = np.random.random(n)                                   # the draw _static_quantize_padding_asymmetric makes
        dct = eng.alloc_vec(n)
        eng.quantize_encrypt_dev(0, c, E.SCHEME_SINGLE, n, J, eng.upload(z[f"x{c}"]), False, alpha, 32, eng.upload(u), dct)
        assert np.array_equal(dct.download(np.uint64, n), z[f"ct{c}"]), c
        dcts.append(dct)
    dagg, dout = eng.alloc_vec(n), eng.alloc(8 * n)
    eng.aggregate_elem_dev(dcts, n, dagg)
    assert np.array_equal(dagg.download(np.uint64, n), z["agg_elem"])
    eng.decrypt_unquantize_dev(0, [], [0, 1], n, J, dagg, alpha, 32, 2, dout)
    assert dout.download(np.float64, n).tobytes() == z["unq_elem"].tobytes()


# ------------------------------------------------------------------------------------------------ a-16 / f-4: the adapter
def test_flashe_client_on_the_device_against_reference_fixture():
    """FlasheClient (the transport-free _Client mirror) on the HIP engine, replaying the jobs recorded from the reference's own
    forwarders: dense double mask + precompute over two rounds (the second with a dropout), and a sparse job where the arbiter's
    dynamic-masking hint switches the clients to single masks over compact positions."""
    from flashe_amd import cipher as cm
    from flashe_amd.block import FlasheClient, dynamic_masking_choice
    g = load_golden("block.json")
    for case in g["dense_precompute"]:
        b, n, C = case["b"], case["n"], case["num_clients"]
        cm.N_JOBS = case["n_jobs"]
        args = {"quantize": {"int_bits": b, "batch": False, "element_bits": 16, "padding": True, "secure": True},
                "precompute": {"enable": True, "num_params": n}}
        clients = []
        for c in range(C):
            cl = FlasheClient(args)
            cl.create_cipher(c, C, KEY)
            assert "add" in cl.cipher.next_iter_encrypt_prepared
            clients.append(cl)
        for rd in case["rounds"]:
            up = rd["uploaded"]
            for c in up:
                clients[c].set_iter_index(rd["iter"])
                ct = clients[c].encrypt(np.array(unhex(rd["pt"][str(c)]), dtype=object))
                assert [int(v) for v in ct] == unhex(rd["ct"][str(c)]), (b, rd["iter"], c)
            agg = clients[up[0]].cipher.aggregate([np.array(unhex(rd["ct"][str(c)]), dtype=object) for c in up])
            assert [int(v) for v in agg] == unhex(rd["agg"])
            for c in up:
                clients[c].prepare_decrypt()
                clients[c].set_idx_list(list(up))
                dec = clients[c].decrypt(agg)
                assert [int(v) for v in dec] == unhex(rd["dec"][str(c)]), (b, rd["iter"], c, "decrypt")
                clients[c].prepare_encrypt()
    for case in g["sparse_dynamic"]:
        b, total, C, it = case["b"], case["total"], case["num_clients"], case["iter"]
        cm.N_JOBS = case["n_jobs"]
        args = {"quantize": {"int_bits": b, "batch": False, "element_bits": 16, "padding": True, "secure": True},
                "precompute": {"enable": False}, "mask": "dynamic"}
        choice = dynamic_masking_choice(case["masks"], total)
        assert choice == case["choice"]
        uploads = []
        clients = []
        for c in range(C):
            cl = FlasheClient(args)
            cl.create_cipher(c, C, KEY)
            cl.set_iter_index(it)
            cl.dynamic_masking(choice, case["masks"])
            assert cl.cipher.masking_scheme == case["scheme_after_hint"]
            cl.cipher.total = total
            ct = cl.encrypt(np.array(unhex(case["pt"][c]), dtype=object))
            assert [int(v) for v in ct] + [case["zeros"][c]] == unhex(case["uploads"][c]), (b, c)
            uploads.append(ct)
            clients.append(cl)
        agg = np.array(unhex(case["agg"]), dtype=object)
        clients[0].set_idx_list(list(range(C)))
        dec = clients[0].decrypt(agg)
        assert [int(v) for v in dec] == unhex(case["dec"]), b


@pytest.mark.parametrize("seed,pos,n", [(1, None, 1), (2, None, 311), (3, None, 312), (4, None, 313), (5, 0, 1000), (6, 1, 1000), (7, 623, 5), (8, 624, 700),
                                        (9, 17, 100_003), (10, None, 2_000_001),
                                        # substreams (65,536 doubles each, started by jumping ahead): exactly one, one word more, odd positions,
                                        # a stream that ends on a block / substream boundary, many substreams
                                        (11, 0, 65_536), (12, 1, 65_536), (13, 2, 65_536), (14, 0, 65_537), (15, 623, 131_072), (16, 624, 196_608),
                                        (17, 5, 65_536 * 3 - 2), (18, None, 10_000_019), (19, 100, 312 * 1000), (20, 101, 312 * 1000 + 5)])
def test_numpy_random_on_the_device_is_numpy_bit_for_bit(seed, pos, n):
    """flashe_mt19937_random_dev: np.random.random(n) generated on the device from NumPy's own MT19937 state -- the same doubles, and the
    same generator state afterwards, as the host call; odd positions (pairs straddling the 624-word blocks), the position-624 state a
    fresh twist leaves, one draw, exactly one block, a million."""
    from flashe_amd.engine import Engine
    eng = Engine(KEY, 64, device=0)
    np.random.seed(seed)
    if pos is not None:
        st = np.random.get_state()
        np.random.set_state((st[0], st[1], pos, st[3], st[4]))
    st0 = np.random.get_state()
    want = np.random.random(n)
    st_want = np.random.get_state()
    follow_want = np.random.random(7)
    np.random.set_state(st0)
    got = eng.numpy_random_dev(n).download(np.float64, n)
    assert got.tobytes() == want.tobytes()
    st_got = np.random.get_state()
    assert st_got[2] == st_want[2] and np.array_equal(st_got[1], st_want[1])
    assert np.random.random(7).tobytes() == follow_want.tobytes()          # host draws continue the stream


def test_quantize_with_device_draws_equals_host_draws(monkeypatch):
    """_static_quantize_padding_asymmetric on a layer large enough for the device RNG: the same integers as with host draws from the
    same seed, and the global generator left in the same state."""
    from flashe_amd import quantize as qz
    for dt in (np.float32, np.float64):
        x = np.random.Generator(np.random.PCG64(5)).standard_normal(200_003).astype(dt)
        res = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("FLASHE_DEVICE_RNG", flag)
            np.random.seed(99)
            np.random.random(3)                                             # an odd position in the stream
            res[flag] = (qz._static_quantize_padding_asymmetric(x, 2.5, 16, as_object=False), np.random.random(2))
        assert np.array_equal(res["1"][0], res["0"][0]) and res["1"][1].tobytes() == res["0"][1].tobytes(), dt


class _W:
    """What the quantiser and the adapter walk (JZFOrderDictWeights' surface: walking_order, _weights)."""

    def __init__(self, layers):
        self.walking_order = sorted(layers)
        self._weights = dict(layers)


def _client_args(case, **extra):
    d = {"quantize": {"int_bits": case["b"], "batch": bool(case.get("batch")), "element_bits": case["element_bits"], "padding": True, "secure": True},
         "precompute": {"enable": False}}
    d.update(extra)
    return d


def _layers_of(case_layers, rec):
    return {nm: _arr(rec["layers"][nm], np.dtype(dt), sh) for nm, sh, dt in case_layers}


@pytest.mark.parametrize("case_i", range(8))
@pytest.mark.parametrize("mode", ["fused-handles", "fused-host", "call-by-call"])
def test_client_step_is_the_reference_jobs(case_i, mode):
    """The client step of a reference JOB against tests/golden/clientstep.json -- recorded by CALLING the reference:
    QuantizingClient.quantize -> Client.flatten_weights (jzf_aggregator.py:625-650) -> JZFOrderDictWeights.encrypted(_Client), the arbiter's
    two reduces, then decrypted -> Client.unflatten_weights (:652-671) -> unquantize.  The layers are flattened BEFORE the encrypt, so PRF
    counters -- and for int_bits = 64 / 20 / 23 the chunks_idx chunking -- run across the whole model: the b <= 64 cases would fail for a
    step that encrypts layer by layer.  FlasheClient.quantize_encrypt (one launch over the flattened model, per-layer alpha from a device
    table) and the same sequence call by call must both give the fixture's flat ciphertext bit for bit with the fixture's seed, leave
    NumPy's generator where the reference left it, and decrypt_unquantize must return the fixture's floats byte for byte from the
    element-wise AND the packed aggregate.  Cases 5-7 are BATCHED jobs ("batch": true: several quantised values per ciphertext element,
    every layer padded to whole elements on its own before the flatten): quantise + batch of the whole model in one launch, the encrypt
    in a second one; back: decrypt, then unbatch + cut + unquantise in one launch."""
    from flashe_amd import cipher as cm
    from flashe_amd.block import FlasheClient
    from flashe_amd.engine import DeviceVector
    from oracle.flashe_oracle import limbs_to_ints
    case = load_golden("clientstep.json")["dense"][case_i]
    b, C, n = case["b"], case["num_clients"], case["n"]
    cm.N_JOBS = case["n_jobs"]
    clients, cts = [], []
    for c, rec in enumerate(case["clients"]):
        cl = FlasheClient(_client_args(case))
        cl.create_cipher(c, C, KEY)
        cl.cipher.masking_scheme = case["scheme"]
        cl.set_iter_index(case["iter"])
        cl.fuse = mode != "call-by-call"
        np.random.seed(rec["seed"])
        out = cl.quantize_encrypt(_W(_layers_of(case["layers"], rec)), device=(mode == "fused-handles"))
        st = np.random.get_state()
        np.random.seed(rec["seed"])
        np.random.random(sum(int(np.prod(sh)) for _nm, sh, _dt in case["layers"]))          # one draw per VALUE of the model
        st_want = np.random.get_state()
        assert st[2] == st_want[2] and np.array_equal(st[1], st_want[1]), "the NumPy stream must be consumed as the reference consumes it"
        assert out.walking_order == [rec["flat_key"]] and list(out._weights) == [rec["flat_key"]]
        v = out._weights[rec["flat_key"]]
        if mode == "fused-handles":
            assert isinstance(v, DeviceVector)
            got = limbs_to_ints(v.to_host())
        elif mode == "fused-host":
            assert isinstance(v, np.ndarray) and v.dtype == np.uint64
            got = limbs_to_ints(v)
        else:
            got = [int(x) for x in v]
        assert got == unhex(rec["flat_ct"]), (b, c, mode)
        assert [float(a).hex() for a in cl.quantizer.alpha_list] == rec["alpha"]
        assert {k: list(sh) for k, sh in cl.shape_dict.items()} == rec["shape_dict"]
        clients.append(cl)
        cts.append(v)
    # the arbiter's two reduces on what the clients produced
    agg_elem = clients[0].cipher.aggregate(cts)
    agg_packed = clients[0].cipher.aggregate(cts, packed=True)
    as_ints = (lambda a: limbs_to_ints(a.to_host())) if mode == "fused-handles" else (lambda a: limbs_to_ints(a) if a.dtype == np.uint64 else [int(x) for x in a])
    assert as_ints(agg_elem) == unhex(case["agg_elem"]) and as_ints(agg_packed) == unhex(case["agg_packed"])
    for agg, out_name in ((agg_elem, "out_elem"), (agg_packed, "out_packed")):
        clients[0].set_idx_list(list(range(C)))
        back = clients[0].decrypt_unquantize(_W({case["clients"][0]["flat_key"]: agg}))
        assert back.walking_order == sorted(nm for nm, _sh, _dt in case["layers"])
        for nm, sh, _dt in case["layers"]:
            a = back._weights[nm]
            assert a.shape == tuple(sh)
            assert np.asarray(a, dtype=np.float64).tobytes() == bytes.fromhex(case[out_name]["unquantized"][nm]), (b, mode, out_name, nm)
    # object ints in (what an unmodified arbiter would hand back) give the same floats
    clients[0].set_idx_list(list(range(C)))
    back = clients[0].decrypt_unquantize(_W({case["clients"][0]["flat_key"]: np.array(unhex(case["agg_elem"]), dtype=object)}))
    for nm, _sh, _dt in case["layers"]:
        assert np.asarray(back._weights[nm], dtype=np.float64).tobytes() == bytes.fromhex(case["out_elem"]["unquantized"][nm])


@pytest.mark.parametrize("mode", ["fused-handles", "fused-host", "call-by-call"])
@pytest.mark.parametrize("case_i", range(2))
def test_client_step_of_the_sparse_job(case_i, mode):
    """The sparse job's client step against the fixture recorded from the reference (jzf_aggregator.py:717-743, :881-899): compact layers
    plus the 'zzz' layer, quantised, flattened, the trailing quantised zero stripped before and re-appended un-encrypted after the
    (dynamic -> single mask, compact positions) encrypt -- as ONE launch over the compact layers with the 'zzz' value quantised beside it
    (handles or uint64 limbs out) and call by call (object ints out), the NumPy stream left where the reference leaves it; the arbiter's
    expand_to_dense + reduce on those uploads (aggregate_sparse_uploads) against the fixture's dense aggregate; back: the dense aggregate
    decrypted with the sparse minus-mask, unflattened by the DENSE shapes and unquantised."""
    from flashe_amd import cipher as cm
    from flashe_amd.block import FlasheClient, aggregate_sparse_uploads
    from flashe_amd.engine import DeviceVector
    from oracle.flashe_oracle import limbs_to_ints
    case = load_golden("clientstep.json")["sparse"][case_i]
    b, C = case["b"], case["num_clients"]
    cm.N_JOBS = case["n_jobs"]
    cl0, uploads = None, []
    for c, rec in enumerate(case["clients"]):
        cl = FlasheClient(_client_args(case, mask="dynamic"))
        cl.create_cipher(c, C, KEY)
        cl.set_iter_index(case["iter"])
        cl.cipher.total = case["total"]
        cl.dynamic_masking(case["choice"], case["masks"])
        assert cl.cipher.masking_scheme == "single"
        cl.fuse = mode != "call-by-call"
        layers = {nm: _arr(rec["layers"][nm], np.dtype(dt)) for nm, _sh, dt in case["dense_layers"]}
        w = _W(layers)
        cl.quantizer.set_layer_size_list(w)                      # (normalize's first call, before 'zzz' exists)
        w._weights["zzz"] = np.array([0.0])
        w.walking_order = sorted(w._weights, key=str)
        np.random.seed(rec["seed"])
        out = cl.quantize_encrypt(w, device=(mode == "fused-handles"))
        st = np.random.get_state()
        np.random.seed(rec["seed"])
        np.random.random(sum(int(np.asarray(v).size) for v in layers.values()) + 1)          # one draw per value, 'zzz' included
        st_want = np.random.get_state()
        assert st[2] == st_want[2] and np.array_equal(st[1], st_want[1]), "the NumPy stream must be consumed as the reference consumes it"
        k0 = rec["flat_key"]
        assert out.walking_order == [k0]
        v = out._weights[k0]
        if mode == "fused-handles":
            assert isinstance(v, DeviceVector)
            got = limbs_to_ints(v.to_host())
        elif mode == "fused-host":
            assert isinstance(v, np.ndarray) and v.dtype == np.uint64
            got = limbs_to_ints(v)
        else:
            got = [int(x) for x in v]
        assert got == unhex(rec["upload"]), (b, c, mode)
        assert [float(a).hex() for a in cl.quantizer.alpha_list] == rec["alpha"]
        uploads.append(v)
        cl0 = cl0 or cl
    # the arbiter: expand_to_dense of every upload + the reduce, on what the clients produced
    agg = aggregate_sparse_uploads(cl0.cipher.engine, uploads, case["masks"], case["total"], device=(mode == "fused-handles"))
    assert limbs_to_ints(agg.to_host() if mode == "fused-handles" else agg) == unhex(case["agg"]), (b, mode)
    for given in (agg, np.array(unhex(case["agg"]), dtype=object)):
        cl0.set_idx_list(list(range(C)))
        cl0.shape_dict = {nm: tuple(sh) for nm, sh, _dt in case["dense_layers"]}         # shape_dict_used_for_sparsification (:893-894)
        back = cl0.decrypt_unquantize(_W({case["clients"][0]["flat_key"]: given}))
        for nm, sh, _dt in case["dense_layers"]:
            a = np.asarray(back._weights[nm])
            assert a.shape == tuple(sh)
            assert np.array([float(v) for v in a.flatten()], dtype=np.float64).tobytes() == bytes.fromhex(case["unquantized"][nm]), (b, nm, mode)


@pytest.mark.parametrize("b,scheme,n_jobs", [(128, "double", 16), (64, "double", 7), (20, "double", 16), (64, "single", 5)])
def test_client_step_of_a_large_model_vs_oracle(oracle, b, scheme, n_jobs):
    """A model large enough for the device-side draws (90,000 + 1,000 + 350 + 70,001 values, float32 and float64 layers, an empty
    layer): FlasheClient.quantize_encrypt in one launch against `_static_quantize_padding_asymmetric` (pinned by codec.json) layer by
    layer with HOST draws from the same seed -> flatten -> the ORACLE's encrypt of the one flattened vector; the generator must end
    where NumPy's own draws end; and decrypt_unquantize of the aggregate of three such clients against the oracle's decrypt + NumPy's
    unquantise arithmetic."""
    from flashe_amd import cipher as cm
    from flashe_amd import quantize as qz
    from flashe_amd.block import FlasheClient
    cm.N_JOBS = n_jobs
    C, eb, it = 3, 12, 9
    case = {"b": b, "element_bits": eb}
    rng = np.random.Generator(np.random.PCG64(b + n_jobs))
    shapes = {"a_conv": ((300, 300), np.float32), "b_bias": ((1000,), np.float32), "c_dense": ((50, 7), np.float64), "d_empty": ((0,), np.float32),
              "e_fc": ((70001,), np.float32)}
    n = sum(int(np.prod(sh)) for sh, _dt in shapes.values())
    handles, want_cts, clients = [], [], []
    for c in range(C):
        layers = {k: (rng.standard_normal(sh) * 0.7).astype(dt) for k, (sh, dt) in shapes.items()}
        cl = FlasheClient(_client_args(case))
        cl.create_cipher(c, C, KEY)
        cl.cipher.masking_scheme = scheme
        cl.set_iter_index(it)
        np.random.seed(1234 + c)
        np.random.random(5)                                        # an odd position in the stream
        st0 = np.random.get_state()
        handles.append(cl.quantize_encrypt(_W({k: v.copy() for k, v in layers.items()}), device=True))
        st_got = np.random.get_state()
        # expectation: host draws, the fixture-pinned quantiser, flatten, the oracle's encrypt of the ONE vector
        np.random.set_state(st0)
        flat_q = []
        for li, k in enumerate(sorted(layers)):
            alpha = cl.quantizer.alpha_list[li]
            u = np.random.random(layers[k].size)
            x = layers[k].reshape(-1)
            if x.size:
                flat_q.append(qz._static_quantize_padding_asymmetric(x, float(alpha), eb, uniforms=u, as_object=False).astype(np.uint64))
        st_want = np.random.get_state()
        assert st_got[2] == st_want[2] and np.array_equal(st_got[1], st_want[1])
        flat_q = np.concatenate(flat_q)
        assert flat_q.size == n
        want = oracle.encrypt(KEY, it, c, scheme, n_jobs, b, flat_q)
        got = handles[-1]._weights["a_conv"].to_host()
        assert np.array_equal(got.reshape(want.shape), want), (b, scheme, c)
        want_cts.append(want)
        clients.append(cl)
    agg = clients[0].cipher.aggregate([h._weights["a_conv"] for h in handles])
    clients[0].set_idx_list(list(range(C)))
    back = clients[0].decrypt_unquantize(_W({"a_conv": agg}))
    agg_want = oracle.aggregate_elem(want_cts, b)
    add_idx, minus_idx = ([C], [0]) if scheme == "double" else ([], list(range(C)))
    dec = oracle.limbs_to_ints(oracle.decrypt(KEY, it, add_idx, minus_idx, n_jobs, b, agg_want))
    at = 0
    for li, k in enumerate(sorted(shapes)):
        sh, _dt = shapes[k]
        size = int(np.prod(sh))
        alpha = clients[0].quantizer.alpha_list[li] * C
        v = np.array(dec[at:at + size], dtype=np.float64) if size else np.zeros(0)
        want = v * (2 * alpha) / (((1 << eb) - 1) * C) - alpha                      # jzf_quantize.py:102-107
        assert back._weights[k].shape == sh and np.asarray(back._weights[k]).tobytes() == want.reshape(sh).tobytes(), (b, k)
        at += size


def test_numpy_random_on_the_device_full_pass_from_an_odd_position():
    """2^28 - 100 draws from stream position 5 -- a FULL pass of the jump tree, whose substream count used to exceed
    the jump table by one.  Same doubles as NumPy (compared as 64-bit patterns), same generator state, and the stream continues."""
    from flashe_amd.engine import Engine
    eng = Engine(KEY, 64, device=0)
    n = (1 << 28) - 100
    np.random.seed(77)
    st = np.random.get_state()
    np.random.set_state((st[0], st[1], 5, st[3], st[4]))
    st0 = np.random.get_state()
    want = np.random.random(n)
    st_want = np.random.get_state()
    follow_want = np.random.random(3)
    np.random.set_state(st0)
    got = eng.numpy_random_dev(n).download(np.float64, n)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    st_got = np.random.get_state()
    assert st_got[2] == st_want[2] and np.array_equal(st_got[1], st_want[1])
    assert np.random.random(3).tobytes() == follow_want.tobytes()


def test_dynamic_masking_on_the_device():
    """Arbiter.dynamic_masking's decision (jzf_flashe_block.py:92-112) from location lists that already live in HBM: the cases recorded
    from the reference (block.json, produced by calling RB.Arbiter.dynamic_masking), random lists against the one-hot formulation the
    reference uses, more clients than one launch's table holds, and a config-5-sized decision (50 clients x 255,570 of 25,557,032
    positions) without any `total`-sized host array, in well under a millisecond of device time."""
    import time
    from flashe_amd.block import dynamic_masking_choice
    from flashe_amd.engine import Engine
    eng = Engine(KEY, 128, device=0)

    def on_device(masks):
        return [(eng.upload(np.asarray(m, dtype=np.uint32)) if len(m) else eng.alloc(16), len(m)) for m in masks]

    for case in load_golden("block.json")["dynamic_masking"]:
        masks = [sorted(set(m)) for m in case["masks"]]
        if [list(m) for m in case["masks"]] != masks:
            continue                                                   # (the device form takes strictly increasing lists)
        assert dynamic_masking_choice(on_device(masks), case["total"], engine=eng) == case["choice"], case
        assert dynamic_masking_choice(case["masks"], case["total"]) == case["choice"]
    rng = np.random.Generator(np.random.PCG64(11))
    for C, total, k in [(3, 1000, 400), (70, 5000, 900), (130, 3000, 2000), (2, 64, 64), (5, 100, 0)]:
        masks = [np.sort(rng.choice(total, k, replace=False)).astype(np.uint32) for _ in range(C)]
        ohs = []
        for m in masks:
            oh = np.zeros(total, dtype=np.uint8)
            oh[m] = 1
            ohs.append(oh)
        canceled = sum(int((ohs[i] & ohs[i + 1]).sum()) for i in range(C - 1))
        single, double = eng.dynamic_masking_cost_dev([d for d, _k in on_device(masks)], [k] * C)
        assert single == 2 * C * k and double == 4 * C * k - 2 * canceled, (C, total, k)
    total, k, C = 25_557_032, 255_570, 50
    masks = [np.sort(rng.choice(total, k, replace=False)).astype(np.uint32) for _ in range(C)]
    dev = on_device(masks)
    eng.dynamic_masking_cost_dev([d for d, _k in dev], [k] * C)           # warm
    t0 = time.perf_counter()
    single, double = eng.dynamic_masking_cost_dev([d for d, _k in dev], [k] * C)
    dt = time.perf_counter() - t0
    canceled = sum(int(np.intersect1d(masks[i], masks[i + 1], assume_unique=True).size) for i in range(C - 1))
    assert single == 2 * C * k and double == 2 * single - 2 * canceled
    assert dynamic_masking_choice(dev, total, engine=eng) == "single"
    print(f"dynamic_masking at config-5 size on the device: {dt * 1e3:.3f} ms (call incl. the host read-back)")
    assert dt < 5e-3, dt


@pytest.mark.parametrize("b,eb,C,n_jobs", [(128, 16, 10, 16), (120, 16, 3, 7), (64, 12, 3, 16)])
def test_batched_client_step_of_a_large_model_vs_oracle(oracle, b, eb, C, n_jobs):
    """The BATCHED client step at a size that takes the device-side draws (90,000 + 1,000 + 350 + 0 + 70,001 values): quantise + batch of
    the whole model in one launch, the encrypt of the flattened batched vector in a second one -- against the fixture-pinned quantiser
    with host draws, the reference's batching arithmetic on Python ints (jzf_quantize.py:162-185: every layer padded to whole elements
    on its own, first value most significant) and the ORACLE's encrypt of the flattened vector; back: decrypt_unquantize of the
    aggregate against the oracle's decrypt, the unbatching (:234-251) and NumPy's unquantise arithmetic."""
    from flashe_amd import cipher as cm
    from flashe_amd import quantize as qz
    from flashe_amd.block import FlasheClient
    cm.N_JOBS = n_jobs
    it = 3
    factor = int(np.ceil(np.log2(C)))
    fb, bs = eb + factor, b // (eb + factor)
    case = {"b": b, "element_bits": eb, "batch": True}
    rng = np.random.Generator(np.random.PCG64(b + eb))
    shapes = {"a_conv": ((300, 300), np.float32), "b_bias": ((1000,), np.float32), "c_dense": ((50, 7), np.float64), "d_empty": ((0,), np.float32),
              "e_fc": ((70001,), np.float32)}
    handles, want_cts, clients = [], [], []
    for c in range(C if C <= 3 else 3):
        layers = {k: (rng.standard_normal(sh) * 0.7).astype(dt) for k, (sh, dt) in shapes.items()}
        cl = FlasheClient(_client_args(case))
        cl.create_cipher(c, C, KEY)
        cl.set_iter_index(it)
        np.random.seed(4321 + c)
        np.random.random(1)
        st0 = np.random.get_state()
        handles.append(cl.quantize_encrypt(_W({k: v.copy() for k, v in layers.items()}), device=True))
        st_got = np.random.get_state()
        np.random.set_state(st0)
        flat = []
        for li, k in enumerate(sorted(layers)):
            x = layers[k].reshape(-1)
            u = np.random.random(x.size)
            if not x.size:
                continue
            qv = qz._static_quantize_padding_asymmetric(x, float(cl.quantizer.alpha_list[li]), eb, uniforms=u, as_object=False).astype(np.uint64)
            pad = (-len(qv)) % bs
            qo = np.concatenate([qv, np.zeros(pad, dtype=np.uint64)]).astype(object).reshape(-1, bs)
            elems = np.zeros(len(qo), dtype=object)
            for t in range(bs):
                elems = elems * (1 << fb) + qo[:, t]                          # temp *= mod; temp += value
            flat.append(elems)
        st_want = np.random.get_state()
        assert st_got[2] == st_want[2] and np.array_equal(st_got[1], st_want[1])
        flat = np.concatenate(flat)
        want = oracle.encrypt(KEY, it, c, "double", n_jobs, b, oracle.ints_to_limbs([int(v) for v in flat], b))
        k0 = handles[-1].walking_order[0]
        assert np.array_equal(handles[-1]._weights[k0].to_host().reshape(want.shape), want), (b, c)
        assert cl.shape_dict["a_conv"] == ((90000 + bs - 1) // bs,) and cl.quantizer.shape_list[0] == (300, 300)
        want_cts.append(want)
        clients.append(cl)
    up = list(range(len(clients)))
    agg = clients[0].cipher.aggregate([h._weights[h.walking_order[0]] for h in handles])
    clients[0].set_idx_list(list(up))
    back = clients[0].decrypt_unquantize(_W({"a_conv": agg}))
    add, minus = cm._engine.telescope(sorted(up))
    dec = oracle.limbs_to_ints(oracle.decrypt(KEY, it, add, minus, n_jobs, b, oracle.aggregate_elem(want_cts, b)))
    at = 0
    for li, k in enumerate(sorted(shapes)):
        sh, _dt = shapes[k]
        size = int(np.prod(sh))
        nb = (size + bs - 1) // bs
        vals = []
        for item in dec[at:at + nb]:
            vals += [(item >> (fb * (bs - 1 - t))) & ((1 << fb) - 1) for t in range(bs)]
        at += nb
        alpha = clients[0].quantizer.alpha_list[li] * C
        v = np.array(vals[:size], dtype=np.float64) if size else np.zeros(0)
        want = v * (2 * alpha) / (((1 << eb) - 1) * C) - alpha
        assert back._weights[k].shape == sh and np.asarray(back._weights[k]).tobytes() == want.reshape(sh).tobytes(), (b, k)
