"""CPU-only: the host logic of flashe_amd.cipher.FlasheCipher (state machine, prefix
selection, precompute caches, int<->limb conversion, sparse bookkeeping), driven with the
oracle-backed engine double from tests/fake_engine.py against the reference's golden vectors.
The same scenarios run on the real HIP engine in test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import load_golden, unhex
from fake_engine import OracleEngine

from flashe_amd import cipher as cm
from cipher_scenarios import (run_precompute_case, run_round_case, run_sparse_dense_double_case,
                              run_sparse_single_case)


@pytest.fixture()
def cipher_cls(oracle, monkeypatch):
    monkeypatch.setattr(cm.FlasheCipher, "_engine_cls", OracleEngine)
    return cm.FlasheCipher


def test_rounds(cipher_cls):
    for c in load_golden("cipher_rounds.json")["cases"]:
        run_round_case(cm, c)


def test_precompute(cipher_cls):
    for c in load_golden("precompute.json")["cases"]:
        run_precompute_case(cm, c)


def test_sparse_single(cipher_cls):
    for c in load_golden("sparse.json")["single"]:
        run_sparse_single_case(cm, c)


def test_sparse_dense_double(cipher_cls):
    for c in load_golden("sparse.json")["dense_double"]:
        run_sparse_dense_double_case(cm, c)


def test_reference_error_convention(cipher_cls):
    c = cm.FlasheCipher(64)
    assert c.encrypt(np.array([1, 2], dtype=object)) is None          # no seed (jzf_flashe.py:503-504)
    assert c.decrypt(np.array([1, 2], dtype=object)) is None
    c.generate_prp_seed(bytes(range(32)))
    c.set_iter_index(0)
    c.idx = 0
    assert c.encrypt([1, 2, 3]) is None                                # not an ndarray (:496-497)
    assert c.decrypt([1, 2, 3]) is None
    with pytest.raises(ZeroDivisionError):
        cm.FlasheCipher(129)                                           # merge_size = 128 // int_bits == 0
    with pytest.raises(OverflowError):
        c.set_iter_index(-1)
    assert c.get_idx_list() == [0]
    assert len(c.get_prp_seed()) == 256                                # the reference's 256-BYTE seed quirk


def test_key_normalisation(cipher_cls):
    for k in load_golden("aes_anchors.json")["key_norm"]:
        c = cm.FlasheCipher(128)
        c.generate_prp_seed(bytes.fromhex(k["seed"]) if "seed" in k else int(k["seed_int"], 16))
        assert c._key.hex() == k["aes_key"] and len(c.get_prp_seed()) == k["prp_seed_len"]


def test_exchanged_keys(cipher_cls):
    c = cm.FlasheCipher(64)
    c.set_self_uuid("me")
    c.set_exchanged_keys({"g": (0, 1, "guest", 0), "me": (2, 5, "host", 1)})
    assert c.idx == 2 and c.get_guest_uuid() == "g"


def test_uint64_fast_path_equals_object_path(cipher_cls):
    key = bytes(range(32))
    cm.N_JOBS = 4
    for b in (128, 64, 20):
        vals = np.arange(1, 40, dtype=np.uint64) * np.uint64(977)
        c = cm.FlasheCipher(b)
        c.generate_prp_seed(key)
        c.set_iter_index(3)
        c.idx = 1
        a = c.encrypt(vals.astype(object))
        f = c.encrypt(vals)
        ints = [int(v) for v in a]
        if b > 64:
            assert f.shape == (39, 2)
            assert [int(lo) | (int(hi) << 64) for lo, hi in f] == ints
        else:
            assert f.shape == (39,) and [int(v) for v in f] == ints


@pytest.mark.parametrize("b", [64, 128, 33])
def test_uint32_arrays_at_wide_moduli_keep_full_width(cipher_cls, b):
    """np.uint32 plaintexts are a fast path only where a uint32 holds a whole element (int_bits <= 32).  At wider
    moduli they are integer arrays like any other: the ciphertext keeps all int_bits bits (it used to come back truncated to its low
    word) and decrypts back, on host arrays and with device=True / handles."""
    key = bytes(range(32))
    cm.N_JOBS = 4
    vals = (np.arange(1, 40, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(2 ** 32)).astype(np.uint32)
    cl = []
    for i in range(2):
        c = cm.FlasheCipher(b)
        c.generate_prp_seed(key)
        c.set_iter_index(3)
        c.idx = i
        cl.append(c)
    want = [cl[i].encrypt(vals.astype(object)) for i in range(2)]
    assert max(int(v) for v in want[0]) >= 2 ** 32                    # the ciphertext really needs more than a uint32
    got = [cl[i].encrypt(vals) for i in range(2)]
    for g, w in zip(got, want):
        assert g.dtype == object and [int(v) for v in g] == [int(v) for v in w]
    # handles: same values, full width, and the decrypt of the aggregate returns the sum
    hd = [cl[i].encrypt(vals, device=True) for i in range(2)]
    assert hd[0].limbs == cl[0]._engine.limbs
    agg = cl[0].aggregate(hd)
    cl[0].set_idx_list(raw_idx_list=[0, 1], mode="decrypt")
    out = cl[0].decrypt(agg, device=False)
    assert [int(v) for v in np.asarray(out).reshape(len(vals), -1)[:, 0]] == [2 * int(v) for v in vals]
    # a uint32 CIPHERTEXT-typed operand (small values) decrypts at full width too: L limbs reach the engine
    small = np.arange(39, dtype=np.uint32)
    cl[1].set_idx_list(raw_idx_list=[0, 1], mode="decrypt")
    d32 = cl[1].decrypt(small)
    cl[1].set_idx_list(raw_idx_list=[0, 1], mode="decrypt")
    dobj = cl[1].decrypt(small.astype(object))
    assert d32.dtype == object and [int(v) for v in d32] == [int(v) for v in dobj]


def test_dropped_prepared_handles_discard_the_ctx_cache(cipher_cls):
    """the dict entries of next_iter_*_prepared are handles of masks the ctx holds.  A caller that resets the dicts
    (plain attributes in the reference) must not leave a valid cache behind in the ctx."""
    cm.N_JOBS = 4
    c = cm.FlasheCipher(64)
    c.generate_prp_seed(bytes(range(32)))
    c.set_num_clients(2)
    c.set_num_params(30)
    c.set_iter_index(0)
    c.idx = 0
    c.prepare_encrypt()
    c.prepare_decrypt()
    eng = c._engine
    assert eng.prepared_download(1, "add") is not None and eng.prepared_download(2, "add") is not None
    c.next_iter_encrypt_prepared = {}                               # the caller drops the encrypt masks ...
    vals = np.arange(30, dtype=np.uint64)
    c.set_iter_index(1)
    ct = c.encrypt(vals)                                            # ... so this is an ONLINE encrypt of iter 1,
    from oracle import flashe_oracle as orc
    assert np.array_equal(ct, orc.encrypt(bytes(range(32)), 1, 0, "double", 4, 64, vals)[:, 0])
    assert eng.prepared_download(1, "add") is None                  # and the ctx cache is gone with the handle
    assert eng.prepared_download(2, "add") is not None              # the decrypt cache was not touched
    del c.next_iter_decrypt_prepared['add'], c.next_iter_decrypt_prepared['minus']
    c.next_iter_decrypt_prepared_idx = {}
    c.set_idx_list(raw_idx_list=[0], mode="decrypt")
    c.decrypt(ct)
    assert eng.prepared_download(2, "add") is None


def test_dynamic_masking_choice_matches_reference_model():
    """jzf_flashe_block.py:92-112 restated literally (object one-hots, Python sum) vs the mirror."""
    from flashe_amd.block import dynamic_masking_choice
    rng = np.random.RandomState(4)
    for total, C, k in [(100, 3, 10), (50, 4, 40), (64, 2, 64), (30, 5, 1), (200, 6, 150)]:
        masks = [sorted(rng.choice(total, size=k, replace=False).tolist()) for _ in range(C)]
        single_cost = 2 * sum([len(m) for m in masks])
        double_cost = 2 * single_cost
        one_hots = []
        for i in range(C):
            oh = np.zeros(total, dtype=object)
            oh[masks[i]] = 1
            one_hots.append(oh)
        canceled = 0
        for i in range(C - 1):
            canceled += sum(one_hots[i] & one_hots[i + 1])
        double_cost -= canceled * 2
        want = "single" if single_cost <= double_cost else "double"
        assert dynamic_masking_choice(masks, total) == want
    # the survey's observation: single_cost <= double_cost always holds
    assert dynamic_masking_choice([[0, 1, 2], [0, 1, 2]], 3) == "single"


def test_flashe_client_adapter_round(cipher_cls):
    """Two clients + precompute through the _Client-style adapter; the aggregate decrypts to the sum."""
    from flashe_amd.block import FlasheClient
    cm.N_JOBS = 4
    args = {"quantize": {"int_bits": 64, "batch": False, "element_bits": 32, "padding": True, "secure": True},
            "precompute": {"enable": True, "num_params": 50}}
    clients = []
    for i in range(2):
        c = FlasheClient(args)
        c.create_cipher(i, 2, bytes(range(32)))
        assert 'add' in c.cipher.next_iter_encrypt_prepared          # iteration-0 masks precomputed (iter_index starts at -1)
        c.set_iter_index(0)
        clients.append(c)
    vals = np.arange(50, dtype=np.uint64) * np.uint64(3)
    cts = [c.encrypt(vals.astype(object)) for c in clients]
    assert clients[0].cipher.next_iter_encrypt_prepared == {}
    agg = clients[0].cipher.aggregate(cts)
    clients[0].prepare_decrypt()
    idx = clients[0].get_idx_list() + clients[1].get_idx_list()
    clients[0].set_idx_list(idx)
    out = clients[0].decrypt(agg)
    assert [int(v) for v in out] == [int(2 * v) for v in vals]


def test_object_array_conversion_fast_and_fallback_paths():
    """_to_limbs: the one-pass conversion (values in [0, 2**64)) and the masked fallback (negative or wider Python ints)
    give the limbs the reference's `& (2**b - 1)` arithmetic implies."""
    import numpy as np
    from flashe_amd.cipher import _from_limbs, _to_limbs
    vals = [0, 1, 2 ** 64 - 1, 12345678901234567890]
    for limbs in (1, 2):
        got, kind = _to_limbs(np.array(vals, dtype=object), limbs)
        assert kind == "object" and [int(v) for v in got[:, 0]] == vals and (limbs == 1 or not got[:, 1].any())
        wide = vals + [-3, 2 ** 70 + 5, 2 ** 128 - 1]
        got, _ = _to_limbs(np.array(wide, dtype=object), limbs)
        assert [int(v) for v in got[:, 0]] == [v & (2 ** 64 - 1) for v in wide]
        if limbs == 2:
            assert [int(v) for v in got[:, 1]] == [(v >> 64) & (2 ** 64 - 1) for v in wide]
            back = _from_limbs(got, "object")
            assert [int(v) for v in back] == [v & (2 ** 128 - 1) for v in wide]


def test_object_array_conversion_handles_negative_and_numpy_scalars():
    """_to_limbs must reduce values mod 2**(64 * limbs) exactly as Python's `&` does in the reference
    ((value + add - minus) & mask, jzf_flashe.py:480-481): negative Python ints, negative NumPy scalars held in an
    object array (astype(uint64) wraps those silently) and values at the int64 / uint64 edges."""
    import numpy as np
    from flashe_amd.cipher import _from_limbs, _to_limbs
    M = (1 << 64) - 1
    cases = [[1, 2, 3], [np.int64(-3), 5], [-3, 5], [2 ** 63, 1], [2 ** 64 - 1, 0], [2 ** 64, 1], [2 ** 100, -2 ** 90],
             [np.int64(-1), 2 ** 63 + 5], [np.uint64(2 ** 64 - 1), 3], []]
    for vals in cases:
        for limbs in (1, 2):
            got, kind = _to_limbs(np.array(vals, dtype=object), limbs)
            assert kind == "object" and got.shape == (len(vals), limbs)
            for i, v in enumerate(vals):
                want = int(v) & ((1 << (64 * limbs)) - 1)
                assert int(got[i, 0]) == want & M and (limbs == 1 or int(got[i, 1]) == want >> 64), (vals, limbs, i)
            back = _from_limbs(got, kind)
            assert [int(x) for x in back] == [int(v) & ((1 << (64 * limbs)) - 1) for v in vals]


def test_dynamic_masking_and_aciq_against_reference_fixtures():
    """a-16 / f-1 host logic pinned by the reference itself: tests/golden/block.json holds the decisions of the unmodified
    Arbiter.dynamic_masking (jzf_flashe_block.py:89-117) run on stub objects, quantclient.json the ACIQ alphas (jzf_aciq.py:10-27)."""
    from conftest import load_golden
    from flashe_amd.block import dynamic_masking_choice
    from flashe_amd.quantize import ACIQ
    g = load_golden("block.json")
    assert len(g["dynamic_masking"]) >= 7
    for c in g["dynamic_masking"]:
        assert dynamic_masking_choice(c["masks"], c["total"]) == c["choice"], c["total"]
    for c in g["sparse_dynamic"]:
        assert dynamic_masking_choice(c["masks"], c["total"]) == c["choice"] == c["scheme_after_hint"]
    for c in load_golden("quantclient.json")["aciq"]:
        a = ACIQ(c["bits"])
        assert float(a.get_alpha_gaus_direct(float.fromhex(c["sigma"]))).hex() == c["alpha_direct"], c
        assert float(a.get_alpha_gaus(float.fromhex(c["min"]), float.fromhex(c["max"]), c["size"])).hex() == c["alpha_gaus"], c


def test_flashe_client_flows_from_reference_fixture(cipher_cls):
    """f-4: the dense double-mask + precompute job (two rounds, the second with a dropout) recorded from the reference's own
    _Client forwarders, replayed through FlasheClient on the engine double: same ciphertexts, same decrypts (including the
    rounds where the reference's unconditional precomputed masks make the result differ from the plaintext sum)."""
    from conftest import load_golden, unhex
    from flashe_amd.block import FlasheClient
    for case in load_golden("block.json")["dense_precompute"]:
        b, n, C = case["b"], case["n"], case["num_clients"]
        cm.N_JOBS = case["n_jobs"]
        args = {"quantize": {"int_bits": b, "batch": False, "element_bits": 16, "padding": True, "secure": True},
                "precompute": {"enable": True, "num_params": n}}
        clients = []
        for c in range(C):
            cl = FlasheClient(args)
            cl.create_cipher(c, C, bytes(range(32)))
            clients.append(cl)
        for rd in case["rounds"]:
            up = rd["uploaded"]
            for c in up:
                clients[c].set_iter_index(rd["iter"])
                ct = clients[c].encrypt(np.array(unhex(rd["pt"][str(c)]), dtype=object))
                assert [int(v) for v in ct] == unhex(rd["ct"][str(c)]), (b, rd["iter"], c)
            agg = np.array(unhex(rd["agg"]), dtype=object)
            for c in up:
                clients[c].prepare_decrypt()
                clients[c].set_idx_list(list(up))
                dec = clients[c].decrypt(agg)
                assert [int(v) for v in dec] == unhex(rd["dec"][str(c)]), (b, rd["iter"], c, "decrypt")
                clients[c].prepare_encrypt()


class _W:
    def __init__(self, layers):
        self._weights = dict(layers)
        self.walking_order = sorted(self._weights.keys(), key=str)


def test_client_step_flatten_semantics_from_reference_fixture(cipher_cls):
    """The client step of a reference JOB (tests/golden/clientstep.json, recorded by calling QuantizingClient.quantize ->
    Client.flatten_weights -> JZFOrderDictWeights.encrypted(_Client) and back): the call-by-call path of FlasheClient on the engine
    double, with the quantiser's per-layer output taken from the fixture (the quantiser itself needs the GPU: tests/test_gpu_adapter.py).
    What this pins on the CPU: the layers are flattened BEFORE the encrypt -- one vector under the first key, PRF counters and the
    int_bits <= 64 chunking across the layers -- the sparse job's trailing quantised zero rides un-encrypted, and unflatten cuts by
    shape_dict."""
    from flashe_amd.block import FlasheClient
    g = load_golden("clientstep.json")
    for case in g["dense"]:
        b, C = case["b"], case["num_clients"]
        cm.N_JOBS = case["n_jobs"]
        args = {"quantize": {"int_bits": b, "batch": case["batch"], "element_bits": case["element_bits"], "padding": True, "secure": True},
                "precompute": {"enable": False}}
        for c, rec in enumerate(case["clients"]):
            # what the quantiser leaves per layer: the layer's own shape -- or, batched job, a 1-D array of batched elements (shape_dict
            # records exactly these shapes, jzf_aggregator.py:641-642)
            shapes = [(nm, tuple(rec["shape_dict"][nm])) for nm, _sh, _dt in case["layers"]]
            sizes = [int(np.prod(sh)) for _nm, sh in shapes]
            cl = FlasheClient(args)
            cl.create_cipher(c, C, bytes(range(32)))
            cl.cipher.masking_scheme = case["scheme"]
            cl.set_iter_index(case["iter"])
            flat_q = unhex(rec["flat_quantized"])
            per_layer, at = {}, 0
            for (nm, sh), size in zip(shapes, sizes):
                per_layer[nm] = np.array(flat_q[at:at + size], dtype=object).reshape(sh)
                at += size
            cl.quantizer.layer_size_list = sizes
            cl.quantizer.quantize = lambda w, per_layer=per_layer: _W(per_layer)        # (the GPU quantiser's output, from the fixture)
            out = cl.quantize_encrypt(_W({nm: None for nm in per_layer}), device=False)
            assert out.walking_order == [rec["flat_key"]]
            assert [int(v) for v in out._weights[rec["flat_key"]]] == unhex(rec["flat_ct"]), (b, c)
            assert {k: list(v) for k, v in cl.shape_dict.items()} == rec["shape_dict"]
            if c == 0:
                for agg_name, out_name in (("agg_elem", "out_elem"), ("agg_packed", "out_packed")):
                    cl.set_idx_list(list(range(C)))
                    w = _W({rec["flat_key"]: np.array(unhex(case[agg_name]), dtype=object)})
                    w._weights[rec["flat_key"]] = cl.decrypt(w._weights[rec["flat_key"]])
                    assert [int(v) for v in w._weights[rec["flat_key"]]] == unhex(case[out_name]["dec"]), (b, agg_name)
                    w = cl.unflatten_weights(w)
                    assert w.walking_order == sorted(per_layer) and all(w._weights[nm].shape == sh for nm, sh in shapes)
                    assert [int(v) for nm in w.walking_order for v in w._weights[nm].flatten()] == unhex(case[out_name]["dec"])
    for case in g["sparse"]:
        b, C = case["b"], case["num_clients"]
        cm.N_JOBS = case["n_jobs"]
        args = {"quantize": {"int_bits": b, "batch": False, "element_bits": case["element_bits"], "padding": True, "secure": True},
                "precompute": {"enable": False}, "mask": "dynamic"}
        for c, rec in enumerate(case["clients"]):
            cl = FlasheClient(args)
            cl.create_cipher(c, C, bytes(range(32)))
            cl.set_iter_index(case["iter"])
            cl.cipher.total = case["total"]
            cl.dynamic_masking(case["choice"], case["masks"])
            flat_q = unhex(rec["flat_quantized"])
            per_layer, at = {}, 0
            for (nm, _sh, _dt), k in zip(case["dense_layers"], case["ks"]):
                per_layer[nm] = np.array(flat_q[at:at + k], dtype=object)
                at += k
            per_layer["zzz"] = np.array(flat_q[at:], dtype=object)
            assert len(per_layer["zzz"]) == 1
            cl.quantizer.layer_size_list = list(case["ks"])
            cl.quantizer.quantize = lambda w, per_layer=per_layer: _W(per_layer)
            out = cl.quantize_encrypt(_W({nm: None for nm in per_layer}))
            assert out.walking_order == [rec["flat_key"]]
            assert [int(v) for v in out._weights[rec["flat_key"]]] == unhex(rec["upload"]), (b, c)
            assert int(out._weights[rec["flat_key"]][-1]) == flat_q[-1]                 # the quantised zero is NOT encrypted
            if c == 0:
                cl.set_idx_list(list(range(C)))
                dec = cl.decrypt(np.array(unhex(case["agg"]), dtype=object))
                assert [int(v) for v in dec] == unhex(case["dec"]), b


def test_quantizing_client_refuses_the_branches_it_does_not_mirror():
    """padding=False leaves the reference's own quantize() without a result (its code there is commented out) and secure=False is the
    plain-text path: the mirror says so instead of quantising silently."""
    from flashe_amd.quantize import QuantizingClient
    for kw in ({"padding": False}, {"secure": False}):
        with pytest.raises(NotImplementedError):
            QuantizingClient(64, **kw)
    QuantizingClient(64, padding=True, secure=True)
    QuantizingClient(64, padding=None, secure=None)        # a job file without the keys
