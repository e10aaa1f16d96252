"""Golden-fixture scenarios for a FlasheCipher implementation `cm.FlasheCipher`; shared by the
CPU host-logic tests (oracle-backed engine double) and the GPU parity tests (HIP engine)."""
import numpy as np

from conftest import unhex

KEY = bytes(range(32))


def obj(vals):
    return np.array([int(v) for v in vals], dtype=object)


def new_cipher(cm, b, scheme, idx, it, C):
    c = cm.FlasheCipher(b, mask=scheme)
    c.set_num_clients(C)
    c.generate_prp_seed(KEY)
    c.set_iter_index(it)
    c.idx = idx
    return c


def ints(arr):
    return [int(v) for v in arr]


def run_round_case(cm, c):
    b, n, it, scheme, C = c["b"], c["n"], c["iter"], c["scheme"], c["num_clients"]
    cm.N_JOBS = c["n_jobs"]
    cts = {}
    for i, pt in c["pt"].items():
        ci = new_cipher(cm, b, scheme, int(i), it, C)
        ct = ci.encrypt(obj(unhex(pt)))
        assert ct.dtype == object and ints(ct) == unhex(c["ct"][i]), (scheme, b, n, i)
        cts[int(i)] = ct
    models = [cts[i] for i in c["uploaded"]]
    d = new_cipher(cm, b, scheme, 0, it, C)
    agg = d.aggregate(models)
    assert ints(agg) == unhex(c["agg_elem"])
    aggp = d.aggregate(models, packed=True)
    assert ints(aggp) == unhex(c["agg_packed"])
    raw = list(c["uploaded"])
    d.set_idx_list(raw_idx_list=raw, mode="decrypt")
    if scheme == "double":
        assert [p.hex() for p in d.index_prefix_for_add] == c["prefix_add"]
        assert raw == sorted(c["uploaded"])                    # sorted in place like the reference
    assert [p.hex() for p in d.index_prefix_for_minus] == c["prefix_minus"]
    if n:
        assert ints(d.decrypt(agg)) == unhex(c["dec_elem"])
        d.set_idx_list(raw_idx_list=list(c["uploaded"]), mode="decrypt")
        assert ints(d.decrypt(aggp)) == unhex(c["dec_packed"])
    run_round_case_with_handles(cm, c)


def run_round_case_with_handles(cm, c):
    """The same round with the results kept on the device between the three calls (DeviceVector handles): every client uploads its
    plaintext once, the ciphertexts never visit the host, one download at the end -- and every intermediate, downloaded for the
    check, equals the fixture."""
    from oracle.flashe_oracle import limbs_to_ints
    b, n, it, scheme, C = c["b"], c["n"], c["iter"], c["scheme"], c["num_clients"]
    cm.N_JOBS = c["n_jobs"]
    if n == 0:
        return
    handles = {}
    for i, pt in c["pt"].items():
        ci = new_cipher(cm, b, scheme, int(i), it, C)
        h = ci.encrypt(obj(unhex(pt)), device=True)
        assert isinstance(h, cm.DeviceVector) and len(h) == n
        assert limbs_to_ints(h.to_host()) == unhex(c["ct"][i]), (scheme, b, n, i)
        handles[int(i)] = h
    d = new_cipher(cm, b, scheme, 0, it, C)
    models = [handles[i] for i in c["uploaded"]]
    agg = d.aggregate(models)                                  # DeviceVector in -> DeviceVector out
    assert isinstance(agg, cm.DeviceVector) and limbs_to_ints(agg.to_host()) == unhex(c["agg_elem"])
    aggp = d.aggregate(models, packed=True)
    assert isinstance(aggp, cm.DeviceVector) and limbs_to_ints(aggp.to_host()) == unhex(c["agg_packed"])
    mixed = d.aggregate([models[0].to_host()] + models[1:], device=False)      # host and device operands mixed, host result
    assert isinstance(mixed, np.ndarray) and limbs_to_ints(mixed) == unhex(c["agg_elem"])
    d.set_idx_list(raw_idx_list=list(c["uploaded"]), mode="decrypt")
    dec = d.decrypt(agg)
    assert isinstance(dec, cm.DeviceVector) and limbs_to_ints(dec.to_host()) == unhex(c["dec_elem"])
    d.set_idx_list(raw_idx_list=list(c["uploaded"]), mode="decrypt")
    dec_host = d.decrypt(aggp, device=False)                   # a handle in, a host array out
    assert isinstance(dec_host, np.ndarray) and limbs_to_ints(dec_host) == unhex(c["dec_packed"])
    t = handles[int(c["uploaded"][0])].to_host()
    h2 = new_cipher(cm, b, scheme, int(c["uploaded"][0]), it, C).encrypt(cm.DeviceVector.from_host(d.engine, t.reshape(len(t), -1)[:, :1] * 0 + 5))
    assert isinstance(h2, cm.DeviceVector)                     # a device-resident plaintext gives a device-resident ciphertext
    if b <= 32 and getattr(d, "_compact_ok", lambda: False)():
        # int_bits <= 32: the handles this class produces are uint32 arrays in HBM (the compact layout), whatever the input was
        assert all(h.compact for h in handles.values()) and agg.compact and dec.compact and h2.compact and not aggp.compact
        assert handles[int(c["uploaded"][0])].to_host().dtype == np.uint32


def run_precompute_case(cm, c):
    b, n, it, C = c["b"], c["n"], c["iter"], c["num_clients"]
    cm.N_JOBS = c["n_jobs"]
    for i, cl in c["clients"].items():
        ci = new_cipher(cm, b, "double", int(i), it - 1, C)
        ci.set_num_params(n)
        ci.prepare_encrypt()
        eng = ci.engine
        assert 'add' in ci.next_iter_encrypt_prepared and 'minus' in ci.next_iter_encrypt_prepared
        got_add = ci.next_iter_encrypt_prepared['add'].to_host(eng)
        got_minus = ci.next_iter_encrypt_prepared['minus'].to_host(eng)
        from oracle.flashe_oracle import limbs_to_ints
        assert limbs_to_ints(got_add) == unhex(cl["pre_add"])
        assert limbs_to_ints(got_minus) == unhex(cl["pre_minus"])
        ci.set_iter_index(it)
        ct = ci.encrypt(obj(unhex(c["pt"][i])))
        assert ints(ct) == unhex(cl["ct"])
        assert ci.next_iter_encrypt_prepared == {}             # consumed (jzf_flashe.py:483-486)
    d = new_cipher(cm, b, "double", 0, it, C)
    d.set_num_params(n)
    d.prepare_decrypt()
    assert d.next_iter_decrypt_prepared_idx == {'add': [C], 'minus': [0]}
    d.set_idx_list(raw_idx_list=list(c["uploaded"]), mode="decrypt")
    assert [p.hex() for p in d.index_prefix_for_add] == c["extra_prefix_add"]
    assert [p.hex() for p in d.index_prefix_for_minus] == c["extra_prefix_minus"]
    dec = d.decrypt(obj(unhex(c["agg"])))
    assert ints(dec) == unhex(c["dec"])
    assert d.next_iter_decrypt_prepared == {} and d.next_iter_decrypt_prepared_idx == {}


def run_sparse_single_case(cm, c):
    b, total, it, C = c["b"], c["total"], c["iter"], c["num_clients"]
    cm.N_JOBS = c["n_jobs"]
    for i in range(C):
        ci = new_cipher(cm, b, "single", i, it, C)
        ct = ci.encrypt(obj(unhex(c["pt"][i])))
        assert ints(ct) == unhex(c["uploads"][i])[:-1]
    d = new_cipher(cm, b, "single", 0, it, C)
    d.masks = [list(l) for l in c["locs"]]
    d.total = total
    d.set_idx_list(raw_idx_list=None, mode="decrypt")
    from oracle.flashe_oracle import limbs_to_ints
    assert limbs_to_ints(d.next_iter_decrypt_prepared["minus"].to_host(d.engine)) == unhex(c["minus_mask"])
    dec = d.decrypt(obj(unhex(c["agg"])))
    assert ints(dec) == unhex(c["dec"])
    assert "minus" not in d.next_iter_decrypt_prepared


def run_sparse_dense_double_case(cm, c):
    """The dense-position double-mask masks the class builds from location lists must equal the
    direct _static_prepare_decrypt_spar(0, total, ...) call recorded in the fixture."""
    b, total, it = c["b"], c["total"], c["iter"]
    cm.N_JOBS = 1
    # rebuild location lists whose run analysis reproduces the fixture's selectors
    C = len(c["minus_sel"])
    one_hots = []
    # minus[0] = oh[0]; minus[c] = oh[c] & ~oh[c-1]; add[C] = oh[C-1]; add[c+1] = oh[c] & ~oh[c+1]
    oh = np.array(c["minus_sel"][0], dtype=np.uint8)
    one_hots.append(oh)
    for k in range(1, C):
        # oh[k] = minus[k] | (oh[k-1] & ~add[k])   (positions kept from the previous client's run)
        prev = one_hots[-1]
        cur = np.array(c["minus_sel"][k], dtype=np.uint8) | (prev & (1 - np.array(c["add_sel"][k], dtype=np.uint8)))
        one_hots.append(cur)
    d = new_cipher(cm, b, "double", 0, it, C)
    d.masks = [np.nonzero(o)[0].tolist() for o in one_hots]
    d.total = total
    d.set_idx_list(raw_idx_list=None, mode="decrypt")
    from oracle.flashe_oracle import limbs_to_ints
    assert limbs_to_ints(d.next_iter_decrypt_prepared["add"].to_host(d.engine)) == unhex(c["add"])
    assert limbs_to_ints(d.next_iter_decrypt_prepared["minus"].to_host(d.engine)) == unhex(c["minus"])
