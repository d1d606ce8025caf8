"""RS(d,p) GF(2^8) HIP kernels vs the CPU oracle and the committed golden
vectors, bit-exact, through the C-ABI (summerset_amd.RSCodewordBatch)."""
import itertools
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "rs_golden.npz")


def _batch(cuda, data2d, d, p):
    import torch
    from summerset_amd import RSCodewordBatch
    return RSCodewordBatch.from_data(torch.from_numpy(np.ascontiguousarray(data2d)).to(cuda), d, p)


def _parity(cw):
    sl = cw.shard_len
    return cw.buf[:, cw.d * sl:(cw.d + cw.p) * sl].cpu().numpy().reshape(cw.n, cw.p, sl)


@pytest.mark.parametrize("lut", [False, True])
def test_golden_vectors(cuda, oracle, lut):
    g = np.load(GOLD)
    for n in [k[:-5] for k in g.files if k.endswith("_data")]:
        d, p = (3, 2)
        if n.startswith("s"):
            d, p = int(n[1:].split("_")[0]), int(n[1:].split("_")[1])
        cw = _batch(cuda, g[n + "_data"][None, :], d, p)
        cw.compute_parity(lut=lut)
        assert np.array_equal(_parity(cw)[0], g[n + "_parity"]), (n, lut)


@pytest.mark.parametrize("L", [1, 2, 3, 5, 16, 31, 48, 97, 1000, 4099, 4113, 65536 + 7])
def test_encode_matches_oracle_ragged(cuda, oracle, L):
    rng = np.random.default_rng(L)
    n = 37
    data = rng.integers(0, 256, (n, L), dtype=np.uint8)
    cw = _batch(cuda, data, 3, 2)
    cw.compute_parity()
    par = _parity(cw)
    for i in range(n):
        assert np.array_equal(par[i], oracle.rs_encode(3, 2, data[i])), (L, i)
    assert cw.verify_parity().all()
    cw2 = _batch(cuda, data, 3, 2)
    cw2.compute_parity(lut=True)
    assert np.array_equal(_parity(cw2), par)


@pytest.mark.parametrize("scheme", [(3, 2), (6, 4), (9, 6), (12, 8), (5, 5), (4, 1), (1, 1)])
def test_other_schemes(cuda, oracle, scheme):
    d, p = scheme
    rng = np.random.default_rng(d * 31 + p)
    data = rng.integers(0, 256, (9, 777), dtype=np.uint8)
    cw = _batch(cuda, data, d, p)
    cw.compute_parity()
    par = _parity(cw)
    for i in range(9):
        assert np.array_equal(par[i], oracle.rs_encode(d, p, data[i]))


def test_all_erasure_patterns_rs32(cuda, oracle):
    """rscoding.rs:815-861: erase <= 2 of 5 shards -> reconstruct_{all,data} -> data identical."""
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, (16, 4099), dtype=np.uint8)
    ref = _batch(cuda, data, 3, 2)
    ref.compute_parity()
    full = ref.buf.clone()
    for k in (1, 2):
        for lost in itertools.combinations(range(5), k):
            for data_only in (False, True):
                cw = _batch(cuda, data, 3, 2)
                cw.compute_parity()
                cw.erase(lost)
                if data_only:
                    cw.reconstruct_data()
                    assert (cw.get_data() == ref.get_data()).all(), lost
                else:
                    cw.reconstruct_all()
                    assert (cw.buf == full).all(), lost
                    assert cw.verify_parity().all()
                # and the oracle agrees on the rebuilt bytes
                sl = cw.shard_len
                sh = full[0, :5 * sl].cpu().numpy().reshape(5, sl).copy()
                pres = np.ones(5, bool)
                for i in lost:
                    sh[i] = 0xEE; pres[i] = False
                out, _ = oracle.rs_reconstruct(3, 2, sh, pres, data_only=data_only)
                got = cw.buf[0, :5 * sl].cpu().numpy().reshape(5, sl)
                keep = [i for i in range(5) if not (data_only and i >= 3 and i in lost)]
                assert np.array_equal(got[keep], out[keep])


def test_error_cases_mirror_reference(cuda):
    """rscoding.rs:697-876 error behaviour."""
    import torch
    from summerset_amd import RSCodewordBatch, SummersetError
    data = torch.zeros((2, 17), dtype=torch.uint8, device=cuda)
    with pytest.raises(SummersetError):
        RSCodewordBatch.from_data(data, 0, 0)                       # num_data_shards is zero
    null = RSCodewordBatch.from_null(2, 3, 2, device=cuda)
    for fn in (null.compute_parity, null.reconstruct_all, null.reconstruct_data, null.verify_parity, null.get_data):
        with pytest.raises(SummersetError):
            fn()                                                    # codeword is null
    cw = RSCodewordBatch.from_data(data, 3, 2)
    with pytest.raises(SummersetError):
        cw.compute_parity(rs=None)                                  # ReedSolomon coder is None
    cw.erase([1])
    with pytest.raises(SummersetError):
        cw.compute_parity()                                         # not all data shards present
    with pytest.raises(SummersetError):
        cw.verify_parity()
    with pytest.raises(SummersetError):
        cw.reconstruct_all()                                        # only 2 of 5 shards
    cw0 = RSCodewordBatch.from_data(data, 3, 0)
    cw0.compute_parity(rs=None)                                     # p == 0: no coder needed
    assert cw0.avail_parity_shards() == 0 and cw0.verify_parity(rs=None).all()
    cw0.erase([1])
    with pytest.raises(SummersetError):
        cw0.reconstruct_all(rs=None)


def test_verify_detects_corruption(cuda):
    import torch
    from summerset_amd import RSCodewordBatch
    data = torch.randint(0, 256, (64, 4099), dtype=torch.uint8, device=cuda)
    cw = RSCodewordBatch.from_data(data, 3, 2)
    cw.compute_parity()
    cw.buf[5, 2] ^= 1          # data byte
    cw.buf[9, 3 * cw.shard_len + 100] ^= 0x80   # parity byte
    ok = cw.verify_parity().cpu().numpy()
    assert (~ok).nonzero()[0].tolist() == [5, 9]


def test_full_size_properties_config4(cuda):
    """BASELINE config 4 size: 16384 codewords x 4 KiB values (L = 4099).  Checked through
    size-independent properties: linearity over GF(2) and encode -> erase -> decode round trip."""
    import torch
    from summerset_amd import RSCodewordBatch
    n, L = 16384, 4099
    g = torch.Generator(device=cuda).manual_seed(7)
    a = torch.randint(0, 256, (n, L), dtype=torch.uint8, device=cuda, generator=g)
    b = torch.randint(0, 256, (n, L), dtype=torch.uint8, device=cuda, generator=g)
    ca, cb, cx = (RSCodewordBatch.from_data(t, 3, 2) for t in (a, b, a ^ b))
    for c in (ca, cb, cx):
        c.compute_parity()
    assert ((ca.buf ^ cb.buf) == cx.buf).all()                      # parity(a ^ b) = parity(a) ^ parity(b)
    assert cx.verify_parity().all()
    keep = cx.buf.clone()
    cx.erase([0, 4])
    cx.reconstruct_all()
    assert (cx.buf == keep).all()
    cx.erase([1, 2])
    cx.reconstruct_data()
    assert (cx.get_data() == (a ^ b)).all()


def test_padding_bytes_are_never_read(cuda, oracle):
    """The serialized bytes are NOT padded by the caller: poison everything past data_len."""
    import torch
    from summerset_amd import _lib, rscoding
    from summerset_amd._lib import check
    L, n, stride = 4099, 8, 4112
    rng = np.random.default_rng(2)
    raw = rng.integers(0, 256, (n, stride), dtype=np.uint8)
    dev = torch.from_numpy(raw).to(cuda)
    sl = 1367
    par = torch.zeros((n, 2, 1376), dtype=torch.uint8, device=cuda)
    check(_lib.load().smr_rs_encode(dev.data_ptr(), L, stride, n, 3, 2, par.data_ptr(), 2 * 1376, 1376,
                                    _lib.stream_ptr(None)))
    got = par.cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i, :, :sl], oracle.rs_encode(3, 2, raw[i, :L]))
    assert (got[:, :, sl:] == 0).all()                              # nothing written past shard_len


def test_subset_copy_and_absorb_other_rspaxos_flow(cuda, oracle):
    """what RSPaxos does with a request batch (rspaxos/request.rs:127-142, messages.rs:227-259): the leader
    encodes, hands every follower a one-shard subset, and a replica that collected any d shards rebuilds
    the data (rscoding.rs:253-346, 697-876 for the error cases)"""
    import torch
    from summerset_amd import RSCodewordBatch
    from summerset_amd._lib import SummersetError
    rng = np.random.default_rng(11)
    data = rng.integers(0, 256, (33, 1000), dtype=np.uint8)
    cw = RSCodewordBatch.from_data(torch.from_numpy(data).to(cuda), 3, 2)
    cw.compute_parity()
    parts = [cw.subset_copy(1 << k) for k in range(5)]
    assert [p.avail_shards() for p in parts] == [1] * 5 and parts[3].avail_parity_shards() == 1
    for got in ([4, 1, 3], [0, 1, 2], [2, 3, 4]):
        mine = RSCodewordBatch.from_null(33, 3, 2, device=cuda)
        for k in got:
            mine.absorb_other(cw.subset_copy(1 << k))
        assert mine.avail_shards() == 3 and mine.data_len == 1000 and mine.shard_len == cw.shard_len
        mine.reconstruct_data()
        assert np.array_equal(mine.get_data().cpu().numpy(), data), got
    two = RSCodewordBatch.from_null(33, 3, 2, device=cuda)
    two.absorb_other(parts[0]); two.absorb_other(parts[4])
    assert parts[0].avail_shards() == 0                       # moved out
    with pytest.raises(SummersetError):
        two.reconstruct_data()                                # too few shards
    with pytest.raises(SummersetError):
        RSCodewordBatch.from_null(33, 3, 2, device=cuda).subset_copy(1)    # "codeword is null"
    with pytest.raises(SummersetError):
        cw.subset_copy(1 << 5)                                # shard index out-of-bound
    with pytest.raises(SummersetError):
        two.absorb_other(RSCodewordBatch.from_data(torch.from_numpy(data[:, :999].copy()).to(cuda), 3, 2))   # data_len mismatch
    with pytest.raises(SummersetError):
        two.absorb_other(RSCodewordBatch.from_data(torch.from_numpy(data).to(cuda), 4, 1))   # scheme mismatch
    again = cw.subset_copy(0b00111)
    again.absorb_other(cw.subset_copy(0b11100))               # overlapping shard 2: kept, not overwritten
    assert again.avail_shards() == 5 and again.verify_parity().all()


@pytest.mark.parametrize("scheme,L", [((3, 2), 1), ((3, 2), 2), ((3, 2), 47), ((3, 2), 4099), ((3, 2), 4113), ((6, 4), 777), ((1, 1), 33),
                                      ((4, 1), 4096), ((12, 8), 1000), ((3, 0), 100)])
def test_from_data_and_encode_one_pass(cuda, oracle, scheme, L):
    """`smr_rs_from_data_encode` -- from_data's pad + split fused into the encode's loads -- against the oracle (parity), the
    source bytes (data shards, zero padding) and the two-step path (from_data, compute_parity) byte for byte; source rows
    packed (stride = L) and strided; an existing batch refilled"""
    import torch
    from summerset_amd import RSCodewordBatch, SummersetError
    d, p = scheme
    rng = np.random.default_rng(L * 7 + d)
    n = 41
    data = rng.integers(0, 256, (n, L), dtype=np.uint8)
    src = torch.from_numpy(data).to(cuda)
    cw = RSCodewordBatch.from_data_and_encode(src, d, p)
    sl = cw.shard_len
    got = cw.buf[:, :(d + p) * sl].cpu().numpy()
    padded = np.zeros((n, d * sl), np.uint8)
    padded[:, :L] = data
    assert np.array_equal(got[:, :d * sl], padded)                           # rscoding.rs:188-200
    for i in range(n):
        if p:
            assert np.array_equal(got[i, d * sl:].reshape(p, sl), oracle.rs_encode(d, p, data[i])), (scheme, L, i)
    two = RSCodewordBatch.from_data(src, d, p)
    two.compute_parity()
    assert np.array_equal(two.buf[:, :(d + p) * sl].cpu().numpy(), got)
    assert cw.avail_shards() == d + p and np.array_equal(cw.get_data().cpu().numpy(), data)
    if p:
        assert cw.verify_parity().all()
    wide = torch.zeros((n, L + 13), dtype=torch.uint8, device=cuda)          # rows with a stride of their own
    wide[:, :L] = src
    cw2 = RSCodewordBatch.from_data_and_encode(wide[:, :L], d, p, out=cw)
    assert cw2 is cw and np.array_equal(cw.buf[:, :(d + p) * sl].cpu().numpy(), got)
    with pytest.raises(SummersetError):
        RSCodewordBatch.from_data_and_encode(src, d, p, out=RSCodewordBatch(n, L + 1, d, p, device=cuda))
    with pytest.raises(SummersetError):
        RSCodewordBatch.from_data_and_encode(src[:, :0], d, p)               # "codeword is null"


def test_from_data_and_encode_fans_the_shards_out(cuda, oracle):
    """`smr_rs_from_data_encode_fanout`: the same pass also fills one store per shard holder; masks; argument errors"""
    import torch
    from summerset_amd import RSCodewordBatch, SummersetError
    rng = np.random.default_rng(99)
    for (d, p), L, n in (((3, 2), 4113, 53), ((3, 2), 2, 5), ((6, 4), 1000, 17)):
        data = rng.integers(0, 256, (n, L), dtype=np.uint8)
        src = torch.from_numpy(data).to(cuda)
        ref = RSCodewordBatch.from_data(src, d, p)
        ref.compute_parity()
        sl = ref.shard_len
        for mask in (None, 0b10110 & ((1 << (d + p)) - 1), 1 << (d + p - 1)):
            fan = torch.full((d + p, n, sl), 0xAB, dtype=torch.uint8, device=cuda)
            cw = RSCodewordBatch.from_data_and_encode(src, d, p, fan_out=fan, fan_mask=mask)
            assert torch.equal(cw.buf[:, :(d + p) * sl], ref.buf[:, :(d + p) * sl])
            for k in range(d + p):
                if mask is None or (mask >> k) & 1:
                    assert torch.equal(fan[k], ref.shard(k)), (d, p, L, mask, k)
                else:
                    assert bool((fan[k] == 0xAB).all()), (d, p, L, mask, k)          # a store outside the mask is not touched
        with pytest.raises(SummersetError):
            RSCodewordBatch.from_data_and_encode(src, d, p, fan_out=torch.zeros((d + p, n, sl + 1), dtype=torch.uint8, device=cuda))
        with pytest.raises(SummersetError):
            RSCodewordBatch.from_data_and_encode(src, d, p, fan_out=torch.zeros((d + p, n, sl), dtype=torch.uint8, device=cuda), fan_mask=1 << (d + p))


def test_from_data_and_encode_into_shard_stores(cuda, oracle):
    """`smr_rs_from_data_encode_stores` (round 4): every shard written ONCE, shard-major -- store k = shard k of every codeword, the
    codeword a view of the stores.  Against the oracle's parity and the source bytes (zero padding included) byte for byte;
    the shard-major batch through verify / erase / reconstruct (every pattern of RS(3,2)) / get_data / subset_copy; bytes
    outside the stores untouched; argument errors."""
    import itertools
    import torch
    from summerset_amd import RSCodewordBatch, SummersetError
    rng = np.random.default_rng(1234)
    for (d, p), L, n in (((3, 2), 4113, 67), ((3, 2), 2, 5), ((3, 2), 48, 64), ((6, 4), 1000, 17), ((5, 0), 77, 9)):
        data = rng.integers(0, 256, (n, L), dtype=np.uint8)
        src = torch.from_numpy(data).to(cuda)
        sl = -(-L // d)
        guard = torch.full(((d + p) * n * sl + 64,), 0xCD, dtype=torch.uint8, device=cuda)
        stores = guard[32:32 + (d + p) * n * sl].view(d + p, n, sl)
        cw = RSCodewordBatch.from_data_and_encode_stores(src, d, p, stores=stores)
        assert cw.buf is None and cw.shard_len == sl and cw.avail_shards() == d + p
        assert bool((guard[:32] == 0xCD).all()) and bool((guard[32 + (d + p) * n * sl:] == 0xCD).all())
        got = stores.cpu().numpy()
        padded = np.zeros((n, d * sl), np.uint8)
        padded[:, :L] = data
        for k in range(d):
            assert np.array_equal(got[k], padded[:, k * sl:(k + 1) * sl]), (d, p, L, k)          # rscoding.rs:188-200
        if p:
            want = oracle.rs_encode_batch(d, p, data, L, L, n).reshape(n, p, sl)
            for r in range(p):
                assert np.array_equal(got[d + r], want[:, r]), (d, p, L, r)
            assert bool(cw.verify_parity().all())
        assert np.array_equal(cw.get_data().cpu().numpy(), data)
        sub = cw.subset_copy(0b101)
        assert sub.avail == 0b101 and torch.equal(sub.shard(2), stores[2])
        if (d, p) == (3, 2):
            keep = stores.clone()
            for pat in itertools.combinations(range(5), 2):
                cw.erase(pat)
                cw.reconstruct_all()
                assert torch.equal(stores, keep), pat
            cw.erase((0, 1))
            cw.reconstruct_data()
            assert np.array_equal(cw.get_data().cpu().numpy(), data)
        if p:
            with pytest.raises(SummersetError):
                cw.compute_parity()                                                      # encoded when it was made
        with pytest.raises(SummersetError):
            RSCodewordBatch.from_data_and_encode_stores(src, d, p, stores=torch.zeros((d + p, n, sl + 1), dtype=torch.uint8, device=cuda))
        with pytest.raises(SummersetError):
            RSCodewordBatch.from_data_and_encode_stores(src, 0, p)
    made = RSCodewordBatch.from_data_and_encode_stores(torch.from_numpy(data).to(cuda), 5, 0)   # stores made by the call
    assert made.stores.shape == (5, n, -(-L // 5))
