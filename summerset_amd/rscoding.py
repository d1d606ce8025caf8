"""Batched mirror of the reference's `RSCodeword<T>` (src/utils/rscoding.rs).

One `RSCodewordBatch` holds n codewords of identical geometry in ONE device
buffer `buf[n, cw_stride]`: bytes [0, data_len) of a row are the serialized
value (what `bincode::encode_into_std_write` produced in `from_data`,
rscoding.rs:223-243), shard k is bytes [k*shard_len, (k+1)*shard_len), the d
data shards are followed by the p parity shards.  The zero padding of
`internal_new` (rscoding.rs:188-189) is fused into the encode kernel.  Method
names, argument meaning and error behaviour follow the reference; every
compute call goes to the HIP kernels behind include/summerset_hip.h.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SummersetError, check, stream_ptr


def rs_matrix(d, p):
    """(d+p) x d coding matrix (host numpy) -- ReedSolomon::new(d, p)."""
    m = np.zeros((d + p, d), np.uint8)
    check(_lib.load().smr_rs_matrix(d, p, m.ctypes.data_as(C.c_void_p)))
    return m


def rs_shard_len(data_len, d):
    """shard_len rule of internal_new (rscoding.rs:177-181)."""
    if d == 0:
        raise SummersetError(_lib.SMR_ERR_ARG, "num_data_shards is zero")
    return int(_lib.load().smr_rs_shard_len(data_len, d))


class RSCodewordBatch:
    """n RSCodewords sharing (d, p, data_len); see module docstring."""

    def __init__(self, n, data_len, num_data_shards, num_parity_shards, device="cuda", zero=True):
        import torch
        if num_data_shards == 0:
            raise SummersetError(_lib.SMR_ERR_ARG, "num_data_shards is zero")
        self.n = int(n)
        self.d = int(num_data_shards)
        self.p = int(num_parity_shards)
        self.data_len = int(data_len)
        self.shard_len = rs_shard_len(self.data_len, self.d) if self.data_len else 0
        total = (self.d + self.p) * self.shard_len
        self.cw_stride = (total + 15) // 16 * 16
        alloc = torch.zeros if zero else torch.empty
        self.buf = alloc((self.n, max(self.cw_stride, 16)), dtype=torch.uint8, device=device)
        self.avail = 0  # bitmap of available shard indexes (avail_shards_map)

    # -- constructors ------------------------------------------------------
    @classmethod
    def from_data(cls, data, num_data_shards, num_parity_shards):
        """`data`: uint8 tensor [n, data_len] of serialized bytes (device)."""
        cw = cls(data.shape[0], data.shape[1], num_data_shards, num_parity_shards, device=data.device)
        cw.buf[:, :cw.data_len].copy_(data)
        cw.avail = (1 << cw.d) - 1
        return cw

    @classmethod
    def from_data_and_encode(cls, data, num_data_shards, num_parity_shards, stream=None, out=None, fan_out=None, fan_mask=None, shard_dst=None):
        """`from_data` followed by `compute_parity` (what an RSPaxos / CRaft leader does with every batch,
        rspaxos/request.rs:88-101) as ONE pass over `data` (`smr_rs_from_data_encode`): the serialized bytes are read once
        and the d data shards (zero padding included) and p parity shards are written -- no separate copy into the codeword
        buffer.  `data`: uint8 [n, data_len], rows contiguous.  `out`: a batch of the same geometry to refill (its buffer is
        reused); else a new one.  `fan_out`: uint8 [d + p, n, shard_len] (contiguous) -- store k receives shard k of every
        codeword in the same pass, for the k in `fan_mask` (default: all): the leader's shard fan-out, rspaxos/request.rs:127-142."""
        if data.dim() != 2 or data.stride(1) != 1:
            raise SummersetError(_lib.SMR_ERR_ARG, "data must be [n, data_len] with contiguous rows")
        n, L = int(data.shape[0]), int(data.shape[1])
        if out is None:
            cw = cls(n, L, num_data_shards, num_parity_shards, device=data.device, zero=False)
        else:
            cw = out
            if (cw.n, cw.data_len, cw.d, cw.p) != (n, L, int(num_data_shards), int(num_parity_shards)):
                raise SummersetError(_lib.SMR_ERR_ARG, "`out` has another geometry")
        if L == 0:
            raise SummersetError(_lib.SMR_ERR_ARG, "codeword is null")
        if shard_dst is not None:                  # per-shard destinations: uint8 [n, shard_len] tensors (rows contiguous, one row stride) or None
            if len(shard_dst) != cw.d + cw.p:
                raise SummersetError(_lib.SMR_ERR_ARG, "shard_dst needs d + p entries")
            live = [t for t in shard_dst if t is not None]
            strides = {int(t.stride(0)) for t in live}
            if any(tuple(t.shape) != (n, cw.shard_len) or t.stride(1) != 1 for t in live) or len(strides) > 1:
                raise SummersetError(_lib.SMR_ERR_ARG, "shard_dst entries must be uint8 [n, shard_len] with contiguous rows and one row stride")
            ptrs = (C.c_void_p * (cw.d + cw.p))(*[None if t is None else t.data_ptr() for t in shard_dst])
            check(_lib.load().smr_rs_from_data_encode_scatter(data.data_ptr(), L, int(data.stride(0)), n, cw.d, cw.p, cw.buf.data_ptr(),
                                                              cw.cw_stride, ptrs, strides.pop() if strides else cw.shard_len, stream_ptr(stream)))
        elif fan_out is None:
            check(_lib.load().smr_rs_from_data_encode(data.data_ptr(), L, int(data.stride(0)), n, cw.d, cw.p, cw.buf.data_ptr(),
                                                      cw.cw_stride, stream_ptr(stream)))
        else:
            if tuple(fan_out.shape) != (cw.d + cw.p, n, cw.shard_len) or not fan_out.is_contiguous():
                raise SummersetError(_lib.SMR_ERR_ARG, "fan_out must be a contiguous uint8 [d + p, n, shard_len]")
            mask = (1 << (cw.d + cw.p)) - 1 if fan_mask is None else int(fan_mask)
            check(_lib.load().smr_rs_from_data_encode_fanout(data.data_ptr(), L, int(data.stride(0)), n, cw.d, cw.p, cw.buf.data_ptr(),
                                                             cw.cw_stride, fan_out.data_ptr(), n * cw.shard_len, cw.shard_len, mask,
                                                             stream_ptr(stream)))
        cw.avail = (1 << (cw.d + cw.p)) - 1
        return cw

    @classmethod
    def from_data_and_encode_stores(cls, data, num_data_shards, num_parity_shards, stores=None, stream=None):
        """`from_data` + `compute_parity` with every shard written ONCE, shard-major (`smr_rs_from_data_encode_stores`): shard k of
        codeword i lands in `stores[k, i, :]` (uint8 [d + p, n, shard_len], contiguous; made if None) -- store k is what
        replica k holds of the batch (the payload of its Accept, rspaxos/request.rs:127-142) and the leader's codeword IS
        the d + p stores: the returned batch keeps no buffer of its own (`buf` is None, `shard(k)` = `stores[k]`);
        reconstruct / verify / erase / subset_copy / absorb_other work on it as they are."""
        import torch
        if data.dim() != 2 or data.stride(1) != 1:
            raise SummersetError(_lib.SMR_ERR_ARG, "data must be [n, data_len] with contiguous rows")
        n, L = int(data.shape[0]), int(data.shape[1])
        if num_data_shards == 0:
            raise SummersetError(_lib.SMR_ERR_ARG, "num_data_shards is zero")
        if L == 0:
            raise SummersetError(_lib.SMR_ERR_ARG, "codeword is null")
        cw = cls.__new__(cls)
        cw.n, cw.d, cw.p, cw.data_len = n, int(num_data_shards), int(num_parity_shards), L
        cw.shard_len = rs_shard_len(L, cw.d)
        cw.cw_stride, cw.buf = cw.shard_len, None
        if stores is None:
            stores = torch.empty((cw.d + cw.p, n, cw.shard_len), dtype=torch.uint8, device=data.device)
        if tuple(stores.shape) != (cw.d + cw.p, n, cw.shard_len) or not stores.is_contiguous() or stores.dtype != torch.uint8:
            raise SummersetError(_lib.SMR_ERR_ARG, "stores must be a contiguous uint8 [d + p, n, shard_len]")
        cw.stores = stores
        check(_lib.load().smr_rs_from_data_encode_stores(data.data_ptr(), L, int(data.stride(0)), n, cw.d, cw.p, stores.data_ptr(),
                                                         n * cw.shard_len, cw.shard_len, stream_ptr(stream)))
        cw.avail = (1 << (cw.d + cw.p)) - 1
        return cw

    def _layout(self):
        """(base pointer, byte stride between the shards of a codeword, byte stride between codewords)"""
        if self.buf is None:
            return self.stores.data_ptr(), self.n * self.shard_len, self.shard_len
        return self.buf.data_ptr(), self.shard_len, self.cw_stride

    def _device(self):
        return self.stores.device if self.buf is None else self.buf.device

    @classmethod
    def from_null(cls, n, num_data_shards, num_parity_shards, device="cuda"):
        return cls(n, 0, num_data_shards, num_parity_shards, device=device)

    # -- shard sets (rscoding.rs:253-346): what an RSPaxos leader fans out and a follower collects ----
    def subset_copy(self, subset, copy_data=False):
        """a batch that owns a copy of the shards whose bit is set in `subset` (a bitmap over shard ids);
        shards the source does not hold stay missing.  (`copy_data`: the reference also clones its
        `data_copy`; this mirror keeps no separate data copy, the data shards are the data.)"""
        if self.data_len == 0:
            raise SummersetError(_lib.SMR_ERR_ARG, "codeword is null")
        if subset >> (self.d + self.p):
            raise SummersetError(_lib.SMR_ERR_ARG, "shard index %d out-of-bound" % (subset.bit_length() - 1))
        out = RSCodewordBatch(self.n, self.data_len, self.d, self.p, device=self._device())
        for k in range(self.d + self.p):
            if (subset >> k) & 1 and (self.avail >> k) & 1:
                out.shard(k).copy_(self.shard(k))
                out.avail |= 1 << k
        return out

    def absorb_other(self, other):
        """take the shards `other` holds and I do not (rscoding.rs:296-346); a null batch adopts the geometry"""
        if self.d != other.d:
            raise SummersetError(_lib.SMR_ERR_ARG, "num_data_shards mismatch: expected %d, other %d" % (self.d, other.d))
        if self.p != other.p:
            raise SummersetError(_lib.SMR_ERR_ARG, "num_parity_shards mismatch: expected %d, other %d" % (self.p, other.p))
        if self.data_len != 0 and self.data_len != other.data_len:
            raise SummersetError(_lib.SMR_ERR_ARG, "data_len mismatch: expected %d, other %d" % (self.data_len, other.data_len))
        if self.shard_len != 0 and self.shard_len != other.shard_len:
            raise SummersetError(_lib.SMR_ERR_ARG, "shard_len mismatch: expected %d, other %d" % (self.shard_len, other.shard_len))
        if self.n != other.n:
            raise SummersetError(_lib.SMR_ERR_ARG, "batch size mismatch: expected %d, other %d" % (self.n, other.n))
        if self.data_len == 0:                      # null so far: same data_len / shard_len as the input
            fresh = RSCodewordBatch(self.n, other.data_len, self.d, self.p, device=other._device())
            self.data_len, self.shard_len, self.cw_stride, self.buf = fresh.data_len, fresh.shard_len, fresh.cw_stride, fresh.buf
        for k in range(self.d + self.p):
            if (other.avail >> k) & 1 and not (self.avail >> k) & 1:
                self.shard(k).copy_(other.shard(k))
                self.avail |= 1 << k
        other.avail = 0                             # the reference moves the shards out of `other`

    # -- accessors (rscoding.rs:343-434) -------------------------------------
    def num_data_shards(self):
        return self.d

    def num_parity_shards(self):
        return self.p

    def num_shards(self):
        return self.d + self.p

    def avail_shards_map(self):
        return [bool((self.avail >> k) & 1) for k in range(self.d + self.p)]

    def avail_data_shards(self):
        return bin(self.avail & ((1 << self.d) - 1)).count("1")

    def avail_parity_shards(self):
        return bin(self.avail >> self.d).count("1")

    def avail_shards(self):
        return bin(self.avail).count("1")

    def shard(self, k):
        """view [n, shard_len] of shard k"""
        if self.buf is None:
            return self.stores[k]
        return self.buf[:, k * self.shard_len:(k + 1) * self.shard_len]

    def erase(self, idxs):
        """drop shards (test helper: `cw.shards[i] = None`)"""
        for k in idxs:
            self.avail &= ~(1 << k)
            self.shard(k).fill_(0xEE)

    # -- compute -------------------------------------------------------------
    def compute_parity(self, rs=True, stream=None, lut=False):
        """rscoding.rs:447-486.  `rs` stands for the `Option<&ReedSolomon>` coder."""
        if self.data_len == 0:
            raise SummersetError(_lib.SMR_ERR_ARG, "codeword is null")
        if self.p == 0:
            return
        if not rs:
            raise SummersetError(_lib.SMR_ERR_ARG, "ReedSolomon coder is None")
        if self.avail_data_shards() < self.d:
            raise SummersetError(_lib.SMR_ERR_ARG, "not all data shards present: %d / %d"
                                 % (self.avail_data_shards(), self.d))
        if self.buf is None:
            raise SummersetError(_lib.SMR_ERR_STATE, "a shard-major batch is encoded when it is made (from_data_and_encode_stores)")
        L = _lib.load()
        fn = L.smr_rs_encode_lut if lut else L.smr_rs_encode
        base = self.buf.data_ptr()
        check(fn(base, self.data_len, self.cw_stride, self.n, self.d, self.p,
                 base + self.d * self.shard_len, self.cw_stride, self.shard_len, stream_ptr(stream)))
        self.avail |= ((1 << self.p) - 1) << self.d

    def _reconstruct(self, rs, data_only, stream):
        if self.data_len == 0:
            raise SummersetError(_lib.SMR_ERR_ARG, "codeword is null")
        if self.p == 0:
            if self.avail_data_shards() == self.d:
                return
            raise SummersetError(_lib.SMR_ERR_ARG, "insufficient data shards: %d / %d"
                                 % (self.avail_data_shards(), self.d))
        if not rs:
            raise SummersetError(_lib.SMR_ERR_ARG, "ReedSolomon coder is None")
        base, shard_stride, cw_stride = self._layout()
        check(_lib.load().smr_rs_reconstruct(base, self.shard_len, shard_stride, cw_stride, self.n, self.d, self.p, self.avail,
                                             int(data_only), stream_ptr(stream)))
        self.avail |= (1 << self.d) - 1
        if not data_only:
            self.avail |= ((1 << self.p) - 1) << self.d

    def reconstruct_all(self, rs=True, stream=None):
        self._reconstruct(rs, False, stream)

    def reconstruct_data(self, rs=True, stream=None):
        self._reconstruct(rs, True, stream)

    def verify_parity(self, rs=True, stream=None):
        """rscoding.rs:541-577 -> bool tensor [n]"""
        import torch
        if self.data_len == 0:
            raise SummersetError(_lib.SMR_ERR_ARG, "codeword is null")
        if self.p == 0:
            if self.avail_data_shards() == self.d:
                return torch.ones(self.n, dtype=torch.bool, device=self._device())
            raise SummersetError(_lib.SMR_ERR_ARG, "not all shards present")
        if not rs:
            raise SummersetError(_lib.SMR_ERR_ARG, "ReedSolomon is None")
        if self.avail_shards() < self.d + self.p:
            raise SummersetError(_lib.SMR_ERR_ARG, "not all shards present: %d / %d"
                                 % (self.avail_shards(), self.d + self.p))
        ok = torch.empty(self.n, dtype=torch.uint8, device=self._device())
        base, shard_stride, cw_stride = self._layout()
        check(_lib.load().smr_rs_verify(base, self.shard_len, shard_stride, cw_stride,
                                        self.n, self.d, self.p, ok.data_ptr(), stream_ptr(stream)))
        return ok.bool()

    def get_data(self):
        """rscoding.rs:583-609: the serialized bytes, all data shards required."""
        if self.data_len == 0:
            raise SummersetError(_lib.SMR_ERR_ARG, "codeword is null")
        if self.avail_data_shards() < self.d:
            raise SummersetError(_lib.SMR_ERR_ARG, "not all data shards present: %d / %d"
                                 % (self.avail_data_shards(), self.d))
        if self.buf is None:                       # shard-major: the data shards side by side (a copy)
            import torch
            return torch.cat([self.stores[k] for k in range(self.d)], dim=1)[:, :self.data_len]
        return self.buf[:, :self.data_len]
