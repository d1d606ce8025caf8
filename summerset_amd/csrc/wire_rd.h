// The lane-side reader of a connection's bytes that the reply-ingest kernels share (csrc/wire_ingest_replies.hip: the [R][G] arrays;
// csrc/raft_engine.hip: raft_wire_replies_kernel, the parse as the prologue of the leader's reply handler): smr_wire's Rd
// (csrc/wire.hip) restated for a lane over a block's LDS copy of its span of the buffer.
#ifndef SMR_WIRE_RD_H
#define SMR_WIRE_RD_H
#include "smr_common.h"

namespace smr {

typedef uint64_t wr_u64_u __attribute__((aligned(1)));
typedef uint32_t wr_u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t WR_STAGE = 32 * 1024;         // bytes of the buffer a block keeps in LDS (+ 16 of slack)
constexpr uint32_t WR_BLOCK = 1024;              // lanes = connections per block
constexpr uint32_t WR_LOC = 512;                 // located frames a block keeps in LDS before it asks for their place

// smr_wire's Rd over the buffer's bytes [n, end) (a frame's payload); `lim` = the buffer's length; the buffer's bytes
// [slo, shi) are in LDS (dword-aligned slo; two more dwords behind shi are there to be read)
struct GlRd {
    const uint8_t *p;
    uint64_t n, end, lim;
    bool ok;
    const uint32_t *lds; uint64_t slo, shi;
    __device__ __forceinline__ uint64_t peek64() const {             // bytes n .. n + 7, little-endian (those past the buffer: zero)
        if (n >= slo && n < shi) {                                   // three aligned dwords, shifted into place
            const uint32_t k = (uint32_t)(n - slo), i = k >> 2, sh = 8 * (k & 3);
            const uint32_t a = lds[i], b = lds[i + 1], c = lds[i + 2];
            const uint32_t lo = (uint32_t)((((uint64_t)b << 32) | a) >> sh), hi = (uint32_t)((((uint64_t)c << 32) | b) >> sh);
            return ((uint64_t)hi << 32) | lo;
        }
        if (n + 8 <= lim) return *(const wr_u64_u *)(p + n);
        uint64_t x = 0;
        for (uint32_t b = 0; b < 8 && n + b < lim; b++) x |= (uint64_t)p[n + b] << (8 * b);
        return x;
    }
    __device__ __forceinline__ uint8_t byte() {
        if (n < end) { const uint8_t b = (uint8_t)peek64(); n++; return b; }
        ok = false;
        return 0;
    }
    __device__ __forceinline__ uint64_t varint() {
        if (n >= end) { ok = false; return 0; }
        const uint64_t x = peek64();
        const uint32_t b = (uint32_t)(x & 0xFF);
        const uint32_t need = b < 251 ? 1 : b == 0xFB ? 3 : b == 0xFC ? 5 : b == 0xFD ? 9 : 0;   // 0xFE (u128), 0xFF: not on this path
        if (need == 0 || n + need > end) { ok = false; n = end; return 0; }
        uint64_t v = b;
        if (need == 3) v = (x >> 8) & 0xFFFF;
        else if (need == 5) v = (x >> 8) & 0xFFFFFFFFull;
        else if (need == 9) { n += 1; v = peek64(); n -= 1; }
        n += need;
        return v;
    }
};


// one AppendEntriesReply frame's payload behind the PeerMessage / PeerMsg tags (raft/mod.rs:203-234, variant 1):
// { term, end_slot, conflict: Option<(Term, usize)> }; false = malformed
__device__ __forceinline__ bool wr_raft_append_reply(GlRd &r, uint64_t &term, uint64_t &end_slot, uint8_t &has, uint64_t &ct, uint64_t &cs) {
    term = r.varint(); end_slot = r.varint();
    has = r.byte();
    ct = 0; cs = 0;
    if (has == 1) { ct = r.varint(); cs = r.varint(); } else if (has != 0) r.ok = false;
    return r.ok && r.n == r.end;
}

}  // namespace smr
#endif
