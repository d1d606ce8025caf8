// Device-resident KV state machine over REAL keys and values (VERDICT r1 "missing" #5):
// `StateMachineExecutorTask::execute` (src/server/statemach.rs:193-202) on `State = HashMap<String, String>`
// (:21-63) for G independent groups, lane = group, commands applied row after row in submission order.
//
// Per group: an open-addressing table of `slots` entries (linear probing on a 64-bit FNV-1a hash of the key bytes;
// an entry = hash, key (off, len) and value (off, len) into the group's heap) and an append-only byte heap holding
// the key and value bytes a Put brought.  A Put of an existing key appends the new value and repoints the entry;
// the old bytes stay where they are, so the (off, len) a result names -- Get { value }, Put { old_value } -- remains
// readable for as long as the object lives (no compaction).  Commands name their bytes by (off, len) into a payload
// buffer of the caller (the decoded ReqBatch bytes, already on the device).  Entry fields are SoA [slots][G]: a
// wavefront probing the same slot index touches one contiguous run per field; the heaps are per-group strips.
// A table or heap that runs full sets the group's sticky `full` flag; its later commands answer as no-ops (state 2).
#include <string.h>

#include "smr_common.h"

namespace smr {

constexpr uint32_t SKV_EMPTY = 0xFFFFFFFFu;

struct SkvView {
    uint32_t G, slots, mask;
    uint64_t heap_bytes;
    uint64_t *hash;                 // [slots][G]
    uint32_t *koff, *klen, *voff, *vlen;   // [slots][G]; klen == SKV_EMPTY: free entry
    uint8_t *heap;                  // [G][heap_bytes]
    uint32_t *heap_used;            // [G]
    uint8_t *full;                  // [G]
    uint32_t *n_keys;               // [G]
};

__device__ __forceinline__ uint64_t fnv1a(const uint8_t *p, uint32_t n) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (uint32_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001B3ull; }
    return h;
}
__device__ __forceinline__ bool bytes_eq(const uint8_t *a, const uint8_t *b, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) if (a[i] != b[i]) return false;
    return true;
}

// kind[n_rows][G]: 0 Get, 1 Put, else no command.  res_state: 0 = None, 1 = Some (res_off / res_len into the group's
// heap), 2 = refused (group full).  Get: value; Put: old_value.
__global__ __launch_bounds__(256) void skv_execute_kernel(const SkvView v, uint32_t n_rows, const uint8_t *__restrict__ kind,
                                                          const uint8_t *__restrict__ payload, uint64_t payload_bytes,
                                                          const uint32_t *__restrict__ key_off, const uint32_t *__restrict__ key_len,
                                                          const uint32_t *__restrict__ val_off, const uint32_t *__restrict__ val_len,
                                                          uint8_t *__restrict__ res_state, uint32_t *__restrict__ res_off,
                                                          uint32_t *__restrict__ res_len) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    uint8_t *const heap = v.heap + (size_t)g * v.heap_bytes;
    uint32_t used = v.heap_used[g], nk = v.n_keys[g];
    bool full = v.full[g] != 0;
    for (uint32_t i = 0; i < n_rows; i++) {
        const size_t o = (size_t)i * v.G + g;
        const uint32_t kd = kind[o];
        uint8_t st = 0; uint32_t ro = 0, rl = 0;
        if (kd <= 1) {
            const uint32_t ko = key_off[o], kl = key_len[o];
            const uint32_t vo = kd == 1 ? val_off[o] : 0u, vl = kd == 1 ? val_len[o] : 0u;
            const bool in_buf = (uint64_t)ko + kl <= payload_bytes && (uint64_t)vo + vl <= payload_bytes && kl != SKV_EMPTY;
            if (full || !in_buf) st = 2;
            else {
                const uint8_t *kp = payload + ko;
                const uint64_t h = fnv1a(kp, kl);
                uint32_t s = (uint32_t)h & v.mask;
                bool found = false, free_slot = false;
                for (uint32_t probe = 0; probe < v.slots; probe++, s = (s + 1) & v.mask) {
                    const size_t e = (size_t)s * v.G + g;
                    const uint32_t el = v.klen[e];
                    if (el == SKV_EMPTY) { free_slot = true; break; }
                    if (el == kl && v.hash[e] == h && bytes_eq(heap + v.koff[e], kp, kl)) { found = true; break; }
                }
                const size_t e = (size_t)s * v.G + g;
                if (found) { st = 1; ro = v.voff[e]; rl = v.vlen[e]; }          // state.get(key) / what insert() returns
                if (kd == 1) {                                                  // state.insert(key, value)
                    const uint64_t need = (uint64_t)vl + (found ? 0u : kl);
                    if ((!found && !free_slot) || (uint64_t)used + need > v.heap_bytes) { full = true; st = 2; ro = rl = 0; }
                    else {
                        if (!found) {
                            for (uint32_t b = 0; b < kl; b++) heap[used + b] = kp[b];
                            v.hash[e] = h; v.koff[e] = used; v.klen[e] = kl;
                            used += kl; nk++;
                        }
                        const uint8_t *vp = payload + vo;
                        for (uint32_t b = 0; b < vl; b++) heap[used + b] = vp[b];
                        v.voff[e] = used; v.vlen[e] = vl;
                        used += vl;
                    }
                }
            }
        }
        res_state[o] = st; res_off[o] = ro; res_len[o] = rl;
    }
    v.heap_used[g] = used; v.n_keys[g] = nk;
    if (full) v.full[g] = 1;
}

}  // namespace smr

using namespace smr;

struct smr_skv { SkvView v; Arena arena; };

extern "C" {

int smr_skv_create(uint32_t n_groups, uint32_t slots, uint64_t heap_bytes, smr_skv **out) {
    if (!out || !n_groups) return fail(SMR_ERR_ARG, "skv: n_groups is zero");
    if (slots < 2 || (slots & (slots - 1))) return fail(SMR_ERR_ARG, "skv: slots must be a power of two >= 2");
    if (heap_bytes == 0 || heap_bytes > 0xFFFFFFFFull) return fail(SMR_ERR_ARG, "skv: heap_bytes must be in 1 .. 2^32 - 1");
    smr_skv *h = new smr_skv();
    SkvView &v = h->v;
    v.G = n_groups; v.slots = slots; v.mask = slots - 1; v.heap_bytes = heap_bytes;
    const size_t G = n_groups, SG = (size_t)slots * G;
    Arena &a = h->arena;
    const size_t o_h = a.reserve(SG * 8), o_ko = a.reserve(SG * 4), o_kl = a.reserve(SG * 4), o_vo = a.reserve(SG * 4), o_vl = a.reserve(SG * 4),
                 o_hp = a.reserve(G * heap_bytes), o_u = a.reserve(G * 4), o_f = a.reserve(G), o_n = a.reserve(G * 4);
    a.size = a.used;
    hipError_t e = hipMalloc((void **)&a.base, a.size);
    if (e == hipSuccess) e = hipMemset(a.base, 0, a.size);
    if (e == hipSuccess) e = hipMemset(a.base + o_kl, 0xFF, SG * 4);          // every entry free
    if (e != hipSuccess) { if (a.base) (void)hipFree(a.base); delete h; return fail(SMR_ERR_DEVICE, std::string("skv: alloc: ") + hipGetErrorString(e)); }
    v.hash = a.at<uint64_t>(o_h); v.koff = a.at<uint32_t>(o_ko); v.klen = a.at<uint32_t>(o_kl); v.voff = a.at<uint32_t>(o_vo);
    v.vlen = a.at<uint32_t>(o_vl); v.heap = a.at<uint8_t>(o_hp); v.heap_used = a.at<uint32_t>(o_u); v.full = a.at<uint8_t>(o_f);
    v.n_keys = a.at<uint32_t>(o_n);
    *out = h;
    return SMR_OK;
}

void smr_skv_destroy(smr_skv *h) {
    if (!h) return;
    (void)hipDeviceSynchronize();
    if (h->arena.base) (void)hipFree(h->arena.base);
    delete h;
}

int smr_skv_execute(smr_skv *h, uint32_t n_rows, const uint8_t *kind_dev, const uint8_t *payload_dev, uint64_t payload_bytes,
                    const uint32_t *key_off_dev, const uint32_t *key_len_dev, const uint32_t *val_off_dev, const uint32_t *val_len_dev,
                    uint8_t *res_state_dev, uint32_t *res_off_dev, uint32_t *res_len_dev, void *stream) {
    if (!h || !kind_dev || !key_off_dev || !key_len_dev || !val_off_dev || !val_len_dev || !res_state_dev || !res_off_dev || !res_len_dev ||
        (!payload_dev && payload_bytes))
        return fail(SMR_ERR_ARG, "skv: null argument");
    if (!n_rows) return SMR_OK;
    hipLaunchKernelGGL(skv_execute_kernel, dim3((h->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->v, n_rows, kind_dev, payload_dev,
                       payload_bytes, key_off_dev, key_len_dev, val_off_dev, val_len_dev, res_state_dev, res_off_dev, res_len_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_skv_heap(smr_skv *h, uint8_t **heap_dev, uint64_t *heap_bytes_per_group) {
    if (!h || !heap_dev) return fail(SMR_ERR_ARG, "skv: null argument");
    *heap_dev = h->v.heap;
    if (heap_bytes_per_group) *heap_bytes_per_group = h->v.heap_bytes;
    return SMR_OK;
}

int smr_skv_read(smr_skv *h, uint32_t group, uint32_t off, uint32_t len, uint8_t *host_buf) {
    if (!h || (!host_buf && len)) return fail(SMR_ERR_ARG, "skv: null argument");
    if (group >= h->v.G || (uint64_t)off + len > h->v.heap_bytes) return fail(SMR_ERR_ARG, "skv: bytes outside the group's heap");
    SMR_HIP_TRY(hipDeviceSynchronize());
    if (len) SMR_HIP_TRY(hipMemcpy(host_buf, h->v.heap + (size_t)group * h->v.heap_bytes + off, len, hipMemcpyDeviceToHost));
    return SMR_OK;
}

int smr_skv_stats(smr_skv *h, uint32_t *n_keys_host, uint32_t *heap_used_host, uint8_t *full_host) {
    if (!h || !n_keys_host || !heap_used_host || !full_host) return fail(SMR_ERR_ARG, "skv: null argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    SMR_HIP_TRY(hipMemcpy(n_keys_host, h->v.n_keys, (size_t)h->v.G * 4, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(heap_used_host, h->v.heap_used, (size_t)h->v.G * 4, hipMemcpyDeviceToHost));
    SMR_HIP_TRY(hipMemcpy(full_host, h->v.full, (size_t)h->v.G, hipMemcpyDeviceToHost));
    return SMR_OK;
}

}  // extern "C"
