// RSPaxos with real bytes over the C-ABI, from C++: five replica objects (smr_rsp_*) decide which shards exist where, five
// payload stores (smr_rsp_pstore_*) hold the bytes in device memory.  The run the reference spreads over five processes
// (src/protocols/rspaxos/): a leader that serializes, RS(3,2)-encodes and fans out one batch per group and tick
// (request.rs:71-142), followers holding ONE shard each (messages.rs:343-403), commits at majority + fault_tolerance acks
// and execution at the leader (durability.rs:125-186) -- then the leader goes away: replica 1 steps up
// (leadership.rs:47-185), collects voted shards in the Prepare phase (messages.rs:87-340), reads the committed instances'
// shards back from two peers (messages.rs:467-594), reconstructs each batch from shards {1, 2, 3} and executes it.  Every
// batch a replica executes is read out of its store (RSCodeword::get_data, rscoding.rs:583-609) and compared with the bytes
// the old leader serialized.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/rsp_payload_loop.cpp -Lsummerset_amd -lsummerset_hip \
//         -Wl,-rpath,$PWD/summerset_amd -o rsp_payload_loop && ./rsp_payload_loop 1024 1000
//
// Only plain C types cross the boundary; HIP is used here for the message buffers alone.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "summerset_hip.h"

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != SMR_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, smr_last_error()); return 1; } \
    } while (0)
#define HIPCHECK(call)                                                                  \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } \
    } while (0)

template <typename T> static T *dalloc(size_t n, int fill = 0) {
    T *p = nullptr;
    if (hipMalloc((void **)&p, n * sizeof(T)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(1); }
    (void)hipMemset(p, fill, n * sizeof(T));
    return p;
}
template <typename T> static std::vector<T> to_host(const T *d, size_t n) {
    std::vector<T> h(n);
    (void)hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost);
    return h;
}

static const uint32_t R = 5, W = 16;
static uint32_t G, L;
// the serialized batch behind a token: its bytes and its length are functions of the token alone
static uint32_t batch_len(uint32_t tok) { return 1 + (uint32_t)(((uint64_t)tok * 7919u) % L); }
static uint8_t batch_byte(uint32_t tok, uint32_t i) { return (uint8_t)((((uint64_t)tok * 2654435761ull + (uint64_t)i * 40503ull) >> 7) & 0xFF); }

// what `rep`'s LAST handler call executed, read out of `store` and compared with the batches' bytes; -1: a mismatch
static long check_executed(smr_rsp_replica *rep, smr_rsp_pstore *store) {
    uint64_t n = 0;
    if (smr_rsp_exec_poll(rep, nullptr, nullptr, nullptr, 0, &n) != SMR_OK) return -1;
    if (n == 0) return 0;
    std::vector<uint32_t> g(n), s(n), v(n);
    if (smr_rsp_exec_poll(rep, g.data(), s.data(), v.data(), n, &n) != SMR_OK) return -1;
    uint32_t *d_g = dalloc<uint32_t>(n), *d_s = dalloc<uint32_t>(n), *d_v = dalloc<uint32_t>(n), *d_len = dalloc<uint32_t>(n);
    uint8_t *d_ok = dalloc<uint8_t>(n), *d_out = dalloc<uint8_t>(n * L);
    (void)hipMemcpy(d_g, g.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_s, s.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_v, v.data(), n * 4, hipMemcpyHostToDevice);
    long bad = smr_rsp_pstore_get_data(store, (uint32_t)n, d_g, d_s, d_v, d_out, L, d_len, d_ok, nullptr) != SMR_OK;
    (void)hipDeviceSynchronize();
    const std::vector<uint8_t> ok = to_host(d_ok, n), out = to_host(d_out, n * L);
    const std::vector<uint32_t> len = to_host(d_len, n);
    for (uint64_t i = 0; i < n && !bad; i++) {
        bad = !ok[i] || len[i] != batch_len(v[i]);
        for (uint32_t b = 0; b < len[i] && !bad; b++) bad = out[i * L + b] != batch_byte(v[i], b);
        if (bad) fprintf(stderr, "group %u slot %u token %u: the store's bytes are not the batch\n", g[i], s[i], v[i]);
    }
    (void)hipFree(d_g); (void)hipFree(d_s); (void)hipFree(d_v); (void)hipFree(d_len); (void)hipFree(d_ok); (void)hipFree(d_out);
    return bad ? -1 : (long)n;
}

int main(int argc, char **argv) {
    G = argc > 1 ? (uint32_t)atoi(argv[1]) : 1024;
    L = argc > 2 ? (uint32_t)atoi(argv[2]) : 1000;
    const uint32_t T = 6;                                                 // slots: T - 2 commit under leader 0, the last two stay open
#ifndef RSP_PAYLOAD_LOOP_ON_THE_EMULATOR        /* tests/test_hostsim.py compiles this file against the kernel-source emulator */
    if (smr_device_count() <= 0) { fprintf(stderr, "no device: %s\n", smr_last_error()); return 1; }
#endif
    smr_rsp_replica *rep[R];
    smr_rsp_pstore *store[R];
    for (uint32_t r = 0; r < R; r++) {
        smr_rsp_cfg cfg = {};
        cfg.n_groups = G; cfg.population = R; cfg.me = (uint8_t)r; cfg.fault_tolerance = 1; cfg.window = W;
        CHECK(smr_rsp_replica_create(&cfg, &rep[r]));
        CHECK(smr_rsp_preset_leader(rep[r], 0));
        CHECK(smr_rsp_pstore_create(G, R, R / 2 + 1, W, L, &store[r]));
    }
    // the peers' stores by replica id, as sources of either plane (a co-located cluster: the stores ARE the messages' payload)
    uint8_t plane_reqs[R] = {0, 0, 0, 0, 0}, plane_voted[R] = {1, 1, 1, 1, 1};
    auto peers_of = [&](uint32_t me, smr_rsp_pstore **out) { for (uint32_t r = 0; r < R; r++) out[r] = r == me ? nullptr : store[r]; };

    // message buffers (device): the Accepts of a call, the replies to them, a Prepare, its replies, reconstruction reads
    smr_rsp_accepts acc = {dalloc<uint32_t>(G), dalloc<uint32_t>((size_t)W * G), dalloc<uint32_t>((size_t)W * G), dalloc<uint64_t>(G)};
    uint64_t *r_ballot = dalloc<uint64_t>((size_t)R * G);
    uint32_t *r_slot = dalloc<uint32_t>(G), *tok_d = dalloc<uint32_t>(G), *len_d = dalloc<uint32_t>(G);
    uint8_t *r_flags = dalloc<uint8_t>((size_t)R * G), *committed = dalloc<uint8_t>(G), *ones = dalloc<uint8_t>(G, 1), *data_d = dalloc<uint8_t>((size_t)G * L);
    uint8_t *is_peer[R], *shard_of[R];
    for (uint32_t r = 0; r < R; r++) { is_peer[r] = dalloc<uint8_t>(G, (int)r); shard_of[r] = dalloc<uint8_t>(G, 1 << r); }
    smr_rsp_heartbeat hb = {dalloc<uint8_t>(G), dalloc<uint64_t>(G), dalloc<uint32_t>(G), dalloc<uint32_t>(G), dalloc<uint32_t>(G)};
    smr_rsp_heartbeat hb_back = {nullptr, dalloc<uint64_t>(G), dalloc<uint32_t>(G), dalloc<uint32_t>(G), dalloc<uint32_t>(G)};
    uint8_t *hb_reply = dalloc<uint8_t>(G);

    std::vector<uint32_t> tok(G), len(G);
    std::vector<uint8_t> data((size_t)G * L);
    long by_old = 0, by_new = 0;
    // ---- steady state under leader 0 -----------------------------------------------------------------------------------
    for (uint32_t t = 0; t < T; t++) {
        for (uint32_t g = 0; g < G; g++) {
            tok[g] = 1 + t * G + g; len[g] = batch_len(tok[g]);
            for (uint32_t i = 0; i < L; i++) data[(size_t)g * L + i] = i < len[g] ? batch_byte(tok[g], i) : 0x5A;   // (junk past the length)
        }
        HIPCHECK(hipMemcpy(tok_d, tok.data(), G * 4, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(len_d, len.data(), G * 4, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(data_d, data.data(), (size_t)G * L, hipMemcpyHostToDevice));
        CHECK(smr_rsp_req_batch(rep[0], tok_d, &acc, nullptr));                                          // handle_req_batch: which slot
        CHECK(smr_rsp_pstore_put(store[0], acc.n, acc.slot, acc.val, data_d, L, len_d, L, nullptr));     // from_data + compute_parity
        CHECK(smr_rsp_pstore_follow(store[0], rep[0], 0, nullptr, nullptr, nullptr, nullptr));           // (its own vote: shard 0)
        HIPCHECK(hipMemset(r_flags, 0, (size_t)R * G));
        const uint32_t reach = t < T - 2 ? R : 3;                                                         // the last two only reach followers 1 and 2
        for (uint32_t q = 1; q < reach; q++) {
            CHECK(smr_rsp_handle_accept(rep[q], ones, is_peer[0], acc.slot, acc.ballot, acc.val, shard_of[q], r_ballot + (size_t)q * G, r_slot, nullptr));
            smr_rsp_pstore *src[R];
            peers_of(q, src);
            CHECK(smr_rsp_pstore_follow(store[q], rep[q], R, src, plane_reqs, is_peer[0], nullptr));     // the Accept's payload: shard q of the leader's row
            HIPCHECK(hipMemset(r_flags + (size_t)q * G, 1, G));
        }
        CHECK(smr_rsp_handle_accept_replies(rep[0], acc.slot, r_ballot, r_flags, nullptr, committed, nullptr));
        CHECK(smr_rsp_pstore_follow(store[0], rep[0], 0, nullptr, nullptr, nullptr, nullptr));
        const long n = check_executed(rep[0], store[0]);
        if (n < 0) return 2;
        by_old += n;
    }
    // followers learn the commits from a Heartbeat; with one shard each they cannot run them
    CHECK(smr_rsp_bcast_heartbeat(rep[0], ones, &hb_back, nullptr));
    HIPCHECK(hipMemset(hb.flags, 1, G));
    HIPCHECK(hipMemcpy(hb.ballot, hb_back.ballot, G * 8, hipMemcpyDeviceToDevice));
    HIPCHECK(hipMemcpy(hb.commit_bar, hb_back.commit_bar, G * 4, hipMemcpyDeviceToDevice));
    HIPCHECK(hipMemcpy(hb.exec_bar, hb_back.exec_bar, G * 4, hipMemcpyDeviceToDevice));
    HIPCHECK(hipMemcpy(hb.snap_bar, hb_back.snap_bar, G * 4, hipMemcpyDeviceToDevice));
    for (uint32_t q = 1; q < R; q++) {
        CHECK(smr_rsp_handle_heartbeat(rep[q], is_peer[0], &hb, hb_reply, &hb_back, nullptr));
        CHECK(smr_rsp_pstore_follow(store[q], rep[q], 0, nullptr, nullptr, nullptr, nullptr));
    }
    // ---- replica 1 takes over ------------------------------------------------------------------------------------------
    uint8_t *p_flags = dalloc<uint8_t>(G);
    uint32_t *p_trig = dalloc<uint32_t>(G), *rc_n = dalloc<uint32_t>(G), *rc_slot = dalloc<uint32_t>((size_t)W * G);
    uint64_t *p_ballot = dalloc<uint64_t>(G);
    CHECK(smr_rsp_become_leader(rep[1], is_peer[0], &hb, p_flags, p_trig, p_ballot, rc_n, rc_slot, nullptr));
    CHECK(smr_rsp_pstore_follow(store[1], rep[1], 0, nullptr, nullptr, nullptr, nullptr));
    smr_rsp_prepare_reply pr = {dalloc<uint32_t>(G), dalloc<uint32_t>(G), dalloc<uint32_t>(G), dalloc<uint64_t>(G), dalloc<uint64_t>((size_t)W * G),
                                dalloc<uint32_t>((size_t)W * G, 0xFF), dalloc<uint8_t>((size_t)W * G)};
    smr_rsp_accepts re = {dalloc<uint32_t>(G), dalloc<uint32_t>((size_t)W * G), dalloc<uint32_t>((size_t)W * G), dalloc<uint64_t>(G)};
    smr_rsp_pstore *src1[R];
    peers_of(1, src1);
    for (uint32_t q : {2u, 3u, 4u}) {                                     // replica 0 is gone; 4 never saw the open instances
        CHECK(smr_rsp_handle_prepare(rep[q], ones, is_peer[1], p_trig, p_ballot, &pr, nullptr));
        CHECK(smr_rsp_pstore_follow(store[q], rep[q], 0, nullptr, nullptr, nullptr, nullptr));
        CHECK(smr_rsp_handle_prepare_replies(rep[1], is_peer[q], &pr, &re, nullptr));
        CHECK(smr_rsp_pstore_follow(store[1], rep[1], R, src1, plane_voted, is_peer[q], nullptr));       // the reply's payload: q's voted shards
    }
    // 4 replies >= population - f with two shards per open instance: both become the empty batch and are re-Accepted
    const std::vector<uint32_t> re_n = to_host(re.n, G), re_val = to_host(re.val, (size_t)W * G);
    unsigned long long empties = 0;
    for (uint32_t g = 0; g < G; g++)
        for (uint32_t k = 0; k < re_n[g]; k++) empties += re_val[(size_t)k * G + g] == 0;
    // the committed instances: reconstruction reads bring shards 2 and 3 in; with its own shard 1 the new leader has three
    smr_rsp_shards rr = {dalloc<uint32_t>(G), dalloc<uint32_t>((size_t)W * G), dalloc<uint64_t>((size_t)W * G), dalloc<uint32_t>((size_t)W * G, 0xFF),
                         dalloc<uint8_t>((size_t)W * G)};
    hipEvent_t ev0, ev1;
    HIPCHECK(hipEventCreate(&ev0)); HIPCHECK(hipEventCreate(&ev1));
    float follow_ms[2] = {0, 0};                                           // the new leader's follow behind each ReconstructReply
    for (uint32_t q : {2u, 3u}) {
        CHECK(smr_rsp_handle_reconstruct(rep[q], ones, rc_n, rc_slot, &rr, nullptr));
        CHECK(smr_rsp_pstore_follow(store[q], rep[q], 0, nullptr, nullptr, nullptr, nullptr));
        CHECK(smr_rsp_handle_reconstruct_reply(rep[1], ones, &rr, nullptr));
        HIPCHECK(hipEventRecord(ev0, nullptr));
        CHECK(smr_rsp_pstore_follow(store[1], rep[1], R, src1, plane_reqs, is_peer[q], nullptr));        // the reply's payload: q's shards
        HIPCHECK(hipEventRecord(ev1, nullptr));
        HIPCHECK(hipEventSynchronize(ev1));
        HIPCHECK(hipEventElapsedTime(&follow_ms[q - 2], ev0, ev1));
        const long n = check_executed(rep[1], store[1]);                  // reconstruct_data (shard 0 rebuilt from {1, 2, 3}), then execution
        if (n < 0) return 2;
        by_new += n;
    }
    uint64_t c[4], tot[4] = {0, 0, 0, 0};
    for (uint32_t r = 0; r < R; r++) {
        CHECK(smr_rsp_pstore_counters(store[r], c));
        for (int k = 0; k < 4; k++) tot[k] += c[k];
    }
    printf("%u groups x 5 replicas, batches of up to %u bytes: %ld batches read back byte for byte at the old leader, %ld at the new leader "
           "after reconstruction, %llu open instances became empty batches\n", G, L, by_old, by_new, empties);
    printf("new leader's follow behind the ReconstructReplies: %.3f ms (absorb one shard of %u instances per group), %.3f ms (absorb one + rebuild "
           "one data shard of each from three: <= %.1f MB of shards rebuilt)\n", follow_ms[0], T - 2, follow_ms[1],
           (double)(T - 2) * G * ((L + 2) / 3) / 1e6);
    printf("payload stores: %llu shards copied, %llu rebuilt, %llu unsatisfied\n", (unsigned long long)tot[0], (unsigned long long)tot[1],
           (unsigned long long)tot[2]);
    for (uint32_t r = 0; r < R; r++) { smr_rsp_pstore_destroy(store[r]); smr_rsp_replica_destroy(rep[r]); }
    const bool good = by_old == (long)(T - 2) * G && by_new == (long)(T - 2) * G && empties == 2ull * G && tot[2] == 0 && tot[1] > 0;
    printf("%s\n", good ? "ok" : "UNEXPECTED");
    return good ? 0 : 2;
}
