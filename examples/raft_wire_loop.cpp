// A Raft replication loop over the C-ABI in which the followers' AppendEntriesReplies travel as WIRE FRAMES that never
// leave the device: replica 0 leads G groups, replicas 1 .. 4 follow; per batch interval
//   leader     smr_raft_leader_append_emit          handle_req_batch + what its appends send to every peer (raft/request.rs)
//   leader     smr_raft_leader_gather_entries       the AppendEntries for follower q as device arrays
//   follower   smr_raft_replica_handle_append_entries   raft/messages.rs:13-218 -> the reply arrays [G]
//   follower   smr_wire_emit_raft_replies           the reply as `[u64 BE length][bincode(PeerMessage)]` (safetcp.rs:127-132), slot g
//   leader     smr_wire_ingest_raft_replies         the four followers' slots as connections -> the [R][G] arrays
//   leader     smr_raft_leader_handle_replies       raft/messages.rs:243-309: match-index quorum, commit
// -- what five processes' `run()` loops and TcpTransport do for the reference, with the device as the network (nothing is
// lost, so an entry commits in the tick it was appended in).
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/raft_wire_loop.cpp -Lsummerset_amd -lsummerset_hip \
//         -Wl,-rpath,$PWD/summerset_amd -o raft_wire_loop && ./raft_wire_loop 4096 20
//
// Only plain C types cross the boundary; HIP is used here for the buffers alone.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

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
template <typename T> static T *dmalloc(size_t n) {
    void *p = nullptr;
    if (hipMalloc(&p, n * sizeof(T) + 16) != hipSuccess) return nullptr;
    (void)hipMemset(p, 0, n * sizeof(T) + 16);
    return (T *)p;
}

int main(int argc, char **argv) {
    const uint32_t G = argc > 1 ? (uint32_t)atoi(argv[1]) : 4096;
    const int ticks = argc > 2 ? atoi(argv[2]) : 20;
    const uint32_t R = 5, W = 64, S = 2, K = 8;                 // S new entries per group and tick, <= K per AppendEntries
#ifndef RAFT_WIRE_LOOP_ON_THE_EMULATOR          /* tests/test_hostsim.py compiles this file against the kernel-source emulator */
    if (smr_device_count() <= 0) { fprintf(stderr, "no device: %s\n", smr_last_error()); return 1; }
#endif
    smr_raft_leader *rep[R];
    for (uint32_t r = 0; r < R; r++) {
        smr_raft_cfg cfg = {};
        cfg.n_groups = G; cfg.population = R; cfg.leader_id = (uint8_t)r; cfg.window = W; cfg.term = 1;
        CHECK(smr_raft_leader_create(&cfg, &rep[r]));
        CHECK(smr_raft_replica_preset(rep[r], r == 0 ? 2 : 0, 0, 1, SMR_NO_REPLICA));   // replica 0 leads, the others follow it
    }
    // the tick's inputs and every message as device arrays
    std::vector<uint32_t> h_new(G, S);
    uint32_t *n_new = dmalloc<uint32_t>(G), *first_sent = dmalloc<uint32_t>((size_t)R * G);
    HIPCHECK(hipMemcpy(n_new, h_new.data(), G * 4, hipMemcpyHostToDevice));
    smr_raft_append_entries msg = {};
    msg.flags = dmalloc<uint8_t>(G); msg.leader = dmalloc<uint8_t>(G); msg.term = dmalloc<uint64_t>(G); msg.prev_slot = dmalloc<uint32_t>(G);
    msg.prev_term = dmalloc<uint64_t>(G); msg.n_entries = dmalloc<uint32_t>(G); msg.entry_term = dmalloc<uint64_t>((size_t)K * G);
    msg.max_entries = K; msg.leader_commit = dmalloc<uint32_t>(G); msg.last_snap = dmalloc<uint32_t>(G);
    smr_raft_append_reply reply = {dmalloc<uint8_t>(G), dmalloc<uint64_t>(G), dmalloc<uint32_t>(G), dmalloc<uint64_t>(G), dmalloc<uint32_t>(G)};
    // the "network": follower q's frame for group g in slot (q - 1) * G + g; the slots are the leader's connections
    const uint32_t n_conn = (R - 1) * G;
    uint8_t *frames = dmalloc<uint8_t>((size_t)n_conn * SMR_WIRE_EMIT_RAFT_STRIDE), *len = dmalloc<uint8_t>(n_conn);
    std::vector<uint64_t> h_off(n_conn);
    std::vector<uint32_t> h_grp(n_conn);
    std::vector<uint8_t> h_peer(n_conn);
    for (uint32_t c = 0; c < n_conn; c++) { h_off[c] = (uint64_t)c * SMR_WIRE_EMIT_RAFT_STRIDE; h_grp[c] = c % G; h_peer[c] = (uint8_t)(1 + c / G); }
    uint64_t *conn_off = dmalloc<uint64_t>(n_conn);
    uint32_t *conn_grp = dmalloc<uint32_t>(n_conn);
    uint8_t *conn_peer = dmalloc<uint8_t>(n_conn);
    HIPCHECK(hipMemcpy(conn_off, h_off.data(), n_conn * 8, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(conn_grp, h_grp.data(), n_conn * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(conn_peer, h_peer.data(), n_conn, hipMemcpyHostToDevice));
    uint64_t *r_term = dmalloc<uint64_t>((size_t)R * G), *r_cterm = dmalloc<uint64_t>((size_t)R * G);
    uint32_t *r_end = dmalloc<uint32_t>((size_t)R * G), *r_cslot = dmalloc<uint32_t>((size_t)R * G);
    uint8_t *r_flags = dmalloc<uint8_t>((size_t)R * G);
    smr_wire_other *others = dmalloc<smr_wire_other>(16);
    uint64_t *counts = dmalloc<uint64_t>(4), *consumed = dmalloc<uint64_t>(n_conn);
    int32_t *status = dmalloc<int32_t>(n_conn);
    if (!n_new || !first_sent || !msg.entry_term || !frames || !status) { fprintf(stderr, "out of device memory\n"); return 1; }

    uint64_t replies = 0, malformed = 0;
    for (int t = 0; t < ticks; t++) {                           // (an entry commits in the tick it was appended in: the replies come back within it)
        CHECK(smr_raft_leader_append_emit(rep[0], n_new, first_sent, nullptr));
        for (uint32_t q = 1; q < R; q++) {
            CHECK(smr_raft_leader_gather_entries(rep[0], first_sent + (size_t)q * G, &msg, nullptr));
            CHECK(smr_raft_replica_handle_append_entries(rep[q], &msg, &reply, nullptr));
            CHECK(smr_wire_emit_raft_replies(reply.flags, reply.term, reply.end_slot, reply.conflict_term, reply.conflict_slot, G,
                                             frames + (size_t)(q - 1) * G * SMR_WIRE_EMIT_RAFT_STRIDE, len + (size_t)(q - 1) * G, nullptr));
        }
        CHECK(smr_wire_ingest_raft_replies(frames, (uint64_t)n_conn * SMR_WIRE_EMIT_RAFT_STRIDE, conn_off, conn_grp, conn_peer, len, n_conn, G, (uint8_t)R,
                                           r_term, r_end, r_cterm, r_cslot, r_flags, others, 16, counts, consumed, status, nullptr));
        CHECK(smr_raft_leader_handle_replies(rep[0], r_term, r_end, r_cterm, r_cslot, r_flags, nullptr, nullptr));
        uint64_t h_counts[4];
        HIPCHECK(hipMemcpy(h_counts, counts, sizeof(h_counts), hipMemcpyDeviceToHost));
        replies += h_counts[0]; malformed += h_counts[2];
    }
    HIPCHECK(hipDeviceSynchronize());
    uint64_t commits = 0;
    CHECK(smr_raft_leader_total_commits(rep[0], &commits));
    printf("%llu entries committed by the leader of %u groups in %d ticks; %llu AppendEntriesReply frames written and parsed on the device, %llu malformed\n",
           (unsigned long long)commits, G, ticks, (unsigned long long)replies, (unsigned long long)malformed);
    for (uint32_t r = 0; r < R; r++) smr_raft_leader_destroy(rep[r]);
    return commits == (uint64_t)G * S * ticks && malformed == 0 ? 0 : 2;
}
