// A host loop over the C-ABI, shaped like the reference's `run()` select loop
// (src/protocols/multipaxos/mod.rs:834-997) for a process that owns G groups: every batch interval
// it hands the engine what arrived (client batches, peers' replies, timer events) and takes back what
// to send, log and execute.  Here the "network" is synthetic: every follower answers every Accept.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/mp_host_loop.cpp -Lsummerset_amd -lsummerset_hip \
//         -Wl,-rpath,$PWD/summerset_amd -o mp_host_loop && ./mp_host_loop 4096 32 100
//
// Only plain C types cross the boundary; HIP is used here for the input buffers alone.
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

int main(int argc, char **argv) {
    const uint32_t G = argc > 1 ? (uint32_t)atoi(argv[1]) : 4096, S = argc > 2 ? (uint32_t)atoi(argv[2]) : 32;
    const int ticks = argc > 3 ? atoi(argv[3]) : 100;
    const uint32_t W = 512, cap = W + 4;
    if (smr_device_count() <= 0) { fprintf(stderr, "no device: %s\n", smr_last_error()); return 1; }

    smr_mp_cfg cfg = {};
    cfg.n_groups = G; cfg.population = 5; cfg.window = W; cfg.win_reserve = W / 8; cfg.outbox_cap = cap;
    cfg.commit_list_cap = 0;                                    // counters only; set it to poll (group, slot) pairs
    smr_mp_cluster *eng = nullptr;
    CHECK(smr_mp_cluster_create(&cfg, &eng));
    CHECK(smr_mp_preset_leader(eng, 0));                        // replica 0 leads every group, ballot 0x101

    // what the ExternalApi ticker would hand over: S batches per group for the leader (opaque tokens)
    std::vector<uint8_t> target(G, 0);
    std::vector<uint32_t> cnt(G, S), tok((size_t)S * G);
    for (size_t i = 0; i < tok.size(); i++) tok[i] = (uint32_t)(i * 2654435761u) | 1u;
    uint8_t *d_target; uint32_t *d_cnt, *d_tok;
    HIPCHECK(hipMalloc((void **)&d_target, G)); HIPCHECK(hipMalloc((void **)&d_cnt, G * 4));
    HIPCHECK(hipMalloc((void **)&d_tok, tok.size() * 4));
    HIPCHECK(hipMemcpy(d_target, target.data(), G, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_cnt, cnt.data(), G * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_tok, tok.data(), tok.size() * 4, hipMemcpyHostToDevice));

    hipStream_t st;
    HIPCHECK(hipStreamCreate(&st));
    hipEvent_t t0, t1;
    HIPCHECK(hipEventCreate(&t0)); HIPCHECK(hipEventCreate(&t1));
    HIPCHECK(hipEventRecord(t0, st));
    for (int t = 0; t < ticks; t++) {
        // R1: client batches (no timer fired: no timeout arrays); the leader's Accepts are in its outbox
        CHECK(smr_mp_round_local(eng, nullptr, nullptr, d_target, d_cnt, d_tok, S, st));
        // R2: the followers take the Accepts and answer (a real host would ship the outbox over its
        //     TransportHub here and write the peers' AcceptReplies into smr_mp_ack_matrix())
        CHECK(smr_mp_round_deliver(eng, st));
        // R3: the leader tallies; identity delivery order, nothing lost (ackctl = NULL)
        const int hb = (t % 4) == 3;
        CHECK(smr_mp_round_replies(eng, nullptr, hb, st));
        if (hb) CHECK(smr_mp_round_heartbeat(eng, st));         // followers learn the commit bar, rings trim
        CHECK(smr_mp_end_tick(eng));
    }
    HIPCHECK(hipEventRecord(t1, st));
    HIPCHECK(hipStreamSynchronize(st));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, t0, t1));
    uint64_t c[4] = {0, 0, 0, 0};
    CHECK(smr_mp_counters(eng, 0, c));
    printf("%u groups x 5 replicas, S = %u, %d ticks: %llu slots committed by replica 0, %.3f ms per tick, %.3g slots/s\n", G, S,
           ticks, (unsigned long long)c[0], ms / ticks, c[0] / (ms * 1e-3));
    smr_mp_group_state gs;
    CHECK(smr_mp_read_group_state(eng, 0, 0, &gs));
    printf("group 0, replica 0: leader %u, commit_bar %u, exec_bar %u, log_len %u\n", gs.leader, gs.commit_bar, gs.exec_bar,
           gs.log_len);
    smr_mp_cluster_destroy(eng);
    (void)hipFree(d_target); (void)hipFree(d_cnt); (void)hipFree(d_tok);
    return 0;
}
