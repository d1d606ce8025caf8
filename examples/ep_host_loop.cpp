// An EPaxos host loop over the C-ABI for a process that holds ALL five replicas of its G groups on one device
// (BASELINE config 5 on one GPU; a simulation / replay driver): one smr_ep_cluster_tick call per batch interval runs the
// tick the reference spreads over five processes' `run()` loops (src/protocols/epaxos/mod.rs) -- every replica proposes,
// PreAccept fan-out, fast / slow decision, the Accept round, CommitNotices, dependency-graph execution behind every handler.
// The "network" is the device: nothing is lost, so every proposed instance commits inside its tick and every command
// executes.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/ep_host_loop.cpp -Lsummerset_amd -lsummerset_hip \
//         -Wl,-rpath,$PWD/summerset_amd -o ep_host_loop && ./ep_host_loop 4096 20
//
// Only plain C types cross the boundary; HIP is used here for the key / output buffers alone.
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
    const uint32_t G = argc > 1 ? (uint32_t)atoi(argv[1]) : 4096;
    const int ticks = argc > 2 ? atoi(argv[2]) : 20;
    const uint32_t R = 5, W = 32, K = 64, POOL = 4;
#ifndef EP_HOST_LOOP_ON_THE_EMULATOR            /* tests/test_hostsim.py compiles this file against the kernel-source emulator */
    if (smr_device_count() <= 0) { fprintf(stderr, "no device: %s\n", smr_last_error()); return 1; }
#endif

    smr_ep_replica *rep[R];
    for (uint32_t r = 0; r < R; r++) {
        smr_ep_cfg cfg = {};
        cfg.n_groups = G; cfg.population = R; cfg.me = (uint8_t)r; cfg.optimized_quorum = 1; cfg.execute = 1; cfg.window = W; cfg.n_keys = K;
        CHECK(smr_ep_replica_create(&cfg, &rep[r]));
    }
    smr_ep_cluster *cl = nullptr;
    CHECK(smr_ep_cluster_create(rep, R, &cl));

    // what five ExternalApi tickers would hand over: one Put per replica, group and tick on a key that collides often
    uint8_t *d_keys[POOL][R];
    std::vector<uint8_t> host(G);
    for (uint32_t p = 0; p < POOL; p++)
        for (uint32_t r = 0; r < R; r++) {
            for (uint32_t g = 0; g < G; g++) host[g] = (uint8_t)(((g * 2654435761u) >> 13) + 3 * r + 7 * p) % (g % 3 ? 2 : K);   // (two thirds of the groups: replicas 0, 2, 4 write one key)
            HIPCHECK(hipMalloc((void **)&d_keys[p][r], G));
            HIPCHECK(hipMemcpy(d_keys[p][r], host.data(), G, hipMemcpyHostToDevice));
        }
    smr_ep_cluster_out out[R];
    for (uint32_t s = 0; s < R; s++) {
        HIPCHECK(hipMalloc((void **)&out[s].proposed, G)); HIPCHECK(hipMalloc((void **)&out[s].col, G * 4));
        HIPCHECK(hipMalloc((void **)&out[s].seq0, G * 8)); HIPCHECK(hipMalloc((void **)&out[s].deps0, (size_t)R * G * 4));
        HIPCHECK(hipMalloc((void **)&out[s].decision, G)); HIPCHECK(hipMalloc((void **)&out[s].committed, G));
        HIPCHECK(hipMalloc((void **)&out[s].seq, G * 8)); HIPCHECK(hipMalloc((void **)&out[s].deps, (size_t)R * G * 4));
    }

    hipStream_t st;
    HIPCHECK(hipStreamCreate(&st));
    hipEvent_t t0, t1;
    HIPCHECK(hipEventCreate(&t0)); HIPCHECK(hipEventCreate(&t1));
    unsigned long long committed = 0, fast = 0;
    std::vector<uint8_t> c(G), d(G);
    float ms_total = 0;
    for (int t = 0; t < ticks; t++) {
        const uint8_t *keys[R];
        for (uint32_t r = 0; r < R; r++) keys[r] = d_keys[t % POOL][r];
        HIPCHECK(hipEventRecord(t0, st));
        CHECK(smr_ep_cluster_tick(cl, keys, nullptr, out, st));          // the whole tick: no host work between its handlers
        HIPCHECK(hipEventRecord(t1, st));
        HIPCHECK(hipStreamSynchronize(st));
        float ms = 0;
        HIPCHECK(hipEventElapsedTime(&ms, t0, t1));
        ms_total += ms;
        for (uint32_t s = 0; s < R; s++) {                                // (a real host would reply to its clients from these)
            HIPCHECK(hipMemcpy(c.data(), out[s].committed, G, hipMemcpyDeviceToHost));
            HIPCHECK(hipMemcpy(d.data(), out[s].decision, G, hipMemcpyDeviceToHost));
            for (uint32_t g = 0; g < G; g++) { committed += c[g]; fast += d[g] == 3; }
        }
    }
    unsigned long long executed = 0;
    std::vector<uint32_t> eb((size_t)R * G);
    std::vector<uint64_t> kv((size_t)K * G), dg(G);
    uint64_t first_digest = 0;
    bool same = true;
    for (uint32_t r = 0; r < R; r++) {
        uint64_t ctr[6] = {0, 0, 0, 0, 0, 0};
        CHECK(smr_ep_exec_dump(rep[r], eb.data(), kv.data(), dg.data(), ctr));
        executed += ctr[0];
        uint64_t x = 0;
        for (size_t i = 0; i < kv.size(); i++) x = (x ^ kv[i]) * 0x100000001B3ull;    // the replicas' KV stores must agree
        if (r == 0) first_digest = x; else same = same && x == first_digest;
    }
    printf("%u groups x 5 replicas, %d ticks: %llu instances committed (%llu on the fast path), %llu commands executed, "
           "%.3f ms per tick, %.3g instances/s\n", G, ticks, committed, fast, executed, ms_total / ticks, committed / (ms_total * 1e-3));
    printf("replicas' KV stores %s\n", same ? "agree" : "DIFFER");
    smr_ep_cluster_destroy(cl);
    for (uint32_t r = 0; r < R; r++) smr_ep_replica_destroy(rep[r]);
    return same ? 0 : 2;
}
