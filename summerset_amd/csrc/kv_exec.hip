// Device-resident KV state machine over G groups: StateMachineExecutorTask::execute
// (src/server/statemach.rs:193-202) applied to each group's commands in submission order, lane = group.
// Model shared with qread.hip and the EPaxos execution kernel: keys < K, a value is a 32-bit token, 0 = None.
// The table kv[K][G] is what a stable leased leader answers ReadQueries from (smr_qread_handle_read_query).
#include <string.h>

#include "smr_common.h"

namespace smr {

struct KvView { uint32_t G, K; uint32_t *kv; };

// kind[B][G]: 0 Get, 1 Put, anything else = no command in that row; res = Get: value, Put: old_value (0 = None)
__global__ __launch_bounds__(256) void kv_execute_kernel(const KvView v, uint32_t B, const uint8_t *__restrict__ kind,
                                                         const uint8_t *__restrict__ key, const uint32_t *__restrict__ val,
                                                         uint32_t *__restrict__ res) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= v.G) return;
    for (uint32_t i = 0; i < B; i++) {
        const size_t o = (size_t)i * v.G + g;
        const uint32_t kd = kind[o], k = key[o];
        uint32_t r = 0;
        if (kd <= 1 && k < v.K) {
            uint32_t *cell = &v.kv[(size_t)k * v.G + g];
            r = *cell;                                       // Get: state.get(key).cloned(); Put: what insert() returns
            if (kd == 1) *cell = val[o];                     // state.insert(key, value)
        }
        res[o] = r;
    }
}

}  // namespace smr

using namespace smr;

struct smr_kv { KvView v; };

extern "C" {

int smr_kv_create(uint32_t n_groups, uint32_t n_keys, smr_kv **out) {
    if (!out || !n_groups || !n_keys || n_keys > 255) return fail(SMR_ERR_ARG, "kv: n_groups > 0 and n_keys in 1..255");
    smr_kv *h = new smr_kv();
    h->v.G = n_groups; h->v.K = n_keys; h->v.kv = nullptr;
    const size_t bytes = (size_t)n_keys * n_groups * 4;
    hipError_t e = hipMalloc((void **)&h->v.kv, bytes);
    if (e == hipSuccess) e = hipMemset(h->v.kv, 0, bytes);
    if (e != hipSuccess) {
        if (h->v.kv) (void)hipFree(h->v.kv);
        delete h;
        return fail(SMR_ERR_DEVICE, std::string("kv: alloc: ") + hipGetErrorString(e));
    }
    *out = h;
    return SMR_OK;
}

void smr_kv_destroy(smr_kv *h) {
    if (!h) return;
    if (h->v.kv) (void)hipFree(h->v.kv);
    delete h;
}

int smr_kv_execute(smr_kv *h, uint32_t n_rows, const uint8_t *kind_dev, const uint8_t *key_dev, const uint32_t *val_dev,
                   uint32_t *res_dev, void *stream) {
    if (!h || !kind_dev || !key_dev || !val_dev || !res_dev) return fail(SMR_ERR_ARG, "kv: null argument");
    if (!n_rows) return SMR_OK;
    hipLaunchKernelGGL(kv_execute_kernel, dim3((h->v.G + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->v, n_rows, kind_dev, key_dev,
                       val_dev, res_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

int smr_kv_table(smr_kv *h, uint32_t **kv_dev) {
    if (!h || !kv_dev) return fail(SMR_ERR_ARG, "kv: null argument");
    *kv_dev = h->v.kv;
    return SMR_OK;
}

int smr_kv_dump(smr_kv *h, uint32_t *kv_host) {
    if (!h || !kv_host) return fail(SMR_ERR_ARG, "kv: null argument");
    SMR_HIP_TRY(hipDeviceSynchronize());
    SMR_HIP_TRY(hipMemcpy(kv_host, h->v.kv, (size_t)h->v.K * h->v.G * 4, hipMemcpyDeviceToHost));
    return SMR_OK;
}

}  // extern "C"
