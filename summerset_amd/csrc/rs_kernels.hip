// Reed-Solomon over GF(2^8) for gfx950: shard-matrix products
//   out_shard[r][i] = XOR_c  M[r][c] * in_shard[c][i]      (GF(2^8), poly 0x11D)
// used for encode (M = parity rows), reconstruct (M = rows of the inverted
// sub-matrix, composed with parity rows) and verify.
//
// Stands in for reed_solomon_erasure::galois_8::ReedSolomon::{encode,
// reconstruct, reconstruct_data, verify} as called by
// src/utils/rscoding.rs:484,515-517,575, with RSCodeword::from_data's shard
// geometry (rscoding.rs:165-220) fused into the address arithmetic: the data
// shards are consecutive shard_len-byte slices of the serialized bytes and the
// zero padding is synthesised, never materialised.
//
// HBM-bound byte work (no MFMA).  One lane owns 16 consecutive byte columns of
// one codeword: d 16-byte loads, p 16-byte stores, so a wave moves 1 KiB per
// memory instruction.  Two arithmetic back ends:
//   * xtime : bit-sliced multiply-by-2 on 4 packed bytes per VGPR, evaluated per output row as
//             a Horner scheme over the coefficient BIT PLANES:
//                 out_r = XOR_b 2^b * (XOR_{c : bit b of M[r][c]} in_c)
//                       = (..((P_7) * 2 ^ P_6) * 2 ^ ..) * 2 ^ P_0
//             so a row costs (highest coefficient bit) doublings whatever the number of input
//             shards (RS(3,2): 3 doublings per 4 bytes of a codeword column instead of 9).  The
//             coefficients are wave-uniform kernel arguments: "is column c in plane b" is a
//             scalar branch;
//   * lut   : 256-entry product tables per coefficient, resident in LDS.
#include <map>
#include <mutex>

#include "smr_common.h"

namespace smr {

constexpr int RS_MAX_IN = 16;   // d  (data / present shards)
constexpr int RS_MAX_OUT = 8;   // p  (or number of shards to rebuild)

struct RsArgs {
    const uint8_t *in_base;
    uint8_t *out_base;
    uint64_t in_cw_stride, out_cw_stride;
    uint64_t in_off[RS_MAX_IN];    // byte offset of input shard c inside a codeword
    uint64_t out_off[RS_MAX_OUT];  // byte offset of output shard r inside a codeword
    uint8_t *copy_base;            // from_data fused in (rs_from_data_xtime): the input columns are also WRITTEN, shard c of codeword
    uint64_t copy_cw_stride;       // i to copy_base + i * copy_cw_stride + copy_off[c], zero padding included; NULL otherwise
    uint64_t copy_off[RS_MAX_IN];  // (= in_off[c] for a codeword buffer; c * shard_stride for shard-major stores, smr_rs_from_data_encode_stores)
    uint8_t *fan_dst[RS_MAX_IN + RS_MAX_OUT];   // shard fan-out fused in as well: shard k (k < n_in: data, else parity k - n_in) of codeword
    uint64_t fan_cw_stride;        // i ALSO to fan_dst[k] + i * fan_cw_stride where fan_dst[k] != NULL -- every holder's shard store (or its
    uint32_t fan_any;              // slice of a send buffer) filled by the pass that makes the shards (rspaxos/request.rs:127-142)
    uint64_t in_valid;             // bytes of a codeword's input that exist (beyond: zero)
    uint64_t in_bytes;             // bytes readable from in_base (end of the last codeword's input)
    uint64_t shard_len;
    uint64_t n_cw;
    uint32_t nblk;                 // ceil(shard_len / 16)
    int n_in, n_out;
    uint8_t coef[RS_MAX_OUT][RS_MAX_IN];
    uint8_t colmask[RS_MAX_IN];    // OR over r of coef[r][c]
    uint16_t plane[RS_MAX_OUT][8]; // plane[r][b]: bit c set iff bit b of coef[r][c] is set
    int8_t hib[RS_MAX_OUT];        // highest non-empty plane of row r (-1: all-zero row)
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// multiply 4 packed GF(2^8) bytes by 2 (poly 0x11D), no cross-byte carries
__device__ __forceinline__ uint32_t gf_xtime4(uint32_t x) {
    uint32_t hi = x & 0x80808080u;
    uint32_t lo = (x << 1) & 0xFEFEFEFEu;
    // per byte: 0x80 -> 0x7F -> & 0x1D ; 0 -> 0
    return lo ^ ((hi - (hi >> 7)) & 0x1D1D1D1Du);
}

__device__ __forceinline__ u32x4 load16(const uint8_t *p) {
    u32x4 v;
    __builtin_memcpy(&v, p, 16);   // align-1 global load: one dwordx4 on gfx950
    return v;
}
__device__ __forceinline__ void store16(uint8_t *p, u32x4 v) { __builtin_memcpy(p, &v, 16); }

// Load 16 byte-columns of input shard c; bytes at codeword offset >= in_valid
// are zero (fused from_data padding, rscoding.rs:188-189).  The block that straddles the end of
// the valid bytes is still ONE 16-byte load with the excess masked off, as long as the load stays
// inside the buffer: with shard_len = ceil(L / d) nearly every wavefront holds such a lane, and a
// byte-wise path there is run (divergently) by the whole wavefront.  Only the last few bytes of
// the whole buffer take the byte loop.
__device__ __forceinline__ u32x4 load_cols(const RsArgs &a, const uint8_t *cw, int c, uint64_t c0) {
    const uint64_t o = a.in_off[c] + c0;
    if (o + 16 <= a.in_valid) return load16(cw + o);
    const uint64_t room = (uint64_t)((a.in_base + a.in_bytes) - cw);   // readable bytes from this codeword's start
    const uint32_t nv = o >= a.in_valid ? 0u : (uint32_t)(a.in_valid - o);   // < 16 bytes of the window exist
    u32x4 v = {0u, 0u, 0u, 0u};
    if (nv == 0) return v;
    if (o + 16 <= room) {
        v = load16(cw + o);
        const uint32_t m0 = nv >= 4 ? 0xFFFFFFFFu : (1u << (8 * nv)) - 1u;
        const uint32_t m1 = nv >= 8 ? 0xFFFFFFFFu : (nv > 4 ? (1u << (8 * (nv - 4))) - 1u : 0u);
        const uint32_t m2 = nv >= 12 ? 0xFFFFFFFFu : (nv > 8 ? (1u << (8 * (nv - 8))) - 1u : 0u);
        const uint32_t m3 = nv > 12 ? (1u << (8 * (nv - 12))) - 1u : 0u;
        v.x &= m0; v.y &= m1; v.z &= m2; v.w &= m3;
        return v;
    }
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t byte = ((uint32_t)i < nv) ? cw[o + i] : 0u;
        if (i < 4) w0 |= byte << (8 * i); else if (i < 8) w1 |= byte << (8 * (i - 4));
        else if (i < 12) w2 |= byte << (8 * (i - 8)); else w3 |= byte << (8 * (i - 12));
    }
    return (u32x4){w0, w1, w2, w3};
}

// 16 byte columns of a shard from its column c0 on, to p; the shard's last block stores its n < 16 bytes as whole dwords
// plus at most three single bytes
__device__ __forceinline__ void store_block(uint8_t *p, uint64_t c0, uint64_t shard_len, u32x4 v) {
    if (c0 + 16 <= shard_len) { store16(p, v); return; }
    const uint32_t n = (uint32_t)(shard_len - c0);
    const uint32_t w0 = v.x, w1 = v.y, w2 = v.z, w3 = v.w;
#define SMR_ST_PART(k, w)                                                                  \
    if (n >= 4 * (k) + 4) __builtin_memcpy(p + 4 * (k), &w, 4);                          \
    else if (n > 4 * (k)) {                                                                \
        p[4 * (k)] = (uint8_t)(w);                                                         \
        if (n > 4 * (k) + 1) p[4 * (k) + 1] = (uint8_t)((w) >> 8);                         \
        if (n > 4 * (k) + 2) p[4 * (k) + 2] = (uint8_t)((w) >> 16);                        \
    }
    SMR_ST_PART(0, w0) SMR_ST_PART(1, w1) SMR_ST_PART(2, w2) SMR_ST_PART(3, w3)
#undef SMR_ST_PART
}
__device__ __forceinline__ void store_cols(const RsArgs &a, uint8_t *cw, int r, uint64_t c0, u32x4 v) {
    store_block(cw + a.out_off[r] + c0, c0, a.shard_len, v);
}

// acc[r] = XOR_c coef[r][c] * in_c for 16 byte columns, all input loads issued up front
// (COPY: the columns just loaded -- zero beyond the valid bytes -- are also stored as the codeword's data shards at copy_cw:
// RSCodeword::from_data's pad + split, rscoding.rs:188-200, at no extra read)
template <int NOUT, int NIN, bool COPY = false>
__device__ __forceinline__ void rs_product_xtime(const RsArgs &a, const uint8_t *cw, uint64_t c0, u32x4 (&acc)[NOUT],
                                                 uint8_t *copy_cw = nullptr, bool fan = false, uint64_t fan_off = 0) {
    u32x4 x[NIN];
#pragma unroll
    for (int c = 0; c < NIN; c++) x[c] = (c < a.n_in) ? load_cols(a, cw, c, c0) : (u32x4){0u, 0u, 0u, 0u};
    if (COPY) {
#pragma unroll
        for (int c = 0; c < NIN; c++)
            if (c < a.n_in) {
                store_block(copy_cw + a.copy_off[c] + c0, c0, a.shard_len, x[c]);
                if (fan && a.fan_dst[c]) store_block(a.fan_dst[c] + fan_off + c0, c0, a.shard_len, x[c]);
            }
    }
#pragma unroll
    for (int r = 0; r < NOUT; r++) {
        u32x4 s = {0u, 0u, 0u, 0u};
        const int hi = (r < a.n_out) ? a.hib[r] : -1;            // wave-uniform
#pragma unroll
        for (int b = 7; b >= 0; b--) {
            if (b > hi) continue;
            if (b < hi) { s.x = gf_xtime4(s.x); s.y = gf_xtime4(s.y); s.z = gf_xtime4(s.z); s.w = gf_xtime4(s.w); }
            const uint32_t pl = a.plane[r][b];
            if (pl == 0) continue;
#pragma unroll
            for (int c = 0; c < NIN; c++)
                if ((pl >> c) & 1u) s ^= x[c];                   // scalar branch
        }
        acc[r] = s;
    }
}

// this lane's (codeword, first byte column), from the flat thread index
__device__ __forceinline__ bool rs_locate(const RsArgs &a, uint64_t &cw_i, uint64_t &c0) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;          // host keeps n_cw * nblk below 2^32
    const uint32_t q = t / a.nblk;
    cw_i = q;
    c0 = (uint64_t)(t - q * a.nblk) * 16;
    return q < a.n_cw;
}

template <int NOUT, int NIN>
__global__ __launch_bounds__(256) void rs_matmul_xtime(const RsArgs a) {
    uint64_t cw_i, c0;
    if (!rs_locate(a, cw_i, c0)) return;
    u32x4 acc[NOUT];
    rs_product_xtime<NOUT, NIN>(a, a.in_base + cw_i * a.in_cw_stride, c0, acc);
    uint8_t *ocw = a.out_base + cw_i * a.out_cw_stride;
#pragma unroll
    for (int r = 0; r < NOUT; r++)
        if (r < a.n_out) store_cols(a, ocw, r, c0, acc[r]);
}

// from_data + compute_parity in one pass over the serialized bytes (rscoding.rs:165-243 + :447-486): read L, write d + p shards
template <int NOUT, int NIN>
__global__ __launch_bounds__(256) void rs_from_data_xtime(const RsArgs a) {
    uint64_t cw_i, c0;
    if (!rs_locate(a, cw_i, c0)) return;
    u32x4 acc[NOUT];
    const bool fan = a.fan_any != 0;
    const uint64_t fan_off = cw_i * a.fan_cw_stride;
    rs_product_xtime<NOUT, NIN, true>(a, a.in_base + cw_i * a.in_cw_stride, c0, acc, a.copy_base + cw_i * a.copy_cw_stride, fan, fan_off);
    uint8_t *ocw = a.out_base + cw_i * a.out_cw_stride;
#pragma unroll
    for (int r = 0; r < NOUT; r++)
        if (r < a.n_out) {
            store_cols(a, ocw, r, c0, acc[r]);
            if (fan && a.fan_dst[a.n_in + r]) store_block(a.fan_dst[a.n_in + r] + fan_off + c0, c0, a.shard_len, acc[r]);
        }
}

// LDS product-table variant: tab[(r * n_in + c) * 256 + v] = coef[r][c] * v
template <int NOUT>
__global__ __launch_bounds__(256) void rs_matmul_lut(const RsArgs a, const uint8_t *__restrict__ tabs) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) uint8_t, lds_tab)
    const int ntab = a.n_out * a.n_in * 256;
    for (int i = threadIdx.x * 16; i < ntab; i += 256 * 16)
        *reinterpret_cast<u32x4 *>(lds_tab + i) = *reinterpret_cast<const u32x4 *>(tabs + i);
    __syncthreads();
    uint64_t cw_i, c0;
    if (!rs_locate(a, cw_i, c0)) return;
    const uint8_t *cw = a.in_base + cw_i * a.in_cw_stride;
    u32x4 acc[NOUT];
#pragma unroll
    for (int r = 0; r < NOUT; r++) acc[r] = (u32x4){0u, 0u, 0u, 0u};
    for (int c = 0; c < a.n_in; c++) {
        u32x4 x = load_cols(a, cw, c, c0);
#pragma unroll
        for (int r = 0; r < NOUT; r++) {
            if (r >= a.n_out) break;
            const uint8_t *tb = lds_tab + (r * a.n_in + c) * 256;
            uint32_t w[4] = {x.x, x.y, x.z, x.w};
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                o[k] = (uint32_t)tb[w[k] & 0xFF] | ((uint32_t)tb[(w[k] >> 8) & 0xFF] << 8) |
                       ((uint32_t)tb[(w[k] >> 16) & 0xFF] << 16) | ((uint32_t)tb[w[k] >> 24] << 24);
            acc[r].x ^= o[0]; acc[r].y ^= o[1]; acc[r].z ^= o[2]; acc[r].w ^= o[3];
        }
    }
    uint8_t *ocw = a.out_base + cw_i * a.out_cw_stride;
#pragma unroll
    for (int r = 0; r < NOUT; r++)
        if (r < a.n_out) store_cols(a, ocw, r, c0, acc[r]);
}

// verify: recompute the n_out shards and compare with what is stored at out_off
template <int NOUT, int NIN>
__global__ __launch_bounds__(256) void rs_verify_xtime(const RsArgs a, uint8_t *ok) {
    uint64_t cw_i, c0;
    if (!rs_locate(a, cw_i, c0)) return;
    u32x4 acc[NOUT];
    rs_product_xtime<NOUT, NIN>(a, a.in_base + cw_i * a.in_cw_stride, c0, acc);
    bool bad = false;
#pragma unroll
    for (int r = 0; r < NOUT; r++) {
        if (r >= a.n_out) break;
        const uint8_t *p = a.out_base + cw_i * a.out_cw_stride + a.out_off[r] + c0;
        uint8_t tmp[16];
        __builtin_memcpy(tmp, &acc[r], 16);
        for (uint64_t i = 0; i < 16 && c0 + i < a.shard_len; i++) bad |= (p[i] != tmp[i]);
    }
    if (bad) ok[cw_i] = 0;
}

// ------------------------------------------------------------------ host ---
// GF(2^8) helpers for building coding / inversion matrices (tiny, host side).
struct Gf {
    uint8_t exp[512], log[256];
    Gf() {
        int x = 1;
        for (int i = 0; i < 255; i++) {
            exp[i] = (uint8_t)x; log[x] = (uint8_t)i;
            x <<= 1; if (x & 0x100) x ^= 0x11D;
        }
        for (int i = 255; i < 512; i++) exp[i] = exp[i - 255];
        log[0] = 0;
    }
    uint8_t mul(uint8_t a, uint8_t b) const { return (a && b) ? exp[log[a] + log[b]] : 0; }
    uint8_t div(uint8_t a, uint8_t b) const { return a ? exp[(log[a] + 255 - log[b]) % 255] : 0; }
    uint8_t pow(uint8_t a, int n) const {
        if (n == 0) return 1;
        if (a == 0) return 0;
        return exp[(log[a] * n) % 255];
    }
};
static const Gf &gf() { static Gf g; return g; }

// in-place Gauss-Jordan inverse of an n x n matrix (row major); false if singular
static bool gf_invert(uint8_t *m, int n) {
    const Gf &g = gf();
    uint8_t aug[RS_MAX_IN][2 * RS_MAX_IN] = {};
    for (int r = 0; r < n; r++) {
        for (int c = 0; c < n; c++) aug[r][c] = m[r * n + c];
        aug[r][n + r] = 1;
    }
    for (int c = 0; c < n; c++) {
        int piv = -1;
        for (int r = c; r < n; r++) if (aug[r][c]) { piv = r; break; }
        if (piv < 0) return false;
        if (piv != c) for (int k = 0; k < 2 * n; k++) { uint8_t t = aug[c][k]; aug[c][k] = aug[piv][k]; aug[piv][k] = t; }
        uint8_t d = aug[c][c];
        for (int k = 0; k < 2 * n; k++) aug[c][k] = g.div(aug[c][k], d);
        for (int r = 0; r < n; r++) {
            if (r == c || !aug[r][c]) continue;
            uint8_t f = aug[r][c];
            for (int k = 0; k < 2 * n; k++) aug[r][k] ^= g.mul(f, aug[c][k]);
        }
    }
    for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) m[r * n + c] = aug[r][n + c];
    return true;
}

// (d+p) x d systematic coding matrix: Vandermonde(r^c) * inverse(top d x d)
static bool rs_build_matrix(int d, int p, uint8_t *out) {
    if (d <= 0 || d > RS_MAX_IN || p < 0 || d + p > 256) return false;
    const Gf &g = gf();
    int t = d + p;
    std::string vbuf((size_t)t * d, '\0');
    uint8_t *v = reinterpret_cast<uint8_t *>(&vbuf[0]);
    for (int r = 0; r < t; r++) for (int c = 0; c < d; c++) v[r * d + c] = g.pow((uint8_t)r, c);
    uint8_t top[RS_MAX_IN * RS_MAX_IN];
    for (int i = 0; i < d * d; i++) top[i] = v[i];
    if (!gf_invert(top, d)) return false;
    for (int r = 0; r < t; r++)
        for (int c = 0; c < d; c++) {
            uint8_t acc = 0;
            for (int k = 0; k < d; k++) acc ^= g.mul(v[r * d + k], top[k * d + c]);
            out[r * d + c] = acc;
        }
    return true;
}

static void rs_finish_args(RsArgs &a) {
    a.nblk = (uint32_t)((a.shard_len + 15) / 16);
    for (int c = 0; c < RS_MAX_IN; c++) {
        uint8_t m = 0;
        for (int r = 0; r < a.n_out; r++) m |= a.coef[r][c];
        a.colmask[c] = (c < a.n_in) ? m : 0;
    }
    for (int r = 0; r < RS_MAX_OUT; r++) {
        a.hib[r] = -1;
        for (int b = 0; b < 8; b++) {
            uint16_t pl = 0;
            if (r < a.n_out)
                for (int c = 0; c < a.n_in; c++) pl |= (uint16_t)(((a.coef[r][c] >> b) & 1) << c);
            a.plane[r][b] = pl;
            if (pl) a.hib[r] = (int8_t)b;
        }
    }
}

// kernel instance by (outputs, inputs) bucket: the input columns live in registers
#define RS_DISPATCH(K, a, ...)                                                                     \
    do {                                                                                           \
        const int no_ = (a).n_out <= 2 ? 2 : (a).n_out <= 4 ? 4 : 8;                               \
        const int ni_ = (a).n_in <= 4 ? 4 : (a).n_in <= 8 ? 8 : 16;                                \
        if (no_ == 2 && ni_ == 4) hipLaunchKernelGGL((K<2, 4>), __VA_ARGS__);                      \
        else if (no_ == 2 && ni_ == 8) hipLaunchKernelGGL((K<2, 8>), __VA_ARGS__);                 \
        else if (no_ == 2) hipLaunchKernelGGL((K<2, 16>), __VA_ARGS__);                            \
        else if (no_ == 4 && ni_ == 4) hipLaunchKernelGGL((K<4, 4>), __VA_ARGS__);                 \
        else if (no_ == 4 && ni_ == 8) hipLaunchKernelGGL((K<4, 8>), __VA_ARGS__);                 \
        else if (no_ == 4) hipLaunchKernelGGL((K<4, 16>), __VA_ARGS__);                            \
        else if (ni_ == 4) hipLaunchKernelGGL((K<8, 4>), __VA_ARGS__);                             \
        else if (ni_ == 8) hipLaunchKernelGGL((K<8, 8>), __VA_ARGS__);                             \
        else hipLaunchKernelGGL((K<8, 16>), __VA_ARGS__);                                          \
    } while (0)

// Product tables of the LUT variant: one IMMUTABLE device buffer per coefficient matrix, built once under a
// mutex and never overwritten or freed, so calls with different schemes / erasure patterns on different
// streams or threads cannot disturb each other's tables (and no call waits for a stream to drain).
static std::mutex g_lut_mu;
static std::map<std::string, uint8_t *> g_lut_cache;

static int rs_lut_table(const RsArgs &a, uint8_t **out) {
    std::string key;
    key.push_back((char)a.n_out); key.push_back((char)a.n_in);
    for (int r = 0; r < a.n_out; r++) for (int c = 0; c < a.n_in; c++) key.push_back((char)a.coef[r][c]);
    std::lock_guard<std::mutex> lk(g_lut_mu);
    auto it = g_lut_cache.find(key);
    if (it != g_lut_cache.end()) { *out = it->second; return SMR_OK; }
    const Gf &g = gf();
    const size_t ntab = (size_t)a.n_out * a.n_in * 256;
    std::string tb(ntab, '\0');
    for (int r = 0; r < a.n_out; r++)
        for (int c = 0; c < a.n_in; c++)
            for (int v = 0; v < 256; v++)
                tb[((size_t)r * a.n_in + c) * 256 + v] = (char)g.mul(a.coef[r][c], (uint8_t)v);
    uint8_t *dev = nullptr;
    SMR_HIP_TRY(hipMalloc((void **)&dev, ntab));
    SMR_HIP_TRY(hipMemcpy(dev, tb.data(), ntab, hipMemcpyHostToDevice));       // blocking, once per matrix
    g_lut_cache.emplace(std::move(key), dev);
    *out = dev;
    return SMR_OK;
}

template <bool LUT>
static int rs_launch(RsArgs &a, hipStream_t st) {
    rs_finish_args(a);
    uint64_t threads = a.n_cw * a.nblk;
    if (threads == 0) return SMR_OK;
    uint64_t blocks = (threads + 255) / 256;
    if (blocks > 0xFFFFFFull) return fail(SMR_ERR_ARG, "rs: too many codewords for one launch");
    dim3 grid((unsigned)blocks), block(256);
    if (LUT) {
        const size_t ntab = (size_t)a.n_out * a.n_in * 256;
        uint8_t *lut = nullptr;
        if (int rc = rs_lut_table(a, &lut)) return rc;
        if (a.n_out <= 2) hipLaunchKernelGGL(rs_matmul_lut<2>, grid, block, ntab, st, a, lut);
        else if (a.n_out <= 4) hipLaunchKernelGGL(rs_matmul_lut<4>, grid, block, ntab, st, a, lut);
        else hipLaunchKernelGGL(rs_matmul_lut<8>, grid, block, ntab, st, a, lut);
    } else if (a.copy_base) {
        RS_DISPATCH(rs_from_data_xtime, a, grid, block, 0, st, a);
    } else {
        RS_DISPATCH(rs_matmul_xtime, a, grid, block, 0, st, a);
    }
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

template <bool LUT>
static int rs_encode_impl(const uint8_t *data, uint64_t data_len, uint64_t cw_stride, uint64_t n_cw, int d,
                          int p, uint8_t *parity, uint64_t par_stride, uint64_t par_shard_stride, void *stream,
                          uint8_t *copy_base = nullptr, uint64_t copy_cw_stride = 0, uint8_t *const *fan_dst = nullptr,
                          uint64_t fan_cw_stride = 0, uint64_t copy_shard_stride = 0) {
    if (d <= 0) return fail(SMR_ERR_ARG, "num_data_shards is zero");          // rscoding.rs:172-174
    if (data_len == 0) return fail(SMR_ERR_ARG, "codeword is null");          // rscoding.rs:451-453
    if (p == 0 && !copy_base) return SMR_OK;                                   // rscoding.rs:454-456
    if (d > RS_MAX_IN || p > RS_MAX_OUT) return fail(SMR_ERR_ARG, "rs: scheme exceeds d<=16, p<=8");
    if (!data || !parity) return fail(SMR_ERR_ARG, "rs: null buffer");
    uint8_t m[(RS_MAX_IN + RS_MAX_OUT) * RS_MAX_IN];
    if (!rs_build_matrix(d, p, m)) return fail(SMR_ERR_ARG, "rs: cannot build coding matrix");
    RsArgs a = {};
    a.in_base = data; a.out_base = parity;
    a.in_cw_stride = cw_stride; a.out_cw_stride = par_stride;
    a.copy_base = copy_base; a.copy_cw_stride = copy_cw_stride;
    a.fan_cw_stride = fan_cw_stride;
    for (int k = 0; fan_dst && k < d + p; k++) { a.fan_dst[k] = fan_dst[k]; if (fan_dst[k]) a.fan_any = 1; }
    a.shard_len = smr_rs_shard_len(data_len, d);
    a.in_valid = data_len;
    a.in_bytes = n_cw ? (n_cw - 1) * cw_stride + data_len : 0;   // rows may be packed tightly (cw_stride == data_len)
    a.n_cw = n_cw; a.n_in = d; a.n_out = p;
    if (par_shard_stride < a.shard_len) return fail(SMR_ERR_ARG, "rs: par_shard_stride < shard_len");
    for (int c = 0; c < d; c++) { a.in_off[c] = (uint64_t)c * a.shard_len; a.copy_off[c] = (uint64_t)c * (copy_shard_stride ? copy_shard_stride : a.shard_len); }
    for (int r = 0; r < p; r++) {
        a.out_off[r] = (uint64_t)r * par_shard_stride;
        for (int c = 0; c < d; c++) a.coef[r][c] = m[(d + r) * d + c];
    }
    return rs_launch<LUT>(a, (hipStream_t)stream);
}

}  // namespace smr

using namespace smr;

extern "C" {

int smr_rs_matrix(int d, int p, uint8_t *out_host) {
    if (!out_host || !rs_build_matrix(d, p, out_host)) return fail(SMR_ERR_ARG, "rs: bad scheme");
    return SMR_OK;
}

uint64_t smr_rs_shard_len(uint64_t data_len, int d) {
    if (d <= 0) return 0;
    return (data_len % (uint64_t)d == 0) ? data_len / d : data_len / d + 1;
}

int smr_rs_encode(const uint8_t *data_dev, uint64_t data_len, uint64_t cw_stride, uint64_t n_cw, int d, int p,
                  uint8_t *parity_dev, uint64_t par_stride, uint64_t par_shard_stride, void *stream) {
    return rs_encode_impl<false>(data_dev, data_len, cw_stride, n_cw, d, p, parity_dev, par_stride,
                                 par_shard_stride, stream);
}

int smr_rs_from_data_encode_scatter(const uint8_t *src_dev, uint64_t data_len, uint64_t src_stride, uint64_t n_cw, int d, int p,
                                    uint8_t *cw_dev, uint64_t cw_stride, uint8_t *const *shard_dst, uint64_t dst_cw_stride, void *stream) {
    if (!cw_dev) return fail(SMR_ERR_ARG, "rs: null buffer");
    const uint64_t sl = smr_rs_shard_len(data_len, d);
    if (d > 0 && cw_stride < (uint64_t)(d + p) * sl) return fail(SMR_ERR_ARG, "rs: cw_stride < (d + p) * shard_len");
    if (src_dev && src_dev < cw_dev + n_cw * cw_stride && cw_dev < src_dev + n_cw * src_stride)
        return fail(SMR_ERR_ARG, "rs: source and codeword buffers overlap");
    if (shard_dst && dst_cw_stride < sl) return fail(SMR_ERR_ARG, "rs: fan-out stride below shard_len");
    return rs_encode_impl<false>(src_dev, data_len, src_stride, n_cw, d, p, cw_dev + (uint64_t)(d > 0 ? d : 0) * sl, cw_stride, sl, stream,
                                 cw_dev, cw_stride, shard_dst, dst_cw_stride);
}

int smr_rs_from_data_encode_fanout(const uint8_t *src_dev, uint64_t data_len, uint64_t src_stride, uint64_t n_cw, int d, int p,
                                   uint8_t *cw_dev, uint64_t cw_stride, uint8_t *fan_dev, uint64_t fan_shard_stride,
                                   uint64_t fan_cw_stride, uint32_t fan_mask, void *stream) {
    uint8_t *dst[RS_MAX_IN + RS_MAX_OUT] = {};
    if (fan_mask) {
        const uint64_t sl = smr_rs_shard_len(data_len, d);
        if (!fan_dev) return fail(SMR_ERR_ARG, "rs: fan-out mask without a buffer");
        if (d <= 0 || d > RS_MAX_IN || p < 0 || p > RS_MAX_OUT) return fail(SMR_ERR_ARG, d <= 0 ? "num_data_shards is zero" : "rs: scheme exceeds d<=16, p<=8");
        if (fan_mask >> (d + p)) return fail(SMR_ERR_ARG, "rs: fan-out mask names a shard beyond d + p");
        if (fan_cw_stride < sl || (n_cw && fan_shard_stride < (n_cw - 1) * fan_cw_stride + sl))
            return fail(SMR_ERR_ARG, "rs: fan-out strides: a store holds n_cw shards of shard_len bytes, fan_cw_stride apart");
        for (int k = 0; k < d + p; k++) dst[k] = ((fan_mask >> k) & 1u) ? fan_dev + (uint64_t)k * fan_shard_stride : nullptr;
    }
    return smr_rs_from_data_encode_scatter(src_dev, data_len, src_stride, n_cw, d, p, cw_dev, cw_stride, fan_mask ? dst : nullptr, fan_cw_stride, stream);
}

int smr_rs_from_data_encode_stores(const uint8_t *src_dev, uint64_t data_len, uint64_t src_stride, uint64_t n_cw, int d, int p,
                                   uint8_t *stores_dev, uint64_t shard_stride, uint64_t cw_stride, void *stream) {
    if (!stores_dev) return fail(SMR_ERR_ARG, "rs: null buffer");
    if (d <= 0) return fail(SMR_ERR_ARG, "num_data_shards is zero");
    if (d > RS_MAX_IN || p < 0 || p > RS_MAX_OUT) return fail(SMR_ERR_ARG, "rs: scheme exceeds d<=16, p<=8");
    const uint64_t sl = smr_rs_shard_len(data_len, d);
    if (cw_stride < sl || (n_cw && shard_stride < (n_cw - 1) * cw_stride + sl))
        return fail(SMR_ERR_ARG, "rs: store strides: a store holds n_cw shards of shard_len bytes, cw_stride apart");
    const uint64_t span = (uint64_t)(d + p - 1) * shard_stride + (n_cw ? (n_cw - 1) * cw_stride + sl : 0);
    if (src_dev && src_dev < stores_dev + span && stores_dev < src_dev + n_cw * src_stride)
        return fail(SMR_ERR_ARG, "rs: source and shard stores overlap");
    // shard k of codeword i at stores + k * shard_stride + i * cw_stride: the data shards through the from_data copy, the parity
    // shards through the product's own stores -- every shard written ONCE, straight into its holder's store
    return rs_encode_impl<false>(src_dev, data_len, src_stride, n_cw, d, p, stores_dev + (uint64_t)d * shard_stride, cw_stride, shard_stride, stream,
                                 stores_dev, cw_stride, nullptr, 0, shard_stride);
}

int smr_rs_from_data_encode(const uint8_t *src_dev, uint64_t data_len, uint64_t src_stride, uint64_t n_cw, int d, int p,
                            uint8_t *cw_dev, uint64_t cw_stride, void *stream) {
    return smr_rs_from_data_encode_fanout(src_dev, data_len, src_stride, n_cw, d, p, cw_dev, cw_stride, nullptr, 0, 0, 0, stream);
}

int smr_rs_encode_lut(const uint8_t *data_dev, uint64_t data_len, uint64_t cw_stride, uint64_t n_cw, int d,
                      int p, uint8_t *parity_dev, uint64_t par_stride, uint64_t par_shard_stride,
                      void *stream) {
    return rs_encode_impl<true>(data_dev, data_len, cw_stride, n_cw, d, p, parity_dev, par_stride,
                                par_shard_stride, stream);
}

int smr_rs_reconstruct(uint8_t *shards_dev, uint64_t shard_len, uint64_t shard_stride, uint64_t cw_stride,
                       uint64_t n_cw, int d, int p, uint32_t present_mask, int data_only, void *stream) {
    if (d <= 0 || d > RS_MAX_IN || p < 0 || p > RS_MAX_OUT) return fail(SMR_ERR_ARG, "rs: bad scheme");
    if (shard_len == 0) return fail(SMR_ERR_ARG, "codeword is null");
    int t = d + p, have = 0;
    for (int k = 0; k < t; k++) have += (present_mask >> k) & 1;
    if (have < d) return fail(SMR_ERR_ARG, "too few shards present");
    if (have == t) return SMR_OK;
    uint8_t m[(RS_MAX_IN + RS_MAX_OUT) * RS_MAX_IN];
    if (!rs_build_matrix(d, p, m)) return fail(SMR_ERR_ARG, "rs: cannot build coding matrix");
    // first d present shards -> sub-matrix -> inverse (upstream reconstruct)
    int src[RS_MAX_IN]; uint8_t inv[RS_MAX_IN * RS_MAX_IN];
    int n = 0;
    for (int k = 0; k < t && n < d; k++)
        if ((present_mask >> k) & 1) { for (int c = 0; c < d; c++) inv[n * d + c] = m[k * d + c]; src[n++] = k; }
    if (!gf_invert(inv, d)) return fail(SMR_ERR_ARG, "rs: singular sub-matrix");
    const Gf &g = gf();
    RsArgs a = {};
    a.in_base = shards_dev; a.out_base = shards_dev;
    a.in_cw_stride = cw_stride; a.out_cw_stride = cw_stride;
    a.shard_len = shard_len; a.in_valid = ~0ull >> 1; a.n_cw = n_cw; a.n_in = d;
    a.in_bytes = n_cw ? (n_cw - 1) * cw_stride + (uint64_t)(t - 1) * shard_stride + shard_len : 0;
    for (int c = 0; c < d; c++) a.in_off[c] = (uint64_t)src[c] * shard_stride;
    int n_out = 0;
    for (int k = 0; k < t; k++) {
        if ((present_mask >> k) & 1) continue;
        if (k >= d && data_only) continue;
        if (n_out == RS_MAX_OUT) return fail(SMR_ERR_ARG, "rs: too many missing shards");
        a.out_off[n_out] = (uint64_t)k * shard_stride;
        for (int c = 0; c < d; c++) {
            if (k < d) a.coef[n_out][c] = inv[k * d + c];
            else {  // parity row composed with the data-recovery matrix
                uint8_t acc = 0;
                for (int j = 0; j < d; j++) acc ^= g.mul(m[k * d + j], inv[j * d + c]);
                a.coef[n_out][c] = acc;
            }
        }
        n_out++;
    }
    a.n_out = n_out;
    if (n_out == 0) return SMR_OK;
    return rs_launch<false>(a, (hipStream_t)stream);
}

int smr_rs_verify(const uint8_t *shards_dev, uint64_t shard_len, uint64_t shard_stride, uint64_t cw_stride,
                  uint64_t n_cw, int d, int p, uint8_t *ok_dev, void *stream) {
    if (d <= 0 || d > RS_MAX_IN || p < 0 || p > RS_MAX_OUT) return fail(SMR_ERR_ARG, "rs: bad scheme");
    if (shard_len == 0) return fail(SMR_ERR_ARG, "codeword is null");
    hipStream_t st = (hipStream_t)stream;
    SMR_HIP_TRY(hipMemsetAsync(ok_dev, 1, n_cw, st));
    if (p == 0 || n_cw == 0) return SMR_OK;
    uint8_t m[(RS_MAX_IN + RS_MAX_OUT) * RS_MAX_IN];
    if (!rs_build_matrix(d, p, m)) return fail(SMR_ERR_ARG, "rs: cannot build coding matrix");
    RsArgs a = {};
    a.in_base = shards_dev; a.out_base = const_cast<uint8_t *>(shards_dev);
    a.in_cw_stride = cw_stride; a.out_cw_stride = cw_stride;
    a.shard_len = shard_len; a.in_valid = ~0ull >> 1; a.n_cw = n_cw; a.n_in = d; a.n_out = p;
    a.in_bytes = n_cw ? (n_cw - 1) * cw_stride + (uint64_t)(d + p - 1) * shard_stride + shard_len : 0;
    for (int c = 0; c < d; c++) a.in_off[c] = (uint64_t)c * shard_stride;
    for (int r = 0; r < p; r++) {
        a.out_off[r] = (uint64_t)(d + r) * shard_stride;
        for (int c = 0; c < d; c++) a.coef[r][c] = m[(d + r) * d + c];
    }
    rs_finish_args(a);
    uint64_t blocks = (a.n_cw * a.nblk + 255) / 256;
    dim3 grid((unsigned)blocks), block(256);
    if (blocks > 0xFFFFFFull) return fail(SMR_ERR_ARG, "rs: too many codewords for one launch");
    RS_DISPATCH(rs_verify_xtime, a, grid, block, 0, st, a, ok_dev);
    SMR_HIP_TRY(hipGetLastError());
    return SMR_OK;
}

}  // extern "C"
