/*
 * oracle/rs_oracle.c -- CPU restatement of Summerset's Reed-Solomon path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the shipped
 * product path; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / CPU
 * baseline.
 *
 * What it restates
 *   - RSCodeword::internal_new shard geometry  (src/utils/rscoding.rs:165-220:
 *     shard_len :177-181, zero padding :188-189, contiguous split :192-200)
 *   - RSCodeword::compute_parity               (src/utils/rscoding.rs:447-486)
 *   - RSCodeword::reconstruct{,_data}          (src/utils/rscoding.rs:490-537)
 *   - RSCodeword::verify_parity                (src/utils/rscoding.rs:541-577)
 *   - the GF(2^8) arithmetic those calls delegate to: third-party crate
 *     `reed-solomon-erasure = "6.0"` (Cargo.toml:45), type
 *     galois_8::ReedSolomon.  The crate is NOT vendored under /root/reference
 *     and there is no Cargo.lock, so its algorithm is restated here from its
 *     published construction (shared with Backblaze JavaReedSolomon and
 *     klauspost/reedsolomon): field GF(2^8) with generating polynomial 0x11D,
 *     generator 2; coding matrix = Vandermonde(rows = d+p, cols = d,
 *     V[r][c] = r^c) times the inverse of its top d x d block, so the top is
 *     the identity (systematic code); parity rows are rows d..d+p-1.
 *
 * PARITY STATUS: byte-level parity is UNPINNED BY THE REFERENCE -- its own
 * tests (rscoding.rs:788-876) only check round trips.  This restatement is
 * pinned instead against the upstream-family known-answer vectors (Galois
 * multiply/exp answers and the 5+5 "one encode" vector, see
 * tests/test_oracle_rs.py), i.e. validated against the published
 * construction, not against the crate binary.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GF_POLY 0x11D

static uint8_t gf_exp_tab[512];
static uint8_t gf_log_tab[256];
static int gf_ready = 0;

static void gf_init(void) {
    if (gf_ready) return;
    int x = 1;
    for (int i = 0; i < 255; i++) {
        gf_exp_tab[i] = (uint8_t)x;
        gf_log_tab[x] = (uint8_t)i;
        x <<= 1;
        if (x & 0x100) x ^= GF_POLY;
    }
    for (int i = 255; i < 512; i++) gf_exp_tab[i] = gf_exp_tab[i - 255];
    gf_log_tab[0] = 0; /* undefined; never used for a zero operand */
    gf_ready = 1;
}

uint8_t orc_gf_mul(uint8_t a, uint8_t b) {
    gf_init();
    if (a == 0 || b == 0) return 0;
    return gf_exp_tab[gf_log_tab[a] + gf_log_tab[b]];
}

uint8_t orc_gf_div(uint8_t a, uint8_t b) {
    gf_init();
    if (a == 0) return 0;
    /* b == 0 is a caller error; mirror "divide by zero" with 0 */
    if (b == 0) return 0;
    int d = (int)gf_log_tab[a] - (int)gf_log_tab[b];
    if (d < 0) d += 255;
    return gf_exp_tab[d];
}

/* a ** n in the field (upstream `galois::exp`): 0**0 == 1, 0**n == 0 */
uint8_t orc_gf_exp(uint8_t a, int n) {
    gf_init();
    if (n == 0) return 1;
    if (a == 0) return 0;
    int l = (gf_log_tab[a] * n) % 255;
    return gf_exp_tab[l];
}

void orc_gf_tables(uint8_t *exp_out /*256*/, uint8_t *log_out /*256*/) {
    gf_init();
    memcpy(exp_out, gf_exp_tab, 256);
    memcpy(log_out, gf_log_tab, 256);
}

/* ---- small dense matrices over GF(2^8), row major ---------------------- */

static void mat_mul(const uint8_t *a, int ar, int ac, const uint8_t *b, int bc,
                    uint8_t *out) {
    for (int r = 0; r < ar; r++)
        for (int c = 0; c < bc; c++) {
            uint8_t v = 0;
            for (int k = 0; k < ac; k++)
                v ^= orc_gf_mul(a[r * ac + k], b[k * bc + c]);
            out[r * bc + c] = v;
        }
}

/* Gauss-Jordan inverse of an n x n matrix; returns 0 on success, -1 if
 * singular. */
static int mat_inv(const uint8_t *m, int n, uint8_t *out) {
    int w = 2 * n;
    uint8_t *a = (uint8_t *)calloc((size_t)n * w, 1);
    for (int r = 0; r < n; r++) {
        memcpy(a + r * w, m + r * n, n);
        a[r * w + n + r] = 1;
    }
    for (int c = 0; c < n; c++) {
        int piv = -1;
        for (int r = c; r < n; r++)
            if (a[r * w + c]) { piv = r; break; }
        if (piv < 0) { free(a); return -1; }
        if (piv != c)
            for (int k = 0; k < w; k++) {
                uint8_t t = a[c * w + k];
                a[c * w + k] = a[piv * w + k];
                a[piv * w + k] = t;
            }
        uint8_t d = a[c * w + c];
        if (d != 1)
            for (int k = 0; k < w; k++) a[c * w + k] = orc_gf_div(a[c * w + k], d);
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            uint8_t f = a[r * w + c];
            if (!f) continue;
            for (int k = 0; k < w; k++)
                a[r * w + k] ^= orc_gf_mul(f, a[c * w + k]);
        }
    }
    for (int r = 0; r < n; r++) memcpy(out + r * n, a + r * w + n, n);
    free(a);
    return 0;
}

/* Full (d+p) x d coding matrix of the upstream construction. */
int orc_rs_matrix(int d, int p, uint8_t *out /* (d+p)*d */) {
    if (d <= 0 || p < 0 || d + p > 256) return -1;
    int t = d + p;
    uint8_t *v = (uint8_t *)malloc((size_t)t * d);
    uint8_t *top_inv = (uint8_t *)malloc((size_t)d * d);
    for (int r = 0; r < t; r++)
        for (int c = 0; c < d; c++) v[r * d + c] = orc_gf_exp((uint8_t)r, c);
    if (mat_inv(v, d, top_inv) != 0) { free(v); free(top_inv); return -1; }
    mat_mul(v, t, d, top_inv, d, out);
    free(v);
    free(top_inv);
    return 0;
}

/* shard_len rule of RSCodeword::internal_new (rscoding.rs:177-181). */
uint64_t orc_rs_shard_len(uint64_t data_len, int d) {
    if (d <= 0) return 0;
    return (data_len % (uint64_t)d == 0) ? data_len / d : data_len / d + 1;
}

/*
 * from_data geometry + compute_parity for ONE codeword.
 *   data[0..data_len)  : the serialized bytes (bincode output in the reference)
 *   parity             : p * shard_len bytes, parity shard k at k*shard_len
 * Returns 0, or -1 for the reference's error cases (null codeword, d == 0).
 */
int orc_rs_encode(int d, int p, const uint8_t *data, uint64_t data_len,
                  uint8_t *parity) {
    if (d <= 0) return -1;           /* "num_data_shards is zero" */
    if (data_len == 0) return -1;    /* "codeword is null" (rscoding.rs:451) */
    if (p == 0) return 0;
    uint64_t sl = orc_rs_shard_len(data_len, d);
    uint8_t *m = (uint8_t *)malloc((size_t)(d + p) * d);
    if (orc_rs_matrix(d, p, m) != 0) { free(m); return -1; }
    for (int k = 0; k < p; k++) {
        const uint8_t *row = m + (size_t)(d + k) * d;
        for (uint64_t i = 0; i < sl; i++) {
            uint8_t acc = 0;
            for (int c = 0; c < d; c++) {
                uint64_t off = (uint64_t)c * sl + i;
                uint8_t b = off < data_len ? data[off] : 0; /* zero padding */
                acc ^= orc_gf_mul(row[c], b);
            }
            parity[(uint64_t)k * sl + i] = acc;
        }
    }
    free(m);
    return 0;
}

/* Batched form used as the CPU baseline: n codewords, fixed strides. */
int orc_rs_encode_batch(int d, int p, const uint8_t *data, uint64_t data_len,
                        uint64_t cw_stride, uint64_t n_cw, uint8_t *parity,
                        uint64_t par_stride) {
    if (d <= 0 || data_len == 0) return -1;
    if (p == 0) return 0;
    gf_init();
    uint64_t sl = orc_rs_shard_len(data_len, d);
    uint8_t *m = (uint8_t *)malloc((size_t)(d + p) * d);
    if (orc_rs_matrix(d, p, m) != 0) { free(m); return -1; }
    /* per-coefficient product tables, the way a table-driven galois_8 encoder
     * multiplies a whole slice by one matrix element */
    uint8_t *tabs = (uint8_t *)malloc((size_t)p * d * 256);
    for (int k = 0; k < p; k++)
        for (int c = 0; c < d; c++)
            for (int v = 0; v < 256; v++)
                tabs[((size_t)k * d + c) * 256 + v] =
                    orc_gf_mul(m[(size_t)(d + k) * d + c], (uint8_t)v);
    uint8_t *padded = (uint8_t *)calloc((size_t)d * sl, 1);
    for (uint64_t w = 0; w < n_cw; w++) {
        /* resize(padded_len, 0) of rscoding.rs:188-189 */
        memcpy(padded, data + w * cw_stride, data_len);
        memset(padded + data_len, 0, (size_t)(d * sl - data_len));
        for (int k = 0; k < p; k++) {
            uint8_t *out = parity + w * par_stride + (uint64_t)k * sl;
            memset(out, 0, sl); /* BytesMut::zeroed, rscoding.rs:475 */
            for (int c = 0; c < d; c++) {
                const uint8_t *t = tabs + ((size_t)k * d + c) * 256;
                const uint8_t *in = padded + (uint64_t)c * sl;
                for (uint64_t i = 0; i < sl; i++) out[i] ^= t[in[i]];
            }
        }
    }
    free(padded);
    free(tabs);
    free(m);
    return 0;
}

/*
 * reconstruct / reconstruct_data for ONE codeword.
 *   shards  : (d+p) * shard_len bytes, shard k at k*shard_len (in/out)
 *   present : d+p flags (in/out); missing shards' bytes are ignored on input
 *   data_only != 0 -> only data shards are rebuilt (reconstruct_data)
 * Returns 0; -1 if fewer than d shards are present ("too few shards").
 * Upstream algorithm: take the first d present shards, invert the matching
 * d x d sub-matrix, rebuild missing data shards, then re-encode missing
 * parity shards from the complete data.
 */
int orc_rs_reconstruct(int d, int p, uint8_t *shards, uint64_t shard_len,
                       uint8_t *present, int data_only) {
    int t = d + p;
    int have = 0;
    for (int i = 0; i < t; i++) have += present[i] ? 1 : 0;
    if (have < d) return -1;
    if (have == t) return 0;
    uint8_t *m = (uint8_t *)malloc((size_t)t * d);
    if (orc_rs_matrix(d, p, m) != 0) { free(m); return -1; }
    uint8_t *sub = (uint8_t *)malloc((size_t)d * d);
    uint8_t *inv = (uint8_t *)malloc((size_t)d * d);
    int *src = (int *)malloc(sizeof(int) * d);
    int n = 0;
    for (int i = 0; i < t && n < d; i++)
        if (present[i]) {
            memcpy(sub + (size_t)n * d, m + (size_t)i * d, d);
            src[n++] = i;
        }
    int rc = mat_inv(sub, d, inv);
    if (rc == 0) {
        for (int k = 0; k < d; k++) {
            if (present[k]) continue;
            uint8_t *out = shards + (uint64_t)k * shard_len;
            for (uint64_t i = 0; i < shard_len; i++) {
                uint8_t acc = 0;
                for (int c = 0; c < d; c++)
                    acc ^= orc_gf_mul(inv[(size_t)k * d + c],
                                      shards[(uint64_t)src[c] * shard_len + i]);
                out[i] = acc;
            }
        }
        for (int k = 0; k < d; k++) present[k] = 1;
        if (!data_only) {
            for (int k = d; k < t; k++) {
                if (present[k]) continue;
                uint8_t *out = shards + (uint64_t)k * shard_len;
                for (uint64_t i = 0; i < shard_len; i++) {
                    uint8_t acc = 0;
                    for (int c = 0; c < d; c++)
                        acc ^= orc_gf_mul(m[(size_t)k * d + c],
                                          shards[(uint64_t)c * shard_len + i]);
                    out[i] = acc;
                }
                present[k] = 1;
            }
        }
    }
    free(src); free(inv); free(sub); free(m);
    return rc;
}

/* verify_parity: 1 if parity shards match the data shards, else 0. */
int orc_rs_verify(int d, int p, const uint8_t *shards, uint64_t shard_len) {
    uint8_t *m = (uint8_t *)malloc((size_t)(d + p) * d);
    if (orc_rs_matrix(d, p, m) != 0) { free(m); return -1; }
    int ok = 1;
    for (int k = d; k < d + p && ok; k++)
        for (uint64_t i = 0; i < shard_len; i++) {
            uint8_t acc = 0;
            for (int c = 0; c < d; c++)
                acc ^= orc_gf_mul(m[(size_t)k * d + c],
                                  shards[(uint64_t)c * shard_len + i]);
            if (acc != shards[(uint64_t)k * shard_len + i]) { ok = 0; break; }
        }
    free(m);
    return ok;
}

/* ---- bincode 2.0 "standard" (varint, little endian) ----------------------
 * Third-party crate bincode = "2.0" (not vendored); layout restated from its
 * published spec: u < 251 -> 1 byte; < 2^16 -> 0xFB + u16 LE; < 2^32 -> 0xFC +
 * u32 LE; else 0xFD + u64 LE.  Determines the bytes RSCodeword::from_data
 * shards (rscoding.rs:229-234).  Unpinned by the reference beyond round trips.
 */
static uint64_t put_varint(uint8_t *out, uint64_t v) {
    if (v < 251) { out[0] = (uint8_t)v; return 1; }
    if (v < (1ull << 16)) {
        out[0] = 0xFB; out[1] = (uint8_t)v; out[2] = (uint8_t)(v >> 8);
        return 3;
    }
    if (v < (1ull << 32)) {
        out[0] = 0xFC;
        for (int i = 0; i < 4; i++) out[1 + i] = (uint8_t)(v >> (8 * i));
        return 5;
    }
    out[0] = 0xFD;
    for (int i = 0; i < 8; i++) out[1 + i] = (uint8_t)(v >> (8 * i));
    return 9;
}

/* bincode(String): varint(len) + bytes.  benches/rse_bench.rs:165-167 shards
 * exactly this for a `String` value.  Returns encoded length. */
uint64_t orc_bincode_string(const uint8_t *s, uint64_t len, uint8_t *out) {
    uint64_t n = put_varint(out, len);
    memcpy(out + n, s, len);
    return n + len;
}

/* bincode(ReqBatch) for a batch holding ONE Put:
 * Vec<(ClientId u64, ApiRequest::Req{id u64, cmd: Command::Put{key,value}})>
 * variant order: ApiRequest::Req = 0 (src/server/external.rs:33-54),
 * Command::Put = 1 (src/server/statemach.rs:21-27). */
uint64_t orc_bincode_reqbatch_put(uint64_t client, uint64_t req_id,
                                  const uint8_t *key, uint64_t klen,
                                  const uint8_t *val, uint64_t vlen,
                                  uint8_t *out) {
    uint64_t n = 0;
    n += put_varint(out + n, 1);       /* Vec len */
    n += put_varint(out + n, client);  /* ClientId */
    n += put_varint(out + n, 0);       /* ApiRequest::Req */
    n += put_varint(out + n, req_id);  /* RequestId */
    n += put_varint(out + n, 1);       /* Command::Put */
    n += put_varint(out + n, klen);
    memcpy(out + n, key, klen); n += klen;
    n += put_varint(out + n, vlen);
    memcpy(out + n, val, vlen); n += vlen;
    return n;
}
