/* CPU ORACLE -- test infrastructure only (never linked into or imported by summerset_amd/).
 *
 * `Bitmap` (src/utils/bitmap.rs:14-133): "compact bitmap for u8 ID -> bool mapping", a FixedBitSet of `size` bits.
 * The engine and the other oracles hold the protocol's bitmaps (accept_acks, prepare_acks, rq_acks, avail_shards_map,
 * peer_alive) as plain integer masks, bit i = replica / shard i; this file restates the type itself so that the
 * reference's own unit tests (bitmap.rs:312-387) can be run against the restatement, and the mask convention checked
 * against it (tests/test_oracle_bitmap.py).  Sizes up to 64 (ReplicaId is u8; populations and shard counts are <= 24).
 * Return codes: 0 = Ok(()), -1 = Err(SummersetError) as the cited line raises it. */
#include <stdint.h>

typedef struct { uint8_t size; uint64_t bits; } orc_bitmap;

/* bitmap.rs:60-71 `new`: size 0 panics ("invalid bitmap size 0") */
int orc_bitmap_new(orc_bitmap *m, uint8_t size, int ones) {
    if (size == 0 || size > 64) return -2;                               /* :62 assert!(size != 0) */
    m->size = size;
    m->bits = ones ? (size == 64 ? ~0ull : ((1ull << size) - 1ull)) : 0ull;   /* :66-68 set_range(.., true) */
    return 0;
}
/* :74-85 */
int orc_bitmap_set(orc_bitmap *m, uint8_t idx, int flag) {
    if (idx >= m->size) return -1;                                       /* :76-81 "index {} out of bound" */
    if (flag) m->bits |= 1ull << idx; else m->bits &= ~(1ull << idx);
    return 0;
}
/* :88-97: *out = the flag */
int orc_bitmap_get(const orc_bitmap *m, uint8_t idx, int *out) {
    if (idx >= m->size) return -1;
    *out = (int)((m->bits >> idx) & 1ull);
    return 0;
}
uint8_t orc_bitmap_size(const orc_bitmap *m) { return m->size; }         /* :100-105 */
uint8_t orc_bitmap_count(const orc_bitmap *m) { return (uint8_t)__builtin_popcountll(m->bits); }   /* :108-113 */
void orc_bitmap_flip(orc_bitmap *m) {                                     /* :116-119 toggle_range(..) */
    m->bits = ~m->bits & (m->size == 64 ? ~0ull : ((1ull << m->size) - 1ull));
}
int orc_bitmap_union(orc_bitmap *m, const orc_bitmap *o) {               /* :122-134 */
    if (m->size != o->size) return -1;                                   /* "unioning sizes mismatch" */
    m->bits |= o->bits;
    return 0;
}
void orc_bitmap_clear(orc_bitmap *m) { m->bits = 0; }                    /* :137-140 */
/* From<(u8, Vec<u8>)> (:156-166) and the other From impls: size + the indexes that are true; an index out of bound
 * is the `unwrap()` panic of :162 -> -2 */
int orc_bitmap_from(orc_bitmap *m, uint8_t size, const uint8_t *ones, uint32_t n) {
    if (orc_bitmap_new(m, size, 0)) return -2;
    for (uint32_t i = 0; i < n; i++)
        if (orc_bitmap_set(m, ones[i], 1)) return -2;
    return 0;
}
/* From<Bitmap> for Vec<u8> (:214-226): the indexes that are true, ascending; returns their number */
uint32_t orc_bitmap_to_vec(const orc_bitmap *m, uint8_t *out) {
    uint32_t n = 0;
    for (uint8_t i = 0; i < m->size; i++)                                /* BitmapIter (:268-290): idx 0 .. size */
        if ((m->bits >> i) & 1ull) out[n++] = i;
    return n;
}
/* the integer mask the engine keeps for the same bitmap: bit i = id i */
uint64_t orc_bitmap_mask(const orc_bitmap *m) { return m->bits; }
/* Encode (:19-29): usize bit length, then the backing blocks as a usize slice -- bincode standard: varint(len),
 * varint(number of blocks), varint(block)...  Block width of fixedbitset 0.5 on a 64-bit target: usize = 64 bits
 * (unpinned: the crate is not vendored).  Returns the number of bytes written (cap >= 32 suffices here). */
static uint32_t put_varint(uint8_t *p, uint64_t v) {
    if (v < 251) { p[0] = (uint8_t)v; return 1; }
    if (v < (1ull << 16)) { p[0] = 0xFB; p[1] = (uint8_t)v; p[2] = (uint8_t)(v >> 8); return 3; }
    if (v < (1ull << 32)) { p[0] = 0xFC; for (int i = 0; i < 4; i++) p[1 + i] = (uint8_t)(v >> (8 * i)); return 5; }
    p[0] = 0xFD; for (int i = 0; i < 8; i++) p[1 + i] = (uint8_t)(v >> (8 * i)); return 9;
}
uint32_t orc_bitmap_bincode(const orc_bitmap *m, uint8_t *out) {
    uint32_t n = put_varint(out, m->size);
    n += put_varint(out + n, 1);                                          /* ceil(size / 64) blocks, size <= 64 */
    n += put_varint(out + n, m->bits);
    return n;
}
