// Device-side data layout of the MultiPaxos / RSPaxos engine.
//
// One `MpRep` = the batched counterpart of the reference's
// `MultiPaxosReplica` struct (src/protocols/multipaxos/mod.rs:387-514) for ONE
// replica id over G groups.  Everything is structure-of-arrays with the group
// index fastest, so a wavefront touching lane-consecutive groups issues one
// contiguous 256/512-byte request per field.
//
//   per-group scalars   X[g]
//   slot ring           X[tix(W, slot & (W-1), g)]     (Vec<Instance>, mod.rs:426)
//   outbox (x2 parity)  X[tix(cap, j, g)], j < cap     (bcast_msg of Prepare /
//                                                       Accept / Heartbeat)
//   ack matrix          ack[tix(cap*R, j*R + r, g)]    u8: AcceptReply (its ballot = the Accept's) of
//                                                       replica r to my j-th
//                                                       outbox entry (0 = none)
//   with tix(rows, row, g) = ((g/64)*rows + row)*64 + g%64   (wave-tiled, see below)
//
// Instance fields are packed so the steady-state path touches 16 bytes per
// (replica, slot): s_bal (8) + s_val (4) + s_meta (4).  Rarely-used fields live
// in side arrays that are only valid when the matching meta flag says so.
#pragma once
#include <stdint.h>

#include "../../include/summerset_hip.h"

// Pointers that live inside a struct in memory would be "flat" to the compiler
// (flat_load / flat_store count against BOTH vmcnt and lgkmcnt, so every scalar
// pointer reload then waits for all outstanding stores).  On the device pass
// they are typed as global address space; the host sees plain pointers of the
// same size.
#if defined(__HIP_DEVICE_COMPILE__)
#define SMR_G __attribute__((address_space(1)))
#else
#define SMR_G
#endif

namespace smr {

// s_meta bit layout
constexpr uint32_t M_STATUS = 0x7u;        // Status, mod.rs:168-174
constexpr uint32_t M_EXT = 1u << 3;        // Instance::external
constexpr uint32_t M_LBK = 1u << 4;        // leader_bk is Some
constexpr uint32_t M_LBKX = 1u << 5;       // leader_bk.{trigger,endprep}_slot / prepare_max_bal in side arrays (else 0)
constexpr uint32_t M_RBK = 1u << 6;        // replica_bk is Some
constexpr uint32_t M_RBKX = 1u << 7;       // replica_bk.{trigger,endprep}_slot in side arrays (else 0)
constexpr int M_ACKS_SH = 8;               // leader_bk.accept_acks  (8 bits)
constexpr int M_PACKS_SH = 16;             // leader_bk.prepare_acks (8 bits)
constexpr int M_SRC_SH = 24;               // replica_bk.source      (3 bits)
constexpr uint32_t M_NONEMPTY = 1u << 27;  // !Instance::reqs.is_empty()  (mirrors s_val != 0)
constexpr int M_VMODE_SH = 28;             // Instance::voted encoding (2 bits)
constexpr uint32_t VM_NONE = 0;            // voted == (0, empty)
constexpr uint32_t VM_SAME = 1;            // voted == (bal, reqs) as stored
constexpr uint32_t VM_SIDE = 2;            // voted in s_vbal / s_vval

// outbox entry kinds, stored in the top 2 bits of ob_slot
constexpr uint32_t OB_PREPARE = 1, OB_ACCEPT = 2, OB_HEARTBEAT = 3;
constexpr int OB_KIND_SH = 30;
constexpr uint32_t OB_SLOT_MASK = (1u << 30) - 1;

constexpr uint32_t NO_REP = 0xFFu;
constexpr int MAXR = 8;

// Wave-tiled indexing of every array with a row dimension (ring slots, outbox entries,
// ack-matrix rows, PrepareReply entries):  X[g / 64][row][g % 64].  A wavefront's 64
// consecutive groups still form one contiguous 256/512-byte request per row, and the
// rows of ONE group tile are adjacent in memory (64 slots of a group span 16-32 KB, not
// 64 pages 256 KB apart), which is what the lane-per-slot cooperative paths need.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline size_t tix(uint32_t rows, uint32_t row, uint32_t g);
// byte of replica r in the ack word of (entry j, group g)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline size_t ack_ix(uint32_t cap, uint32_t j, uint32_t r, uint32_t g) { return tix(cap, j, g) * 8 + r; }
// Experiment (tools/experiments/README.md): the answers to the first 64 entries of an outbox as ONE word per
// (follower q, group) -- bit j = follower q accepted entry j -- behind the ack words of the same replica:
// word index tix(SMR_MAX_REPLICAS, q, g).  A follower stores one word per group instead of a byte per entry; the
// leader loads R words per group instead of one word per row.  Entries >= 64 (the long outbox of a re-Accept
// round) keep their byte cells.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline SMR_G uint64_t *ack_bits_base(SMR_G uint8_t *ack, uint32_t cap, uint32_t G) {
    return (SMR_G uint64_t *)(ack + (size_t)cap * (((size_t)G + 63) / 64 * 64) * 8);
}
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint64_t ack_range_bits(uint32_t lo, uint32_t hi) {       // bits [lo, hi) clipped to [0, 64)
    if (hi > 64u) hi = 64u;
    if (lo >= hi) return 0ull;
    const uint64_t upto_hi = hi == 64u ? ~0ull : ((1ull << hi) - 1ull);
    return upto_hi & ~((1ull << lo) - 1ull);
}
#if defined(__HIPCC__)
__host__ __device__
#endif
inline size_t tix(uint32_t rows, uint32_t row, uint32_t g) {
#ifdef SMR_ROWMAJOR_G   /* experiment only: plain [row][g] with a compile-time group count */
    (void)rows;
    return (size_t)row * SMR_ROWMAJOR_G + g;
#else
    return ((((size_t)(g >> 6)) * rows + row) << 6) | (size_t)(g & 63u);
#endif
}

struct MpRep {
    // scalars [G]
    SMR_G uint8_t *leader;
    SMR_G uint64_t *bal_prep_sent, *bal_prepared, *bal_max_seen;
    SMR_G uint32_t *start_slot, *log_len, *accept_bar, *commit_bar, *exec_bar, *snap_bar;
    SMR_G uint32_t *null_lb;        // aux: no Null instance in [exec_bar, null_lb)
    SMR_G uint32_t *bal_lo;         // experiment: every slot in [bal_lo, log_len) holds bal == bal_max_seen (0xFFFFFFFF: none)
    SMR_G uint32_t *peer_exec_bar;  // [R][G]
    // slot ring [W][G]
    SMR_G uint64_t *s_bal;
    SMR_G uint32_t *s_val;          // reqs token (0 = empty batch)
    SMR_G uint32_t *s_meta;
    SMR_G uint64_t *s_vbal; SMR_G uint32_t *s_vval;
    SMR_G uint64_t *s_pmax;
    SMR_G uint32_t *s_ltrig, *s_lendp, *s_rtrig, *s_rendp;
    // outbox [2][cap][G]
    SMR_G uint32_t *ob_cnt[2];
    SMR_G uint32_t *ob_slot[2];     // kind<<30 | slot   (HB: commit_bar)
    SMR_G uint64_t *ob_bal[2];
    SMR_G uint32_t *ob_val[2];      // Accept: reqs token; HB: exec_bar
    SMR_G uint32_t *ob_aux[2];      // HB: snap_bar
    // "regular outbox" descriptor [G]: ob_reg != 0 means the outbox holds ONLY Accepts for the
    // consecutive slots (ob_reg - 1) + j, j < ob_cnt, all at ballot ob_rbal (what the steady-state
    // append produces), so consumers need not read ob_slot / ob_bal.  Any other push clears it.
    SMR_G uint32_t *ob_reg[2];
    SMR_G uint64_t *ob_rbal[2];
    // replies to my Accepts: one 8-byte word per (entry j, group), byte r = 1 iff replica r answered my j-th
    // message.  An AcceptReply always carries the ballot of the Accept it answers (durability.rs:108-131),
    // so the cell need not repeat it; followers store their byte, the leader loads the word
    SMR_G uint8_t *ack;
    // my PrepareReply batch of this tick: header [G] + entries [pcap][G]
    SMR_G uint32_t *pr_cnt; SMR_G uint8_t *pr_dest;
    SMR_G uint32_t *pr_trig, *pr_endp, *pr_abar; SMR_G uint64_t *pr_bal;
    SMR_G uint64_t *pr_vbal; SMR_G uint32_t *pr_vval;
    // heartbeat record [G]
    SMR_G uint64_t *hb_bal; SMR_G uint32_t *hb_commit, *hb_exec, *hb_snap;
    // outputs
    SMR_G unsigned long long *counters;   // [0] commits [1] redirects [2] rejects
    SMR_G unsigned long long *clist;      // (group << 32) | slot
    SMR_G unsigned int *clist_n;
};

// The arena lays every replica's arrays out identically, `rep_stride` bytes apart, so the arrays
// of replica r are replica 0's shifted by r * rep_stride.  RepView is that addressing: the round
// kernels use it for their own replica (uniform per block) and for a sender or leader that
// differs from group to group, i.e. from lane to lane.
#if defined(__HIPCC__)
#define SMR_HD __host__ __device__ __forceinline__
#else
#define SMR_HD inline
#endif
struct RepView {
    const MpRep &b;
    size_t ro;                      // byte offset of my replica's arrays from replica 0's
    template <typename T> SMR_HD T *sh(T *p0) const { return (T *)((SMR_G char *)p0 + ro); }
    SMR_HD SMR_G uint8_t *leader() const { return sh(b.leader); }
    SMR_HD SMR_G uint64_t *bal_prep_sent() const { return sh(b.bal_prep_sent); }
    SMR_HD SMR_G uint64_t *bal_prepared() const { return sh(b.bal_prepared); }
    SMR_HD SMR_G uint64_t *bal_max_seen() const { return sh(b.bal_max_seen); }
    SMR_HD SMR_G uint32_t *start_slot() const { return sh(b.start_slot); }
    SMR_HD SMR_G uint32_t *log_len() const { return sh(b.log_len); }
    SMR_HD SMR_G uint32_t *accept_bar() const { return sh(b.accept_bar); }
    SMR_HD SMR_G uint32_t *commit_bar() const { return sh(b.commit_bar); }
    SMR_HD SMR_G uint32_t *exec_bar() const { return sh(b.exec_bar); }
    SMR_HD SMR_G uint32_t *snap_bar() const { return sh(b.snap_bar); }
    SMR_HD SMR_G uint32_t *null_lb() const { return sh(b.null_lb); }
    SMR_HD SMR_G uint32_t *bal_lo() const { return sh(b.bal_lo); }
    SMR_HD SMR_G uint32_t *peer_exec_bar() const { return sh(b.peer_exec_bar); }
    SMR_HD SMR_G uint64_t *s_bal() const { return sh(b.s_bal); }
    SMR_HD SMR_G uint32_t *s_val() const { return sh(b.s_val); }
    SMR_HD SMR_G uint32_t *s_meta() const { return sh(b.s_meta); }
    SMR_HD SMR_G uint64_t *s_vbal() const { return sh(b.s_vbal); }
    SMR_HD SMR_G uint32_t *s_vval() const { return sh(b.s_vval); }
    SMR_HD SMR_G uint64_t *s_pmax() const { return sh(b.s_pmax); }
    SMR_HD SMR_G uint32_t *s_ltrig() const { return sh(b.s_ltrig); }
    SMR_HD SMR_G uint32_t *s_lendp() const { return sh(b.s_lendp); }
    SMR_HD SMR_G uint32_t *s_rtrig() const { return sh(b.s_rtrig); }
    SMR_HD SMR_G uint32_t *s_rendp() const { return sh(b.s_rendp); }
    SMR_HD SMR_G uint8_t *ack() const { return sh(b.ack); }
    SMR_HD SMR_G uint32_t *pr_cnt() const { return sh(b.pr_cnt); }
    SMR_HD SMR_G uint8_t *pr_dest() const { return sh(b.pr_dest); }
    SMR_HD SMR_G uint32_t *pr_trig() const { return sh(b.pr_trig); }
    SMR_HD SMR_G uint32_t *pr_endp() const { return sh(b.pr_endp); }
    SMR_HD SMR_G uint32_t *pr_abar() const { return sh(b.pr_abar); }
    SMR_HD SMR_G uint64_t *pr_bal() const { return sh(b.pr_bal); }
    SMR_HD SMR_G uint64_t *pr_vbal() const { return sh(b.pr_vbal); }
    SMR_HD SMR_G uint32_t *pr_vval() const { return sh(b.pr_vval); }
    SMR_HD SMR_G uint64_t *hb_bal() const { return sh(b.hb_bal); }
    SMR_HD SMR_G uint32_t *hb_commit() const { return sh(b.hb_commit); }
    SMR_HD SMR_G uint32_t *hb_exec() const { return sh(b.hb_exec); }
    SMR_HD SMR_G uint32_t *hb_snap() const { return sh(b.hb_snap); }
    SMR_HD SMR_G unsigned long long *counters() const { return sh(b.counters); }
    SMR_HD SMR_G unsigned long long *clist() const { return sh(b.clist); }
    SMR_HD SMR_G unsigned int *clist_n() const { return sh(b.clist_n); }
    SMR_HD SMR_G uint32_t *ob_cnt(int p) const { return sh(b.ob_cnt[p]); }
    SMR_HD SMR_G uint32_t *ob_slot(int p) const { return sh(b.ob_slot[p]); }
    SMR_HD SMR_G uint64_t *ob_bal(int p) const { return sh(b.ob_bal[p]); }
    SMR_HD SMR_G uint32_t *ob_val(int p) const { return sh(b.ob_val[p]); }
    SMR_HD SMR_G uint32_t *ob_aux(int p) const { return sh(b.ob_aux[p]); }
    SMR_HD SMR_G uint32_t *ob_reg(int p) const { return sh(b.ob_reg[p]); }
    SMR_HD SMR_G uint64_t *ob_rbal(int p) const { return sh(b.ob_rbal[p]); }
};

struct MpParams {
    uint32_t G, W, Wmask, cap, pcap, win_reserve, clist_cap;
    uint32_t R, quorum, thresh, rspaxos;
    SMR_G uint8_t *overflow;        // [G] sticky, shared by all replicas of a group
    SMR_G unsigned long long *dbg;  // [64] debug clock stamps
    SMR_G uint8_t *r3_need;         // [R][ceil(G/64)]: this 64-group tile has work left for mp_round_replies
    SMR_G uint8_t *r2_need;         // [R][ceil(G/64)]: a lane of this row's tile left messages to mp_round_deliver_rest (round 5) ...
    SMR_G uint32_t *r2_res;         // [R][G] ... and where it stopped: 1 | sender << 4 | entry << 8 (0: nothing left)
    // straggler list of the tick (mp_mark_stragglers): groups in a leader change are taken out of the
    // bulk launches and run one per wavefront on a side stream
    SMR_G uint8_t *slow;            // [G] 1 = on the list this tick
    SMR_G uint8_t *slow_ttl;        // [G] ticks left on the list
    SMR_G uint32_t *slow_list;      // [slow_cap]
    SMR_G uint32_t *slow_n;         // [2] list length, by tick parity
    uint32_t slow_cap;
    // role rotation (smr_mp_set_role_rotation): block row y of a bulk round launch takes, of group g, replica
    // (y + role_rot[g]) mod R instead of replica y; role_rot[g] = the group's leader as the tick's mark pass found it, so row 0
    // runs every group's LEADER and rows 1 .. R - 1 its followers -- a wavefront then runs one role's code, whoever leads
    SMR_G uint8_t *role_rot;        // [G]
    // round 6: mp_quorum_tally of tick t may already have run the leader's steady-state handle_req_batch calls of tick t + 1
    // (MpNextLocal, smr_mp_run_ticks); r1_body of tick t + 1 then finds 1 here, skips that replica's batches and clears it
    SMR_G uint8_t *r1_done;         // [G]
    uint32_t rot_on;
    uint32_t live;                  // bit r: replica r runs on this device (spread layout: the others are images, see mp_img_*)
    size_t rep_stride;              // bytes from an array of replica d to the same array of replica d + 1
    MpRep rep[MAXR];
};

// One tick's device inputs (the arguments of smr_mp_tick), and a batch of consecutive ticks handed to ONE launch of
// the fused tick kernel (mp_ticks_fused) as a by-value kernel argument: the tick index is block-uniform, so the
// descriptors are read with scalar loads from the kernarg segment.
struct MpTickIn {
    const uint8_t *timeout_rep, *timeout_src, *req_target;
    const uint32_t *req_cnt, *req_val, *ackctl;
    uint32_t S;
    int32_t heartbeat;
};
// The client batches of the NEXT tick, handed to mp_quorum_tally (round 6): a group whose tick the tally's closed form completes
// gets its leader's steady-state appends of that tick in the same lanes (req_target == nullptr: none).
struct MpNextLocal {
    const uint8_t *timeout_rep, *req_target;
    const uint32_t *req_cnt, *req_val;
    uint32_t S;
};
constexpr uint32_t MP_FUSED_MAXT = 16;
struct MpTickBatch {
    uint32_t n;
    int32_t par0;                   // outbox parity of the first tick
    MpTickIn t[MP_FUSED_MAXT];
};

}  // namespace smr
