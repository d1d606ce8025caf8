/*
 * summerset_hip.h -- C-ABI of libsummerset_hip.so, the MI355X (gfx950) batched
 * multi-group consensus engine.
 *
 * The reference (josehu07/summerset) has NO FFI/C-ABI/plugin interface for this
 * path; its nearest boundary is the Rust trait `GenericReplica`
 * (src/server/replica.rs:16-42) plus the per-protocol event-handler
 * convention `handle_req_batch / handle_msg_recv / handle_log_result /
 * handle_cmd_result` (dispatch sites src/protocols/multipaxos/mod.rs:837-905).
 * Every entry point below names the reference handler(s) it stands in for;
 * INTEGRATION.md shows the Rust `extern "C"` binding a maintainer would add.
 *
 * Conventions:
 *   - plain pointers + sizes, no C++/torch types;
 *   - `int` return: 0 = ok, < 0 = error class, text via smr_last_error()
 *     (mirrors SummersetError(String), src/utils/error.rs:7);
 *   - outdated / out-of-range inputs are IGNORED and counted, never errors
 *     (multipaxos/messages.rs:377-379,394-406);
 *   - a handle has ONE owner thread (the reference's protocol struct is
 *     `&mut self` single-owner, README.md:59); it is not thread-safe;
 *   - pointers named *_dev are DEVICE (HBM) pointers, *_host are host
 *     pointers; `stream` is a hipStream_t passed as void* (NULL = default).
 *
 * All per-group arrays are structure-of-arrays with the GROUP index fastest
 * ("one group = one lane"): X[g], Y[k][g] == Y[k * n_groups + g].
 */
#ifndef SUMMERSET_HIP_H
#define SUMMERSET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMR_OK 0
#define SMR_ERR_ARG (-1)     /* bad argument / configuration */
#define SMR_ERR_DEVICE (-2)  /* HIP runtime error (no device, OOM, launch failure) */
#define SMR_ERR_STATE (-3)   /* call not valid in the current state */

#define SMR_NO_REPLICA 0xFFu /* Option<ReplicaId>::None */
#define SMR_MAX_REPLICAS 8

/* Status enum of multipaxos/mod.rs:168-174 (also RSPaxos). */
enum { SMR_ST_NULL = 0, SMR_ST_PREPARING = 1, SMR_ST_ACCEPTING = 2, SMR_ST_COMMITTED = 3, SMR_ST_EXECUTED = 4 };

/* ackctl word (per outbox entry, per group): delivery order of the peers'
 * replies + which of them are lost.  bits 0..23: replica ids, 3 bits each, in
 * delivery order (the receiver skips its own id and ids >= population);
 * bits 24..31: drop mask by replica id. */
#define SMR_CTL_IDENTITY 0x00FAC688u

const char *smr_last_error(void);
/* number of visible HIP devices, or < 0 */
int smr_device_count(void);
/* library/ABI version, bumped on incompatible change */
uint32_t smr_abi_version(void);

/* ------------------------------------------------------------------------
 * Reed-Solomon over GF(2^8)  (RSPaxos / CRaft / Crossword value sharding)
 * replaces: RSCodeword::compute_parity  src/utils/rscoding.rs:447-486
 *           (-> reed_solomon_erasure::galois_8::ReedSolomon::encode :484)
 *           with the from_data shard geometry of :165-220 fused in
 *           (shard_len = ceil(data_len / d), zero padding, contiguous split);
 *           RSCodeword::reconstruct{,_data}  :490-537;  verify_parity :541-577.
 * ---------------------------------------------------------------------- */

/* (d+p) x d coding matrix, row major, into a host buffer. */
int smr_rs_matrix(int d, int p, uint8_t *out_host);

/* ceil(data_len / d), rscoding.rs:177-181 */
uint64_t smr_rs_shard_len(uint64_t data_len, int d);

/* Encode n_cw codewords.  Codeword i occupies data_dev[i*cw_stride ..
 * i*cw_stride + data_len) (the serialized bytes; need NOT be padded -- bytes
 * at and beyond data_len are treated as zero and never read).  Data shard k
 * is bytes [k*shard_len, (k+1)*shard_len) of that range.  Parity shard k of
 * codeword i is written to parity_dev[i*par_stride + k*par_shard_stride ..
 * + shard_len).  Returns SMR_ERR_ARG for the reference's error cases
 * (d == 0, data_len == 0 "codeword is null"); p == 0 is a no-op. */
int smr_rs_encode(const uint8_t *data_dev, uint64_t data_len, uint64_t cw_stride, uint64_t n_cw,
                  int d, int p, uint8_t *parity_dev, uint64_t par_stride,
                  uint64_t par_shard_stride, void *stream);

/* RSCodeword::from_data + compute_parity in ONE pass (rscoding.rs:165-243, :447-486): codeword i's serialized bytes are
 * src_dev[i*src_stride .. + data_len) (rows may be packed: src_stride == data_len); its d data shards -- consecutive
 * shard_len-byte slices of those bytes, the last one zero-padded (rscoding.rs:188-200) -- and its p parity shards are
 * written to cw_dev[i*cw_stride + k*shard_len .. + shard_len), k in [0, d+p).  The bytes are read once: against
 * "copy into the codeword buffer, then smr_rs_encode" this moves (1 + (d+p)/d) L instead of (3 + p/d) L per codeword.
 * p == 0 only lays out the data shards.  src and cw buffers must not overlap.  Errors as smr_rs_encode. */
int smr_rs_from_data_encode(const uint8_t *src_dev, uint64_t data_len, uint64_t src_stride, uint64_t n_cw, int d, int p,
                            uint8_t *cw_dev, uint64_t cw_stride, void *stream);

/* ... and with the shard fan-out of an RSPaxos / CRaft leader in the same pass (rspaxos/request.rs:127-142: shard k goes to
 * replica k): besides the codeword at cw_dev, shard k of codeword i is ALSO written to
 * fan_dev[k*fan_shard_stride + i*fan_cw_stride .. + shard_len) for every k whose bit is set in fan_mask -- one contiguous
 * store per shard holder (fan_cw_stride >= shard_len, fan_shard_stride >= the bytes of one store), filled without reading
 * the codewords again.  fan_mask == 0: exactly smr_rs_from_data_encode. */
int smr_rs_from_data_encode_fanout(const uint8_t *src_dev, uint64_t data_len, uint64_t src_stride, uint64_t n_cw, int d, int p,
                                   uint8_t *cw_dev, uint64_t cw_stride, uint8_t *fan_dev, uint64_t fan_shard_stride,
                                   uint64_t fan_cw_stride, uint32_t fan_mask, void *stream);
/* from_data + compute_parity with every shard written ONCE, shard-major: shard k (k < d: data, zero padding included; else
 * parity k - d) of codeword i goes to stores_dev + k * shard_stride + i * cw_stride -- store k is what replica k holds of the
 * batch (rspaxos/request.rs:127-142: the Accept for peer k carries subset_copy({k})), and the leader's own codeword IS the
 * d + p stores (rscoding.rs:255-346: a codeword is a set of shards, not a buffer).  No contiguous codeword is written:
 * L bytes read, (d + p) * shard_len written per codeword -- smr_rs_from_data_encode_fanout wrote every shard twice.
 * smr_rs_reconstruct / smr_rs_verify take the same layout (shard_stride, cw_stride). */
int smr_rs_from_data_encode_stores(const uint8_t *src_dev, uint64_t data_len, uint64_t src_stride, uint64_t n_cw, int d, int p,
                                   uint8_t *stores_dev, uint64_t shard_stride, uint64_t cw_stride, void *stream);
/* The general form: shard k of codeword i also to shard_dst[k] + i*dst_cw_stride for every k with shard_dst[k] != NULL
 * (shard_dst: HOST array of d + p DEVICE pointers) -- e.g. straight into the per-destination slices of a collective's send
 * buffer, which are not equally spaced (summerset_amd/spread_rsp.py). */
int smr_rs_from_data_encode_scatter(const uint8_t *src_dev, uint64_t data_len, uint64_t src_stride, uint64_t n_cw, int d, int p,
                                    uint8_t *cw_dev, uint64_t cw_stride, uint8_t *const *shard_dst, uint64_t dst_cw_stride, void *stream);

/* Same, with the GF(2^8) multiplies done through LDS-resident product tables
 * instead of bit-sliced xtime arithmetic (kept selectable for A/B runs). */
int smr_rs_encode_lut(const uint8_t *data_dev, uint64_t data_len, uint64_t cw_stride, uint64_t n_cw,
                      int d, int p, uint8_t *parity_dev, uint64_t par_stride,
                      uint64_t par_shard_stride, void *stream);

/* Rebuild missing shards of n_cw codewords that all share one erasure
 * pattern.  shards_dev: shard k of codeword i at i*cw_stride + k*shard_stride,
 * shard_len bytes each, k in [0, d+p).  present_mask bit k = shard k valid.
 * data_only != 0 rebuilds data shards only (reconstruct_data).  Returns
 * SMR_ERR_ARG when fewer than d shards are present. */
int smr_rs_reconstruct(uint8_t *shards_dev, uint64_t shard_len, uint64_t shard_stride,
                       uint64_t cw_stride, uint64_t n_cw, int d, int p, uint32_t present_mask,
                       int data_only, void *stream);

/* verify_parity for n_cw codewords; ok_dev[i] = 1 if parity matches. */
int smr_rs_verify(const uint8_t *shards_dev, uint64_t shard_len, uint64_t shard_stride,
                  uint64_t cw_stride, uint64_t n_cw, int d, int p, uint8_t *ok_dev, void *stream);

/* ------------------------------------------------------------------------
 * MultiPaxos / RSPaxos cluster of G groups x R replicas in lock-step (LS-1)
 * ---------------------------------------------------------------------- */
typedef struct smr_mp_cluster smr_mp_cluster;
#define SMR_STRAGGLER_OFF 0xFF

typedef struct {
    uint32_t n_groups;      /* G */
    uint8_t population;     /* R, 3..8 */
    uint8_t commit_extra;   /* RSPaxos fault_tolerance f (rspaxos/messages.rs:438-439); 0 = MultiPaxos */
    uint8_t straggler_ticks; /* ticks a group spends on the engine's side stream after a HearTimeout, so its leader
                              * change runs beside the steady-state kernels (results do not depend on it):
                              * 0 / SMR_STRAGGLER_OFF = never (default) */
    uint8_t reserved1;
    uint32_t window;        /* W: ring slots per replica per group, power of two */
    uint32_t win_reserve;   /* leader refuses new batches once W - win_reserve slots are live */
    uint32_t outbox_cap;    /* max messages a replica may emit per tick (>= W + 4 recommended) */
    uint32_t commit_list_cap; /* entries of the per-replica committed-slot list (0 = no list) */
} smr_mp_cfg;

/* Per-replica, per-group state as the reference keeps it
 * (multipaxos/mod.rs:387-514). */
typedef struct {
    uint8_t leader;                  /* SMR_NO_REPLICA = None */
    uint8_t overflow;                /* group frozen: ring window / outbox exhausted */
    uint64_t bal_prep_sent, bal_prepared, bal_max_seen;
    uint32_t start_slot, log_len;    /* log covers [start_slot, log_len) */
    uint32_t accept_bar, commit_bar, exec_bar, snap_bar;
    uint32_t peer_exec_bar[SMR_MAX_REPLICAS];
} smr_mp_group_state;

int smr_mp_cluster_create(const smr_mp_cfg *cfg, smr_mp_cluster **out);
void smr_mp_cluster_destroy(smr_mp_cluster *c);

/* Synthetic start state of SURVEY.md §8d config 2: replica `rep` already
 * prepared with ballot make_unique_ballot(1) on an empty log. */
int smr_mp_preset_leader(smr_mp_cluster *c, uint8_t rep);

/*
 * One lock-step tick = rounds R1..R4 below for every group.  Device inputs
 * (NULL = none):
 *   timeout_rep_dev[G], timeout_src_dev[G]   HeartbeatEvent::HearTimeout on
 *       replica timeout_rep (SMR_NO_REPLICA = none) about peer timeout_src
 *   req_target_dev[G], req_cnt_dev[G], req_val_dev[S][G]   client batches:
 *       req_cnt[g] <= S opaque non-zero batch tokens delivered to replica
 *       req_target[g]
 *   ackctl_dev[outbox_cap][G]   reply order / loss per outbox entry
 *   do_heartbeat   run the heartbeat round + ring trim this tick
 */
int smr_mp_tick(smr_mp_cluster *c, const uint8_t *timeout_rep_dev, const uint8_t *timeout_src_dev,
                const uint8_t *req_target_dev, const uint32_t *req_cnt_dev,
                const uint32_t *req_val_dev, uint32_t S, const uint32_t *ackctl_dev,
                int do_heartbeat, void *stream);

/* The four rounds of a tick as separate calls (a host that owns real I/O, or
 * the multi-GPU driver, interleaves its exchange between them):
 *  R1  become_a_leader (leadership.rs:73-214) on HearTimeout, handle_req_batch
 *      (request.rs:112-224) + leader self-ack (durability.rs:85-107)
 *  R2  handle_msg_prepare / handle_msg_accept (messages.rs:12-83,295-367) on
 *      every peer + their WAL completions (durability.rs:10-145): fills the
 *      senders' ack matrices and PrepareReply lists
 *  R3  handle_msg_prepare_reply / handle_msg_accept_reply (messages.rs:87-292,
 *      370-443) + handle_logged_commit_slot (durability.rs:148-218) +
 *      handle_cmd_result exec-bar scan (execution.rs:56-79): THE quorum kernel
 *  R4  all-to-all Heartbeat -> heard_heartbeat / advance_commit_bar
 *      (leadership.rs:217-427), then ring trim (snapshot.rs:121-186 log part)
 */
int smr_mp_round_local(smr_mp_cluster *c, const uint8_t *timeout_rep_dev,
                       const uint8_t *timeout_src_dev, const uint8_t *req_target_dev,
                       const uint32_t *req_cnt_dev, const uint32_t *req_val_dev, uint32_t S,
                       void *stream);
int smr_mp_round_deliver(smr_mp_cluster *c, void *stream);
int smr_mp_round_replies(smr_mp_cluster *c, const uint32_t *ackctl_dev, int publish_heartbeat,
                         void *stream);
int smr_mp_round_heartbeat(smr_mp_cluster *c, void *stream);
/* A batch of consecutive ticks in as few launches as possible (co-located layout only: every replica of a group is
 * on this device).  ticks[i] holds the arguments smr_mp_tick takes for tick i (device pointers, NULL = none); the
 * result is bit for bit what n smr_mp_tick calls give.  One fused kernel runs the four rounds of up to 16 ticks back
 * to back with block barriers where the per-round launches have kernel boundaries: a block owns 64 groups with all
 * their replicas, so a leader change's serial latency delays its own block only.  Needs straggler_ticks = off. */
typedef struct {
    const uint8_t *timeout_rep_dev, *timeout_src_dev, *req_target_dev;
    const uint32_t *req_cnt_dev, *req_val_dev;
    uint32_t S;
    const uint32_t *ackctl_dev;
    int do_heartbeat;
} smr_mp_tick_in;
int smr_mp_run_ticks(smr_mp_cluster *c, const smr_mp_tick_in *ticks, uint32_t n, void *stream);
/* ---- spread layout: the replicas of a group on different ranks --------------------------------
 * smr_mp_set_live: bit r of live_mask = replica r of this cluster object runs on this device; the round calls then
 * skip the others, which are IMAGES: only what the rounds hand from one replica to another exists of them here.
 * Between rounds the host moves those pieces (server/transport.rs:208-275 send_msg / bcast_msg stand-in):
 *   after R1  SMR_IMG_OUTBOX           of every live replica -> the ranks holding the group's other replicas
 *   after R2  SMR_IMG_ACKS             (rep = an image's id, other = my live follower): the AcceptReplies it is owed -> its rank
 *             SMR_IMG_PREPARE_REPLIES  of every live replica -> the ranks holding the group's other replicas
 *   after R3  SMR_IMG_HEARTBEAT        of every live replica -> ... (ticks with a heartbeat round)
 * as fixed-size device buffers: smr_mp_image_bytes(kind, rows, ovf_cap) bytes, `rows` = outbox entries per group shipped
 * in the per-group part (the tick's S), ovf_cap = entries of the overflow list that carries what a leader change adds.
 * pack reads a LIVE replica (ACKS: the image `rep` my live follower `other` wrote into), unpack writes an image
 * (ACKS: into my live sender `rep`, the column of follower `other`).  An overflow list that ran full is counted in the
 * image header (uint32 [2]) and the entries are lost: size ovf_cap for the leader changes a tick can hold. */
enum { SMR_IMG_OUTBOX = 0, SMR_IMG_ACKS = 1, SMR_IMG_PREPARE_REPLIES = 2, SMR_IMG_HEARTBEAT = 3 };
int smr_mp_set_live(smr_mp_cluster *c, uint32_t live_mask);
/* Role rotation (round 3; needs straggler_ticks > 0 and every replica on this device).  The bulk round launches are a grid of
 * (64-group tiles) x (R rows); row y normally runs replica y of every group, so once leaders have moved a wavefront holds
 * leaders and followers side by side and runs both roles' code.  With rotation on, row y runs replica (y + leader[g]) mod R of
 * group g -- leader[g] as the tick's mark pass read it --: row 0 is every group's leader, rows 1 .. R - 1 its followers.
 * Every (group, replica) pair is still handled exactly once per round; results are bit for bit the same. */
int smr_mp_set_role_rotation(smr_mp_cluster *c, int on);
int64_t smr_mp_image_bytes(smr_mp_cluster *c, int kind, uint32_t rows, uint32_t ovf_cap);
int smr_mp_image_pack(smr_mp_cluster *c, int kind, uint8_t rep, uint8_t other, uint8_t *img_dev, uint64_t img_bytes,
                      uint32_t rows, uint32_t ovf_cap, void *stream);
int smr_mp_image_unpack(smr_mp_cluster *c, int kind, uint8_t rep, uint8_t other, const uint8_t *img_dev, uint64_t img_bytes,
                        uint32_t rows, uint32_t ovf_cap, void *stream);
/* A whole exchange at once.  Which replica's piece goes into which slice of the send / receive buffer never changes, so
 * the operations are handed over once (smr_mp_image_plan_create) and an exchange is then three launches to pack -- clear the
 * headers, pack every image, duplicate the pieces that go to several ranks (copy_of_dev != NULL: this image is a copy of
 * the one packed at that address) -- and one to unpack, whatever the number of images.  All clusters of a plan must be in
 * the same tick. */
typedef struct {
    smr_mp_cluster *cluster;
    int kind;
    uint8_t rep, other;
    uint8_t *img_dev;
    uint64_t img_bytes;
    const uint8_t *copy_of_dev;
} smr_mp_image_op;
typedef struct smr_mp_image_plan smr_mp_image_plan;
int smr_mp_image_plan_create(const smr_mp_image_op *ops, uint32_t n, uint32_t rows, uint32_t ovf_cap, smr_mp_image_plan **out);
void smr_mp_image_plan_destroy(smr_mp_image_plan *p);
int smr_mp_image_plan_run(smr_mp_image_plan *p, int unpack, void *stream);
/* closes the tick (flips the outbox parity); smr_mp_tick calls it itself */
int smr_mp_end_tick(smr_mp_cluster *c);
/* A rank's part of an L2 job as ONE object, the tick's orchestration in the library: `clusters` = the rank's block clusters (in the
 * order the inputs will come in), pack[k] / unpack[k] = the plans of exchange k (0 outbox, 1 replies, 2 heartbeat; they stay the
 * caller's, as do the clusters).  A tick is then one call per SEGMENT with the caller's collective -- one all_to_all_single over
 * the plans' send / receive buffers (torch.distributed = RCCL) -- between the segments:
 *   0: R1 on every block, pack(outbox)            | exchange 0 |
 *   1: unpack(outbox), R2 on every block, pack(replies)         | exchange 1 |
 *   2: unpack(replies), R3 on every block; heartbeat tick: pack(heartbeat) | exchange 2 | else: the tick ends
 *   3: (heartbeat ticks) unpack(heartbeat), R4 on every block, the tick ends
 * in[b] = block b's inputs (segments 0 and 2 read them; do_heartbeat is ignored: `heartbeat` is the tick's).  What
 * summerset_amd/spread_mp.py used to do call by call (~25 host calls per segment at 4 ranks). */
typedef struct smr_mp_spread smr_mp_spread;
int smr_mp_spread_create(smr_mp_cluster *const *clusters, uint32_t n_blocks, smr_mp_image_plan *const *pack, smr_mp_image_plan *const *unpack,
                         smr_mp_spread **out);
void smr_mp_spread_destroy(smr_mp_spread *s);
int smr_mp_spread_segment(smr_mp_spread *s, int segment, const smr_mp_tick_in *in, int heartbeat, void *stream);
/* A segment (or the collective between two) failed: close the open tick so that the object takes segment 0, bind_comm and
 * smr_mp_spread_tick again.  Runs and undoes nothing: the blocks hold a partly run tick, the host restores them (smr_mp_load_state
 * or new clusters) before it ticks again.  smr_mp_spread_tick does this itself when one of its steps fails. */
int smr_mp_spread_abort_tick(smr_mp_spread *s);
/* How the blocks' rounds inside a segment are launched: 2 (default since round 5) = ONE launch for all blocks of the rank
 * (blockIdx.z = the block; ~12 host calls per segment less than 1); 1 = side by side on streams of the object's own, forked
 * behind the segment's unpack and joined in front of its pack on `stream` (round 4); 0 = one after the other on `stream`.
 * Same results either way. */
int smr_mp_spread_set_concurrent(smr_mp_spread *s, int on);

/* ---- the exchange itself: RCCL behind the C-ABI -----------------------------------------------------------------------------
 * Replaces `TransportHub::send_msg` / `bcast_msg` (/root/reference/src/server/transport.rs:208-275) for a host that binds
 * this library: instead of one bincode frame per peer through a TCP messenger task, a rank hands over ONE packed device
 * buffer with a byte count per peer rank and receives ONE buffer back (an all-to-all with split sizes), issued as grouped
 * ncclSend / ncclRecv pairs over xGMI on the caller's HIP stream.  The call only enqueues; the buffers stay the caller's.
 *   smr_comm_unique_id   one process makes the 128-byte id (the manager's role: clusman.rs hands out ids and peer lists) and
 *                        ships it over the host's control channel;
 *   smr_comm_init_rank   every rank, with its device current (hipSetDevice), blocks until all `world` ranks have called;
 *   smr_comm_exchange    segment k of `send_dev` (send_bytes[k] bytes, segments back to back in rank order) goes to rank k,
 *                        recv_bytes[k] bytes from rank k land in segment k of `recv_dev`; a rank's own segment is a device
 *                        copy (SMR_COMM_SELF_VIA_RCCL: through an ncclSend / ncclRecv pair to itself, for tests);
 *   smr_comm_all_reduce_u64  sum / max of n u64 words in place (agreeing on a tick count, a job's max elapsed time);
 *   smr_comm_info        out = {rank, world, exchanges so far, bytes sent to other ranks, bytes received from them}. */
typedef struct smr_comm smr_comm;
#define SMR_COMM_ID_BYTES 128
#define SMR_COMM_SELF_VIA_RCCL 1u
#define SMR_COMM_SUM 0
#define SMR_COMM_MAX 1
int smr_comm_unique_id(uint8_t *out, uint64_t cap);
int smr_comm_init_rank(const uint8_t *id, uint64_t id_bytes, uint32_t rank, uint32_t world, smr_comm **out);
void smr_comm_destroy(smr_comm *c);
int smr_comm_exchange(smr_comm *c, const void *send_dev, const uint64_t *send_bytes, void *recv_dev, const uint64_t *recv_bytes,
                      uint32_t flags, void *stream);
int smr_comm_all_reduce_u64(smr_comm *c, uint64_t *inout_dev, uint64_t n, int op, void *stream);
int smr_comm_info(smr_comm *c, uint64_t out[5]);
/* The L2 tick with its exchanges INSIDE the library: bind the communicator and the three exchanges' buffers -- send_dev[k] /
 * recv_dev[k] = the buffers exchange k's pack / unpack plans were made over, send_bytes[k * world + r] / recv_bytes[k * world + r]
 * = the bytes exchange k moves to / from rank r (k: 0 outbox, 1 replies, 2 heartbeat) -- then ONE call per tick:
 * smr_mp_spread_tick = segment 0, exchange 0, segment 1, exchange 1, segment 2 [, exchange 2, segment 3], all enqueued on
 * `stream`.  comm = NULL unbinds (the segments' caller runs the collectives again). */
int smr_mp_spread_bind_comm(smr_mp_spread *s, smr_comm *comm, const void *const send_dev[3], const uint64_t *send_bytes,
                            void *const recv_dev[3], const uint64_t *recv_bytes, uint32_t world);
int smr_mp_spread_tick(smr_mp_spread *s, const smr_mp_tick_in *in, int heartbeat, void *stream);

/* Device pointer + geometry of replica `rep`'s ack matrix: one 8-byte word per (outbox entry,
 * group); byte r of the word = 1 iff replica r sent an AcceptReply to that entry.  An AcceptReply
 * always carries the ballot of the Accept it answers (multipaxos/durability.rs:108-131), so the cell
 * does not repeat it.  Wave-tiled: the word of my j-th outbox entry for group g lives at 64-bit index
 * ((g/64)*outbox_cap + j)*64 + g%64 (n_groups rounded up to a multiple of 64).  A host that receives
 * real AcceptReply messages (or the multi-GPU exchange) sets the bytes before R3, dropping replies
 * whose ballot is not the entry's. */
int smr_mp_ack_matrix(smr_mp_cluster *c, uint8_t rep, uint8_t **ack_dev, uint64_t *n_bytes);

/* The same acknowledgements as RECORDS: one per PeerMsg::AcceptReply { slot, ballot } that replica `peer` sent to
 * replica `rep` of `group` (multipaxos/messages.rs:370-443; emitted at durability.rs:108-131).  This is the form a host
 * with real sockets holds them in (what `smr_wire_*` decodes a frame to) and the form the multi-GPU exchange ships.
 *   smr_mp_deliver_acks  between R2 and R3: puts n device records into `rep`'s ack matrix, i.e. the batched
 *       "handle_msg_accept_reply was called with these" -- the handler's own filters (ballot == bal_prepared, status,
 *       duplicates; messages.rs:377-406) run in R3 as for locally produced acknowledgements.  A record that answers no
 *       Accept `rep` sent this tick (other ballot, slot not in the outbox, bad group / peer, peer == rep) is ignored and
 *       counted in *dropped_dev (device u64, may be NULL; the caller zeroes it) -- never an error.
 *   smr_mp_collect_acks  the inverse, after R2: every set cell of `rep`'s ack matrix as a record, in no particular
 *       order (arrival order is ackctl's business); *n_dev (device u64) = their number, of which at most `cap` are stored.
 *   smr_mp_clear_acks    zeroes `rep`'s ack matrix (a host that delivers records instead of running R2 for remote
 *       followers starts the tick's matrix from nothing).
 * All three only enqueue work on `stream`. */
typedef struct {
    uint32_t group, slot;
    uint64_t ballot;
    uint32_t peer;          /* the replica that accepted (ReplicaId) */
    uint32_t reserved;      /* 0 */
} smr_mp_ack;               /* 24 bytes */
int smr_mp_deliver_acks(smr_mp_cluster *c, uint8_t rep, const smr_mp_ack *acks_dev, uint64_t n, uint64_t *dropped_dev,
                        void *stream);
/* smr_mp_deliver_acks for records in one segment per connection, as smr_wire_ingest_mp_conn leaves them: connection c -- replica
 * conn_peer_dev[c] of group conn_group_dev[c] -- has cnt_dev[3 * c] 12-byte records { slot, ballot lo, ballot hi }
 * (smr_wire_ack12, declared with that call) from record conn_off_dev[c] / 13 on (records at or past ack_cap are not there).
 * conn_off_dev [n_conn + 1] and cnt_dev are the arrays of the SAME ingest call: a count that reaches past the connection's own
 * segment (conn_off[c + 1] / 13) or past ack_cap is cut there and the rest counted in *dropped_dev. */
int smr_mp_deliver_acks_conn(smr_mp_cluster *c, uint8_t rep, const void *acks12_dev, uint64_t ack_cap, const uint64_t *conn_off_dev,
                             const uint32_t *conn_group_dev, const uint8_t *conn_peer_dev, const uint32_t *cnt_dev, uint32_t n_conn,
                             uint64_t *dropped_dev, void *stream);
int smr_mp_collect_acks(smr_mp_cluster *c, uint8_t rep, smr_mp_ack *out_dev, uint64_t cap, uint64_t *n_dev, void *stream);
int smr_mp_clear_acks(smr_mp_cluster *c, uint8_t rep, void *stream);

/* --- read-back (host buffers; each call synchronizes the device) -------- */
int smr_mp_read_group_state(smr_mp_cluster *c, uint32_t group, uint8_t rep, smr_mp_group_state *out);

/* Whole-replica canonical dump, SoA over groups; slot arrays are [W][G]
 * indexed by slot % W, zero outside [start_slot, log_len).  flags: bit0
 * leader_bk present, bit1 replica_bk present, bit2 external. */
typedef struct {
    uint8_t *leader; uint64_t *bal_prep_sent, *bal_prepared, *bal_max_seen;
    uint32_t *start_slot, *log_len, *accept_bar, *commit_bar, *exec_bar, *snap_bar;
    uint32_t *peer_exec_bar;            /* [R][G] */
    uint64_t *s_bal; uint8_t *s_status; uint32_t *s_reqs; uint64_t *s_vbal; uint32_t *s_vreqs;
    uint8_t *s_flags, *s_acks, *s_packs; uint64_t *s_pmax; uint32_t *s_ltrig, *s_lendp;
    uint8_t *s_src; uint32_t *s_rtrig, *s_rendp;
    uint8_t *overflow;                  /* [G] */
} smr_mp_dump_bufs;
int smr_mp_dump(smr_mp_cluster *c, uint8_t rep, const smr_mp_dump_bufs *host_bufs);
/* the same for groups [g0, g0 + n) only (g0 a multiple of 64): host arrays are [n] / [R][n] / [W][n].  What a
 * parity check at BASELINE sizes uses: 65 536 groups run on the device, slices of them are compared against an
 * oracle started at that group offset (1.8 GB per replica would cross PCIe for a full dump at W = 512). */
int smr_mp_dump_range(smr_mp_cluster *c, uint8_t rep, uint32_t g0, uint32_t n, const smr_mp_dump_bufs *host_bufs);

/* counters of replica `rep`: [0] leader-side commits (Accepting->Committed,
 * messages.rs:412-433), [1] redirected batches (request.rs:128-154),
 * [2] batches refused by ring back-pressure */
int smr_mp_counters(smr_mp_cluster *c, uint8_t rep, uint64_t out[3]);

/* debug: client batches of replica `rep` that were handled by the generic per-lane path
 * instead of the steady-state fast path (a performance, not a correctness, figure) */
int smr_mp_debug_generic_units(smr_mp_cluster *c, uint8_t rep, uint64_t *out);

/* debug: client batches of replica `rep` that the quorum-tally launch of the PREVIOUS tick appended (smr_mp_run_ticks, round 6:
 * a leader's steady-state handle_req_batch calls of tick t + 1 -- multipaxos/request.rs:13-127 -- ride in the launch that
 * completes its tick t, where its scalars are in registers already) instead of that tick's own smr_mp_round_local launch
 * (a performance, not a correctness, figure; only with SMR_MP_FOLD_R1 in the environment at smr_mp_create: off by default,
 * measured a wash -- DESIGN.md 10) */
int smr_mp_debug_folded_batches(smr_mp_cluster *c, uint8_t rep, uint64_t *out);

/* the straggler list (smr_mp_cfg.straggler_ticks): out[0] = its capacity in groups (0: list off), out[1] = the number of
 * groups the LAST mark pass (of smr_mp_tick / of a smr_mp_run_ticks batch) wanted on it.  out[1] > out[0]: the list was
 * full and the groups beyond it stayed with the bulk kernels (same results, slower tick).  Synchronises the device. */
int smr_mp_straggler_stats(smr_mp_cluster *c, uint64_t out[2]);

/* debug: 64 wall-clock stamps (100 MHz) of the cooperative jobs' phases; all zero unless the library was
 * built with -DSMR_JOB_STAMPS (tools/dbg_stamps.py) */
int smr_mp_debug_stamps(smr_mp_cluster *c, uint64_t *out64);

/* Drain replica `rep`'s committed-slot list (leader-side commits since the
 * last poll) into host arrays.  Order: ascending commit order within a group;
 * unspecified across groups.  *n_out may exceed cap (entries were dropped). */
int smr_mp_poll_commits(smr_mp_cluster *c, uint8_t rep, uint32_t *groups_host, uint32_t *slots_host,
                        uint64_t cap, uint64_t *n_out);

/* Per-kernel device time accounting (HIP events on `stream` around each
 * round).  which: 0 = R1, 1 = R2, 2 = R3 (both kernels), 3 = R4,
 * 4 = mp_quorum_tally alone (the kernel the roofline is quoted on). */
int smr_mp_profile_enable(smr_mp_cluster *c, int on);
int smr_mp_profile_read(smr_mp_cluster *c, int which, double *total_ms, uint64_t *launches);

/* ------------------------------------------------------------------------
 * Raft replica over G groups (one replica per group)
 * leader side replaces: RaftReplica::handle_req_batch (raft/request.rs:70-90) log append,
 *           handle_msg_append_entries_reply (raft/messages.rs:222-388)
 * ---------------------------------------------------------------------- */
typedef struct smr_raft_leader smr_raft_leader;

typedef struct {
    uint32_t n_groups;
    uint8_t population;     /* R */
    uint8_t leader_id;      /* my replica id in every group */
    uint8_t commit_extra;   /* CRaft fault_tolerance (craft/messages.rs:301-313); 0 = Raft */
    uint8_t reserved0;
    uint32_t window;        /* W: ring of entry terms, power of two */
    uint64_t term;          /* curr_term at creation (role = Leader) */
} smr_raft_cfg;

int smr_raft_leader_create(const smr_raft_cfg *cfg, smr_raft_leader **out);
void smr_raft_leader_destroy(smr_raft_leader *l);
/* append n_new[g] entries of the current term per group (handle_req_batch) */
int smr_raft_leader_append(smr_raft_leader *l, const uint32_t *n_new_dev, void *stream);
/* One AppendEntriesReply per (peer, group): reply_term[R][G], end_slot[R][G],
 * conflict_term[R][G], conflict_slot[R][G], flags[R][G] (bit0 valid, bit1
 * conflict present).  Peers are processed in `order_dev[g]` order (ackctl
 * encoding; NULL = identity). */
int smr_raft_leader_handle_replies(smr_raft_leader *l, const uint64_t *reply_term_dev,
                                   const uint32_t *end_slot_dev, const uint64_t *conflict_term_dev,
                                   const uint32_t *conflict_slot_dev, const uint8_t *flags_dev,
                                   const uint32_t *order_dev, void *stream);
/* A BATCH of ticks in one call and -- per <= 16 ticks -- ONE launch: tick t = smr_raft_leader_append(n_new) followed by
 * smr_raft_leader_handle_replies(the tick's replies), exactly as the two calls would do it, the group's state kept in
 * registers from tick to tick (groups never talk to each other, so a lane runs its group's ticks back to back) and the next
 * tick's inputs requested while this tick's are worked on.  n_new == NULL: no appends in that tick; flags == NULL: no
 * replies in that tick.  Every array of every tick of the batch must stay untouched until the call's stream work is done.
 * Plain Raft leaders (SMR_ERR_STATE after smr_raft_craft_enable). */
typedef struct {
    const uint32_t *n_new;          /* [G] */
    const uint64_t *reply_term;     /* [R][G] */
    const uint32_t *end_slot;       /* [R][G] */
    const uint64_t *conflict_term;  /* [R][G] or NULL */
    const uint32_t *conflict_slot;  /* [R][G] or NULL */
    const uint8_t *flags;           /* [R][G] */
    const uint32_t *order;          /* [G] or NULL = identity */
} smr_raft_tick;
int smr_raft_leader_run_ticks(smr_raft_leader *l, const smr_raft_tick *ticks, uint32_t n_ticks, void *stream);
typedef struct {
    uint8_t *role; uint64_t *curr_term; uint32_t *log_len, *last_commit, *last_snap;
    uint32_t *next_slot, *try_next_slot, *match_slot;   /* [R][G] */
    uint64_t *entry_term;                               /* [W][G] by slot % W */
    uint8_t *leader; uint32_t *start_slot;
} smr_raft_dump_bufs;
int smr_raft_leader_dump(smr_raft_leader *l, const smr_raft_dump_bufs *host_bufs);
int smr_raft_leader_total_commits(smr_raft_leader *l, uint64_t *out);

/* ---- CRaft leader variant (src/protocols/craft/, a fork of raft/) ----------------------
 * smr_raft_craft_enable turns every group's leader of `l` into a CRaft leader (call it right after create):
 * handle_replies then follows craft/messages.rs:256-404 -- every reply taken counts as a heard heartbeat
 * (Heartbeater::update_heard_cnt, server/heartbeat.rs:280-296), success replies are not tested for staleness, and an
 * entry commits at `majority + fault_tolerance` matches, `majority` while the group is in full-copy mode (:307-313).
 * The leader created every entry of its log, so it holds all shards and every committed entry is executable.
 * repeat_threshold = hear_timeout_min / send_interval of the Heartbeater (heartbeat.rs:257-259). */
int smr_raft_craft_enable(smr_raft_leader *l, uint8_t fault_tolerance, uint8_t repeat_threshold);
/* CRaftReplica::bcast_heartbeats on the send tick (craft/leadership.rs:249-291): the empty AppendEntries per peer
 * (hb_flags[R][G] 1 = sent; prev_slot, prev_term [R][G]; leader_commit, last_snap [G]), update_bcast_cnts
 * (heartbeat.rs:240-276), then fall back to full-copy mode if `population - alive >= fault_tolerance` (:283-288). */
int smr_raft_craft_bcast_heartbeats(smr_raft_leader *l, uint8_t *hb_flags_dev, uint32_t *prev_slot_dev, uint64_t *prev_term_dev,
                                    uint32_t *leader_commit_dev, uint32_t *last_snap_dev, void *stream);
/* switch_assignment_mode (craft/leadership.rs:80-141): to_full[G] 0 / 1, anything else = no call for the group */
int smr_raft_craft_switch_assignment_mode(smr_raft_leader *l, const uint8_t *to_full_dev, void *stream);
/* Shard masks of a new entry's RS codeword (smr_rs_encode makes the shards): persist[G] = what the leader's WAL entry holds
 * (craft/request.rs:86-100), send[R][G] = what an AppendEntries to each peer carries (craft/durability.rs:41-80,
 * messages.rs:416-460): the data shards 0..majority in full-copy mode, else the receiver's own shard. */
int smr_raft_craft_assignment(smr_raft_leader *l, uint32_t *persist_dev, uint32_t *send_dev, void *stream);
/* host arrays: full_copy_mode[G], peer_alive[G] (bitmap), reply_cnts .0 / .1 / .2 as [R][G] */
int smr_raft_craft_dump(smr_raft_leader *l, uint8_t *full_copy_host, uint8_t *alive_host, uint64_t *hb_replied_host,
                        uint64_t *hb_seen_host, uint8_t *hb_repeat_host);

/* ---- Raft follower side and elections, same replica object ---------------------------
 * One replica per group; `l` is the object created above (role Leader at creation;
 * smr_raft_replica_preset puts every group's replica into another role for a scenario). */
int smr_raft_replica_preset(smr_raft_leader *l, uint8_t role /* 0 Follower 1 Candidate 2 Leader */, uint8_t leader,
                            uint64_t term, uint8_t voted_for /* SMR_NO_REPLICA = None */);

/* One AppendEntries message per group (device arrays [G]; entry_term[k][G], k < max_entries).
 * flags bit0 = a message is present.  entry_mask[k][G] (CRaft only, else NULL): avail_shards_map of the k-th entry's
 * codeword, bit i = shard i (data shards 0 .. majority-1). */
typedef struct {
    const uint8_t *flags, *leader;
    const uint64_t *term;
    const uint32_t *prev_slot;
    const uint64_t *prev_term;
    const uint32_t *n_entries;
    const uint64_t *entry_term;
    uint32_t max_entries;
    const uint32_t *leader_commit, *last_snap;
    const uint8_t *entry_mask;
} smr_raft_append_entries;
/* The AppendEntriesReply each group's replica sends (device arrays [G]): flags bit0 = a
 * reply is sent, bit1 = it carries `conflict`. */
typedef struct {
    uint8_t *flags;
    uint64_t *term;
    uint32_t *end_slot;
    uint64_t *conflict_term;
    uint32_t *conflict_slot;
} smr_raft_append_reply;
/* RaftReplica::handle_msg_append_entries (raft/messages.rs:13-218) followed by the WAL
 * completions of the appended entries (handle_logged_follower_append, durability.rs:97-132). */
int smr_raft_replica_handle_append_entries(smr_raft_leader *l, const smr_raft_append_entries *msg_dev,
                                           const smr_raft_append_reply *reply_dev, void *stream);

/* become_a_candidate (raft/leadership.rs:76-142) on HearTimeout events: timeout_src[g] =
 * replica the timer was about, SMR_NO_REPLICA = no event.  rv_flags bit0 = RequestVote
 * {rv_term, rv_last_slot, rv_last_term} is broadcast. */
int smr_raft_replica_become_candidate(smr_raft_leader *l, const uint8_t *timeout_src_dev, uint8_t *rv_flags_dev,
                                      uint64_t *rv_term_dev, uint32_t *rv_last_slot_dev, uint64_t *rv_last_term_dev,
                                      void *stream);
/* handle_msg_request_vote (raft/messages.rs:391-482).  r_flags bit0 = a RequestVoteReply is
 * sent (the reference sends none when it neither refuses by term nor grants), bit1 = granted. */
int smr_raft_replica_handle_request_vote(smr_raft_leader *l, const uint8_t *flags_dev, const uint8_t *candidate_dev,
                                         const uint64_t *term_dev, const uint32_t *last_slot_dev,
                                         const uint64_t *last_term_dev, uint8_t *r_flags_dev, uint64_t *r_term_dev,
                                         void *stream);
/* handle_msg_request_vote_reply (raft/messages.rs:485-510) for one reply per (peer, group):
 * term[R][G], flags[R][G] bit0 = present, peers taken in order_dev[g] order (ackctl encoding).
 * At a quorum: become_the_leader (leadership.rs:145-179); hb_prev_slot[p][g] = prev_slot of the
 * heartbeat it computes for peer p (bcast_heartbeats, :182-218), 0xFFFFFFFF if not elected here. */
int smr_raft_replica_handle_vote_replies(smr_raft_leader *l, const uint64_t *term_dev, const uint8_t *flags_dev,
                                         const uint32_t *order_dev, uint32_t *hb_prev_slot_dev, uint8_t *elected_dev,
                                         void *stream);
/* smr_raft_leader_append that also reports, per peer p, the first slot of the entries its appends sent to p
 * (first_sent[p][g]; 0xFFFFFFFF = nothing: handle_logged_leader_append, raft/durability.rs:28-88) */
int smr_raft_leader_append_emit(smr_raft_leader *l, const uint32_t *n_new_dev, uint32_t *first_sent_dev, void *stream);
/* Fills `msg` (the arrays are written) with the AppendEntries for one peer: the entries
 * [first[g], min(first[g] + max_entries, log end)) as ONE message per group (the reference sends one per
 * appended batch; a follower that handles them in order ends in the same state). */
int smr_raft_leader_gather_entries(smr_raft_leader *l, const uint32_t *first_dev, const smr_raft_append_entries *msg,
                                   void *stream);
/* A leader's AppendEntries for n <= 8 followers that live on its device, and the followers' handlers, in ONE launch: per k,
 * smr_raft_leader_gather_entries(leader, first_dev[k], &msgs[k]) followed by
 * smr_raft_replica_handle_append_entries(followers[k], &msgs[k], &replies[k]) -- the state, the messages (written: a checker or a
 * wire emitter may read them) and the replies are those of the 2 n calls (the fan-out of raft/durability.rs:57-80 and
 * handle_msg_append_entries, raft/messages.rs:13-218 / craft/messages.rs:14-254, of a co-located cluster).  first_dev, msgs,
 * replies: host arrays of n; the messages share max_entries; leader and followers share groups / window / population and are all
 * Raft or all CRaft replicas (CRaft: msgs[k].entry_mask = what follower k is sent). */
int smr_raft_cluster_replicate(smr_raft_leader *leader, uint32_t n, smr_raft_leader *const *followers, const uint32_t *const *first_dev,
                               const smr_raft_append_entries *msgs, const smr_raft_append_reply *replies, void *stream);
/* A co-located cluster's steady tick in ONE launch (round 6): smr_raft_leader_append_emit(leader, n_new_dev, first_sent_dev), then
 * smr_raft_cluster_replicate(leader, n, followers, first_dev, msgs, replies), then smr_raft_leader_handle_replies(leader, reply_term_dev,
 * end_slot_dev, conflict_term_dev, conflict_slot_dev, flags_dev, order_dev) -- handle_req_batch + the fan-out of raft/durability.rs:28-88,
 * the followers' handle_msg_append_entries (raft/messages.rs:13-218, craft/messages.rs:14-254) and the leader's
 * handle_msg_append_entries_reply (raft/messages.rs:222-335, craft/messages.rs:256-404) -- with the state, messages, replies and counters
 * of the three calls.  A block owns 64 groups: the leader's wavefront, one wavefront per follower, a block barrier between the steps.
 * For the replies to be this tick's, first_dev[k] points at row (follower k's id) of first_sent_dev and replies[k] at that row of
 * the [R][G] reply arrays (any other arrangement is taken as given: the three calls would read the same memory). */
int smr_raft_cluster_tick(smr_raft_leader *leader, const uint32_t *n_new_dev, uint32_t *first_sent_dev, uint32_t n, smr_raft_leader *const *followers,
                          const uint32_t *const *first_dev, const smr_raft_append_entries *msgs, const smr_raft_append_reply *replies,
                          const uint64_t *reply_term_dev, const uint32_t *end_slot_dev, const uint64_t *conflict_term_dev,
                          const uint32_t *conflict_slot_dev, const uint8_t *flags_dev, const uint32_t *order_dev, void *stream);
int smr_raft_replica_dump_votes(smr_raft_leader *l, uint8_t *voted_for_host, uint8_t *votes_host, uint32_t *n_exec_host,
                                uint32_t *n_trunc_host);
/* The CRaft FOLLOWER (after smr_raft_craft_enable): smr_raft_replica_handle_append_entries then follows the fork's
 * handle_msg_append_entries (craft/messages.rs:14-254: consistency check on heartbeats too, shards of a re-sent entry
 * absorbed, execution only with `majority` shards and after reconstruct_data when too few are data shards) and needs
 * smr_raft_append_entries.entry_mask.  handle_msg_reconstruct (craft/messages.rs:622-663): n[g] asked (slot, term) pairs
 * [max_slots][G] -> r_has / r_mask [max_slots][G] (the codeword of every slot held under that term, as its bitmap),
 * r_n[g] of them (0: no ReconstructReply).  dump: the entries' bitmaps [W][G] by slot % W, counters[2] = reconstruct_data
 * calls, executions postponed for lack of shards.
 * A leader that was a follower once holds entries without every shard: smr_raft_leader_handle_replies then executes only
 * what it has `majority` shards of and queues the slots it lacks (craft/messages.rs:315-358); smr_raft_craft_poll_reconstructs
 * hands over (and clears) that queue -- the Reconstruct { slots } to broadcast: n[g] <= max_slots <= 16, slot / term
 * [max_slots][G].  smr_raft_craft_handle_reconstruct_reply: handle_msg_reconstruct_reply (:665-745) for the ReconstructReply of
 * peer[g] (SMR_NO_REPLICA: none): n[g] (slot, bitmap) pairs [max_slots][G]. */
int smr_raft_craft_handle_reconstruct(smr_raft_leader *l, const uint32_t *n_dev, const uint32_t *slot_dev, const uint64_t *term_dev,
                                      uint32_t max_slots, uint32_t *r_n_dev, uint8_t *r_has_dev, uint8_t *r_mask_dev, void *stream);
int smr_raft_craft_poll_reconstructs(smr_raft_leader *l, uint32_t max_slots, uint32_t *n_dev, uint32_t *slot_dev, uint64_t *term_dev, void *stream);
int smr_raft_craft_handle_reconstruct_reply(smr_raft_leader *l, const uint8_t *peer_dev, const uint32_t *n_dev, const uint32_t *slot_dev,
                                            const uint8_t *mask_dev, uint32_t max_slots, void *stream);
int smr_raft_craft_dump_masks(smr_raft_leader *l, uint8_t *mask_host, uint64_t *counters);
/* How many entries of AppendEntries messages this replica's follower path skipped because they had left its W-entry term ring
 * (it takes such an entry as matching; raft/messages.rs:128-140 compares terms in an unbounded Vec).  A harness rule for
 * bounded memory, shared with the oracle: size `window` so that this stays 0 -- the parity tests assert it. */
int smr_raft_ring_guard_hits(smr_raft_leader *l, uint64_t *out);

/* ------------------------------------------------------------------------
 * EPaxos command leader / acceptor over G groups (one replica id per group)
 * replaces: EPaxosReplica::handle_req_batch (epaxos/request.rs:10-108) with
 *           identify_deps / refresh_highest_cols / max_seq_num (dependency.rs:85-167),
 *           handle_msg_pre_accept (messages.rs:10-93), handle_msg_pre_accept_reply
 *           (:96-270) with fast_quorum_eligibility (dependency.rs:175-240),
 *           handle_msg_accept (:273-345), handle_msg_accept_reply (:348-436), the
 *           WAL completions and commit bars (durability.rs:10-135).
 * A request batch is one Put on key id < n_keys (SMR_EP_NO_KEY = nothing proposed / empty batch).
 * DepSets are R uint32 per group ([R][G]), SMR_EP_NONE = Option::None.
 * ---------------------------------------------------------------------- */
typedef struct smr_ep_replica smr_ep_replica;
#define SMR_EP_NONE 0xFFFFFFFFu
#define SMR_EP_NO_KEY 0xFFu

typedef struct {
    uint32_t n_groups;
    uint8_t population;        /* R, 3..8 */
    uint8_t me;                /* my replica id (= my row of the instance space) in every group */
    uint8_t optimized_quorum;  /* ReplicaConfigEPaxos::optimized_quorum (mod.rs:59,91): super quorum F + ceil(F/2) */
    uint8_t execute;           /* 1: run attempt_execution (execution.rs:25-149) behind every handler call, see below */
    uint32_t window;           /* W: columns kept per row, power of two */
    uint32_t n_keys;           /* key space per group, 1..255 */
    uint32_t recovery;         /* 1: explicit prepare (below); leader bookkeeping for every row (may be combined with execute) */
} smr_ep_cfg;

int smr_ep_replica_create(const smr_ep_cfg *cfg, smr_ep_replica **out);
void smr_ep_replica_destroy(smr_ep_replica *e);

/* One PreAccept / Accept message (or a reply to one) per group: device arrays [G], deps [R][G].
 * flags bit0 = present.  `peer` = sender; `row` = the row of `col` where that is not the sender's (an instance under
 * explicit prepare, needs recovery = 1), NULL = the sender's row.  Replies leave peer / col / key / row unused. */
typedef struct {
    uint8_t *flags;
    uint8_t *peer;
    uint32_t *col;
    uint64_t *ballot;
    uint64_t *seq;
    uint32_t *deps;
    uint8_t *key;
    uint8_t *row;
} smr_ep_msg;

/* handle_req_batch for key_dev[g] (+ my own PreAcceptSlot completion = my own reply); `out` receives the
 * PreAccept that is broadcast (flags, col, seq, deps; ballot is make_default_ballot(me) = me + 1).
 * exploded_dev (may be NULL): bit p = peer p's hear timer has exploded (dependency.rs:205-211). */
int smr_ep_propose(smr_ep_replica *e, const uint8_t *key_dev, const uint8_t *exploded_dev, const smr_ep_msg *out,
                   void *stream);
/* handle_msg_pre_accept + WAL completion: reply = PreAcceptReply {ballot, seq, deps} (flags bit0 = sent) */
int smr_ep_handle_pre_accept(smr_ep_replica *e, const smr_ep_msg *msg, const smr_ep_msg *reply, void *stream);
/* handle_msg_accept + WAL completion: reply = AcceptReply {ballot} */
int smr_ep_handle_accept(smr_ep_replica *e, const smr_ep_msg *msg, const smr_ep_msg *reply, void *stream);
/* handle_msg_commit_notice (epaxos/messages.rs:438-508) + its CommitSlot completion (commit bar) */
int smr_ep_handle_commit_notice(smr_ep_replica *e, const smr_ep_msg *msg, void *stream);
/* The PreAcceptReplies to my instance (me, col[g]): ballot / seq / flags [R][G], deps [R][R][G]
 * (peer, dep row, group); ballot 0 = the "failure suspected" re-evaluation call; peers in
 * order_dev[g] order (ackctl encoding, NULL = identity).  decision[g]: 0 = undecided,
 * 3 = Committed on the fast path, 2 = Accepting (slow path) with (d_seq, d_deps [R][G]). */
int smr_ep_handle_pre_accept_replies(smr_ep_replica *e, const uint32_t *col_dev, const uint64_t *ballot_dev,
                                     const uint64_t *seq_dev, const uint32_t *deps_dev, const uint8_t *flags_dev,
                                     const uint32_t *order_dev, const uint8_t *exploded_dev, uint8_t *decision_dev,
                                     uint64_t *d_seq_dev, uint32_t *d_deps_dev, void *stream);
/* The AcceptReplies to my instance (me, col[g]): ballot / flags [R][G]; committed[g] = 1 if it commits here */
int smr_ep_handle_accept_replies(smr_ep_replica *e, const uint32_t *col_dev, const uint64_t *ballot_dev,
                                 const uint8_t *flags_dev, const uint32_t *order_dev, uint8_t *committed_dev, void *stream);
/* ---- explicit prepare (smr_ep_cfg.recovery = 1): recovery of a suspected peer's row --------------------------------------
 * replaces: EPaxosReplica::heartbeat_timeout (epaxos/heartbeat.rs:17-125, the protocol part), handle_msg_exp_prepare
 *           (messages.rs:511-574), handle_msg_exp_prepare_reply (:577-821) with exp_prepare_next_step (dependency.rs:249-327).
 * exp_prepare_voteds is a HashMap in the reference and next_step takes the last reply of a Status in iteration order as its
 * representative (dependency.rs:266-273); here: peer-id order, i.e. the highest peer id.
 * smr_ep_heartbeat_timeout: HearTimeout { peer: src[g] } (SMR_NO_REPLICA: none).  Re-evaluates the fast quorum of every
 *   PreAccepting instance I lead (a PreAcceptReply with ballot 0 from src, exploded_dev as in smr_ep_propose), then starts
 *   ExpPrepare for every in-progress instance of src's row and handles my own ExpPrepareReply to each.  Out: the
 *   ExpPrepare { slot: (src, col), new_ballot } broadcasts in column order, n[g] of them, col / ballot as [W][G].
 * smr_ep_handle_exp_prepare: one ExpPrepare per group -> the ExpPrepareReply (flags bit0 = sent).
 * smr_ep_handle_exp_prepare_replies: the replies [R][G] (voted_deps [R][R][G]) to my ExpPrepare of (row[g], col[g]) under
 *   new_ballot[R][G], peers in order_dev[g] order; decision[g] = the Status of what is broadcast here -- 3: CommitNotice,
 *   2: Accept, 1: PreAccept (the instance then avoids the fast path) -- with its ballot / seq / deps / key; 0: nothing yet.
 * The Accept / PreAccept rounds that follow go through the handlers above with `row` set (smr_ep_msg.row, *_replies_at). */
typedef struct {
    uint8_t *flags, *peer, *row;
    uint32_t *col;
    uint64_t *new_ballot;
} smr_ep_exp_prepare;
typedef struct {
    uint8_t *flags;            /* as a message: [G]; as the replies handed to smr_ep_handle_exp_prepare_replies: [R][G] */
    uint64_t *voted_bal;
    uint8_t *voted_status;
    uint64_t *voted_seq;
    uint32_t *voted_deps;      /* [R][G] resp. [R][R][G] */
    uint8_t *voted_key;
} smr_ep_exp_prepare_reply;
int smr_ep_heartbeat_timeout(smr_ep_replica *e, const uint8_t *src_dev, const uint8_t *exploded_dev, uint32_t *n_dev, uint32_t *col_dev,
                             uint64_t *ballot_dev, void *stream);
int smr_ep_handle_exp_prepare(smr_ep_replica *e, const smr_ep_exp_prepare *msg, const smr_ep_exp_prepare_reply *reply, void *stream);
int smr_ep_handle_exp_prepare_replies(smr_ep_replica *e, const uint8_t *row_dev, const uint32_t *col_dev, const uint64_t *new_ballot_dev,
                                      const smr_ep_exp_prepare_reply *replies, const uint32_t *order_dev, uint8_t *decision_dev,
                                      uint64_t *d_ballot_dev, uint64_t *d_seq_dev, uint32_t *d_deps_dev, uint8_t *d_key_dev, void *stream);
/* the reply handlers for an instance I lead outside my own row (row_dev NULL = my row: the calls above) */
int smr_ep_handle_pre_accept_replies_at(smr_ep_replica *e, const uint8_t *row_dev, const uint32_t *col_dev,
                                        const uint64_t *ballot_dev, const uint64_t *seq_dev, const uint32_t *deps_dev,
                                        const uint8_t *flags_dev, const uint32_t *order_dev, const uint8_t *exploded_dev,
                                        uint8_t *decision_dev, uint64_t *d_seq_dev, uint32_t *d_deps_dev, void *stream);
int smr_ep_handle_accept_replies_at(smr_ep_replica *e, const uint8_t *row_dev, const uint32_t *col_dev, const uint64_t *ballot_dev,
                                    const uint8_t *flags_dev, const uint32_t *order_dev, uint8_t *committed_dev, void *stream);
/* ---- one tick of a co-located EPaxos cluster as ONE call ------------------------------------------------------------
 * The closed loop a host runs when all R replicas of its groups live on this device (BASELINE config 5 on one GPU; what
 * summerset_amd/ep_cluster.py drives handler by handler): every replica r proposes keys_dev[r][g] (0xFF: nothing), the
 * PreAccepts go to all peers (senders ascending at every acceptor), every command leader tallies its PreAcceptReplies
 * (handle_msg_pre_accept_reply, epaxos/messages.rs:96-270), sends Accepts where it took the slow path (the round always
 * runs, its flags zero elsewhere), tallies the AcceptReplies (:348-436) and sends CommitNotices -- leaders ascending.  The
 * same kernels the per-handler entry points launch, back to back on `stream`, the peers' replies written straight into the
 * leader's stacked reply arrays: no host work between them.  drop_dev (may be NULL): [R * R] pointers, entry s * R + q
 * (may be NULL) = u8 [G], 1 where the PreAccept from s to q is lost (with its reply).  out[s], device arrays the caller
 * owns: what leader s proposed (proposed = flags, col, seq0 / deps0 [R][G] = the PreAccept's) and decided (decision 0 /
 * 2 Accepting / 3 Committed on the fast path, committed = 1 where the instance is committed after the tick, seq, deps
 * [R][G] of the decision).  Replicas: created with me = index, population = n, equal groups; they stay the caller's.
 * While the cluster lives (populations <= 5) its replicas keep their per-key tables -- a key's highest columns, the executor's
 * KV word -- in ONE table of the cluster's, a 128-byte line per (group, key) with a slot per replica: every call on such a
 * replica, its own handlers and dumps included, works there; smr_ep_cluster_destroy hands the entries back to the replicas'
 * own tables, a replica destroyed first leaves its seat empty (smr_ep_cluster_tick then answers SMR_ERR_STATE). */
typedef struct smr_ep_cluster smr_ep_cluster;
typedef struct {
    uint8_t *proposed;
    uint32_t *col;
    uint64_t *seq0;
    uint32_t *deps0;
    uint8_t *decision, *committed;
    uint64_t *seq;
    uint32_t *deps;
} smr_ep_cluster_out;
int smr_ep_cluster_create(smr_ep_replica *const *reps, uint32_t n, smr_ep_cluster **out);
void smr_ep_cluster_destroy(smr_ep_cluster *c);
int smr_ep_cluster_tick(smr_ep_cluster *c, const uint8_t *const *keys_dev, const uint8_t *const *drop_dev, const smr_ep_cluster_out *out,
                        void *stream);
/* How smr_ep_cluster_tick runs the tick.  0 (default): ONE launch -- a block is the R replicas (one wavefront each) of 64 groups,
 * the handlers run as steps of one kernel, messages cross wavefronts behind block barriers, execution runs behind its handler
 * on the same lane.  1: one launch per handler and (replica, sender) pair, the kernels of the per-handler entry points back to
 * back (round 2's path: 115 launches per tick at R = 5 with execution on); kept as the decomposition the one-launch tick is
 * checked against and timed beside.  Both are the handler-by-handler loop bit for bit.
 * Bit 1 of `mode` (2: one launch, 3: launch by launch) orders the command leaders' part of the tick PHASE BY PHASE -- every
 * leader's PreAcceptReplies, then every Accept round, every AcceptReply tally, every CommitNotice (an acceptor still takes the
 * senders in ascending order) -- instead of leader by leader: another legal delivery order of the same messages
 * (epaxos/messages.rs handles them one at a time in whatever order the transport delivers), in which all R replicas of a group
 * work in every step of the one-launch kernel.  Same commits and decisions; with execution on, the executors' attempt order (and
 * with it the counters of abandoned attempts) differs from modes 0 / 1.  The loops `ep_cluster.tick(.., phase_major=True)` and
 * tests/ep_cluster.py run that order for the engines and for the oracle. */
int smr_ep_cluster_set_mode(smr_ep_cluster *c, uint32_t mode);
/* Round 6: with mode 2 (one launch, phase by phase) on a cluster of <= 5 replicas without explicit prepare, a lane's four
 * PreAccepts (epaxos/messages.rs:10-93) and its four CommitNotices with their executions (messages.rs:438-508,
 * durability.rs:104-160, execution.rs:25-211) each run as ONE batched step -- one round of loads for the whole phase -- where
 * every message of the phase is the common case (a PreAccept for a fresh cell at its row's end; a CommitNotice for a cell that
 * holds the PreAccepted instance whose dependencies are all executed); any other lane runs the phase's handlers one by one.
 * Same results bit for bit.  out[0] / out[1] = (replica, group, tick) lanes that ran their PreAccept / CommitNotice phase one
 * by one since the replicas were created (each wraps at 2^32). */
int smr_ep_cluster_batch_stats(smr_ep_cluster *c, uint64_t out[2]);

/* ---- layout L2 of the EPaxos cluster as ONE call per tick (round 6; BASELINE config 5 as written: the groups block-partitioned
 * over `world` ranks, replica r of block b on rank (b + r) mod world) ----------------------------------------------------
 * What summerset_amd/spread_ep.py drove from Python in rounds 3-5 -- the handlers of smr_ep_cluster_tick's launch-by-launch
 * mode cut at the five points where a message crosses replicas, the messages packed into one send buffer per exchange, one
 * all-to-all with static split sizes per exchange (TransportHub::send_msg / bcast_msg, server/transport.rs:208-275) -- with the
 * schedule, the message plan, the packing and the exchanges inside the library.
 *   smr_ep_spread_create   reps[i] = replica rep_id[i] of block rep_block[i], every one that lives on `rank` (created with
 *                          me = rep_id[i], n_groups = block_groups[rep_block[i]]); block_groups[world]; ordered = 0: 5 exchanges per
 *                          tick, all command leaders tally together (the co-located loop's phase-by-phase order, mode 2 of
 *                          smr_ep_cluster_set_mode); 1: 2 + 3 R exchanges, leader by leader (modes 0 / 1, execution state included).
 *   smr_ep_spread_segment  the compute between exchange seg - 1 and exchange seg (seg = n_exchanges: behind the last), for a host
 *                          that moves the buffers itself (smr_ep_spread_buffers: exchange k's send / receive buffer and the bytes
 *                          to / from every rank -- a gloo job, a test); segments come in order, smr_ep_spread_abort_tick reopens.
 *   smr_ep_spread_tick     all of it: segments and smr_comm_exchange calls back to back on `stream` (smr_ep_spread_bind_comm
 *                          first; a job of one rank needs none).
 * keys_dev[i] / out[i]: replica i's proposals (u8 [G], 0xFF: none) and its results as a command leader, as in smr_ep_cluster_tick;
 * drop_dev (may be NULL): [n_reps * R] pointers, entry i * R + q (may be NULL) = u8 [G], 1 where replica i's PreAccept to q is
 * lost (with its reply).  smr_ep_spread_info: out = {exchanges per tick, bytes put into send buffers so far}. */
typedef struct smr_ep_spread smr_ep_spread;
int smr_ep_spread_create(smr_ep_replica *const *reps, const uint32_t *rep_block, const uint8_t *rep_id, uint32_t n_reps,
                         const uint32_t *block_groups, uint32_t world, uint32_t rank, uint8_t population, int ordered, smr_ep_spread **out);
void smr_ep_spread_destroy(smr_ep_spread *s);
int smr_ep_spread_n_exchanges(const smr_ep_spread *s);
int smr_ep_spread_buffers(smr_ep_spread *s, uint32_t exchange, void **send_dev, uint64_t *send_bytes, void **recv_dev, uint64_t *recv_bytes);
int smr_ep_spread_bind_comm(smr_ep_spread *s, smr_comm *comm);
int smr_ep_spread_segment(smr_ep_spread *s, uint32_t seg, const uint8_t *const *keys_dev, const uint8_t *const *drop_dev,
                          const smr_ep_cluster_out *out, void *stream);
int smr_ep_spread_abort_tick(smr_ep_spread *s);
int smr_ep_spread_tick(smr_ep_spread *s, const uint8_t *const *keys_dev, const uint8_t *const *drop_dev, const smr_ep_cluster_out *out,
                       void *stream);
int smr_ep_spread_info(const smr_ep_spread *s, uint64_t out[2]);

/* host buffers [R][W][G] by col % W like smr_ep_dump: exp_prepare_acks, exp_prepare_max_bal, avoid_fast_path, the peers with
 * an entry in exp_prepare_voteds (bitmap); those entries [R][W][R][G], deps [R][W][R][R][G]; counters[4] = decisions
 * Committed, Accepting, PreAccepting with a command, PreAccepting as a no-op */
int smr_ep_xp_dump(smr_ep_replica *e, uint8_t *acks, uint64_t *max_bal, uint8_t *avoid, uint8_t *has, uint8_t *vstatus, uint64_t *vseq,
                   uint8_t *vkey, uint32_t *vdeps, uint64_t *counters);

/* host buffers: len / commit_bars [R][G]; per instance [R][W][G] by col % W (cells outside the last W
 * columns of a row read as null); deps [R][W][G][R]; highest_cols [n_keys][R][G]; counters[3] =
 * fast-path commits, slow-path entries, slow-path commits */
typedef struct {
    uint32_t *len, *commit_bars;
    uint64_t *bal, *seq;
    uint8_t *status, *key;
    uint32_t *deps;
    uint8_t *pa_acks, *acc_acks, *bk;
    uint32_t *highest_cols;
    uint64_t *counters;
} smr_ep_dump_bufs;
int smr_ep_dump(smr_ep_replica *e, const smr_ep_dump_bufs *host_bufs);

/* Dependency-graph execution (smr_ep_cfg.execute = 1; needs population * window <= 32768).  Every entry
 * point above then launches a second kernel behind its own that does what handle_logged_commit_slot
 * does once a row's commit bar has moved (durability.rs:136-160): attempt_execution on the row's new
 * tail (execution.rs:25-149: the walk over deps, the graph whose edges join consecutively popped
 * slots, its components in tarjan_scc's order), the re-attempts on the rows whose tail is still
 * Committed, and then handle_cmd_result (:152-211) for every submitted command in submission order:
 * Executing -> Executed, exec bars.  The state machine is one Put per instance: kv[key] = token,
 * token(row, col) = (row + 1) << 32 | col, result = the old token; digest chains
 * d = (d ^ token) * 0x100000001B3, d = (d ^ old) * 0x100000001B3 over the submissions of a group.
 * Host buffers: exec_bars [R][G], kv [n_keys][G], digest [G], counters[6] = commands submitted, of them
 * re-submissions of an already executing instance, pops of an instance that left the ring (counted
 * as executed), 0, attempts, abandoned attempts.
 * The device keeps a key's KV word in 32 bits (4 of row + 1, 28 of the column): once an instance has been executed at a column
 * >= 2^28 this call and smr_ep_exec_poll answer SMR_ERR_STATE (the buffers are still filled) -- tokens that are not the
 * reference's are never handed out silently. */
int smr_ep_exec_dump(smr_ep_replica *e, uint32_t *exec_bars, uint64_t *kv, uint64_t *digest, uint64_t *counters);
/* the commands the LAST handler call submitted to the state machine (state_machine.submit_cmd, execution.rs:113-131),
 * group-major, in submission order within a group: instance (row, col) of group -- the host applies them in this
 * order.  *n_out = how many there were; the first `cap` are written and the list is consumed; with NULL host arrays
 * the call only counts and leaves the list in place. */
int smr_ep_exec_poll(smr_ep_replica *e, uint32_t *group_host, uint8_t *row_host, uint32_t *col_host, uint64_t cap, uint64_t *n_out);

/* ------------------------------------------------------------------------
 * RSPaxos replica: G groups, one replica id per object, one call = one handler
 * of RSPaxosReplica per group (+ the WAL / command completions it triggers, LS-1 rule 0).
 *   smr_rsp_req_batch              handle_req_batch (rspaxos/request.rs:10-151): from_data + compute_parity,
 *                                  one shard per peer (subset_copy, rscoding.rs:255-293)
 *   smr_rsp_handle_accept          handle_msg_accept (messages.rs:343-403) + handle_logged_accept_data
 *   smr_rsp_handle_accept_replies  handle_msg_accept_reply (:406-464): threshold majority + fault_tolerance;
 *                                  handle_logged_commit_slot (durability.rs:125-186): the commit-bar run stops
 *                                  at an instance holding fewer than `majority` shards; handle_cmd_result
 *   smr_rsp_become_leader          become_a_leader on HearTimeout (leadership.rs:47-185): step-up Heartbeat,
 *                                  Prepare, the Reconstruct list (committed instances short of shards)
 *   smr_rsp_handle_prepare         handle_msg_prepare (:12-84) + the PrepareReply batch (voted ballot, shards)
 *   smr_rsp_handle_prepare_replies handle_msg_prepare_reply (:87-340): shards of the highest voted ballot
 *                                  merged (absorb_other, rscoding.rs:296-346); at the quorum, instances with
 *                                  >= majority shards are reconstructed and re-Accepted, at >= population -
 *                                  fault_tolerance replies the others become empty batches
 *   smr_rsp_handle_reconstruct / _reply   handle_msg_reconstruct (:467-515), handle_msg_reconstruct_reply (:518-594)
 *   smr_rsp_handle_heartbeat / smr_rsp_bcast_heartbeat   heard_heartbeat, bcast_heartbeats (leadership.rs:187-340)
 * A request batch is an opaque 32-bit token (0 = the empty batch, SMR_RSP_NULL = none / a null codeword); a
 * codeword is (token, mask of shards present): the shard BYTES are smr_rs_encode / smr_rs_reconstruct's.
 * Device arrays: one entry per group [G]; lists are [W][G] with a count [G]; per-peer matrices [R][G].
 * ---------------------------------------------------------------------- */
typedef struct smr_rsp_replica smr_rsp_replica;
#define SMR_RSP_NULL 0xFFFFFFFFu

typedef struct {
    uint32_t n_groups;
    uint8_t population;        /* R, 3..8; RS(majority, R - majority) */
    uint8_t me;                /* my replica id in every group */
    uint8_t fault_tolerance;   /* ReplicaConfigRSPaxos::fault_tolerance (mod.rs:75,600-604) */
    uint8_t reserved0;
    uint32_t window;           /* W: slots kept per group, power of two >= 8 */
} smr_rsp_cfg;

typedef struct { uint32_t *n, *slot, *val; uint64_t *ballot; } smr_rsp_accepts;            /* Accepts to every peer (shard {peer} each) */
typedef struct { uint8_t *flags; uint64_t *ballot; uint32_t *commit_bar, *exec_bar, *snap_bar; } smr_rsp_heartbeat;
typedef struct { uint32_t *n, *trig, *endp; uint64_t *ballot, *vbal; uint32_t *vval; uint8_t *vmask; } smr_rsp_prepare_reply;
typedef struct { uint32_t *n, *slot; uint64_t *bal; uint32_t *val; uint8_t *mask; } smr_rsp_shards;   /* ReconstructReply rows */

int smr_rsp_replica_create(const smr_rsp_cfg *cfg, smr_rsp_replica **out);
void smr_rsp_replica_destroy(smr_rsp_replica *e);
/* every replica believes in `leader`, which is prepared at make_unique_ballot(1) (host call) */
int smr_rsp_preset_leader(smr_rsp_replica *e, uint8_t leader);
int smr_rsp_req_batch(smr_rsp_replica *e, const uint32_t *val_dev, const smr_rsp_accepts *out, void *stream);
int smr_rsp_handle_accept(smr_rsp_replica *e, const uint8_t *flags_dev, const uint8_t *peer_dev, const uint32_t *slot_dev,
                          const uint64_t *ballot_dev, const uint32_t *val_dev, const uint8_t *mask_dev, uint64_t *r_ballot_dev,
                          uint32_t *r_slot_dev, void *stream);
int smr_rsp_handle_accept_replies(smr_rsp_replica *e, const uint32_t *slot_dev, const uint64_t *ballot_dev, const uint8_t *flags_dev,
                                  const uint32_t *order_dev, uint8_t *committed_dev, void *stream);
int smr_rsp_become_leader(smr_rsp_replica *e, const uint8_t *src_dev, const smr_rsp_heartbeat *hb, uint8_t *p_flags_dev,
                          uint32_t *p_trig_dev, uint64_t *p_ballot_dev, uint32_t *rc_n_dev, uint32_t *rc_slot_dev, void *stream);
int smr_rsp_handle_prepare(smr_rsp_replica *e, const uint8_t *flags_dev, const uint8_t *peer_dev, const uint32_t *trig_dev,
                           const uint64_t *ballot_dev, const smr_rsp_prepare_reply *out, void *stream);
int smr_rsp_handle_prepare_replies(smr_rsp_replica *e, const uint8_t *peer_dev, const smr_rsp_prepare_reply *in,
                                   const smr_rsp_accepts *out, void *stream);
int smr_rsp_handle_reconstruct(smr_rsp_replica *e, const uint8_t *flags_dev, const uint32_t *rc_n_dev, const uint32_t *rc_slot_dev,
                               const smr_rsp_shards *out, void *stream);
int smr_rsp_handle_reconstruct_reply(smr_rsp_replica *e, const uint8_t *flags_dev, const smr_rsp_shards *in, void *stream);
/* in->flags: a Heartbeat from peer_dev[g] arrives; reply_dev[g] = 1: mine goes back (fields in `out`) */
/* The steady state of a co-located RSPaxos cluster (one prepared leader, no timeouts) as ONE launch per tick: a block is the R
 * replicas (a wavefront each) of 64 groups; the leader's handle_req_batch on val_dev[g] (0xFFFFFFFF: none), the followers'
 * handle_msg_accept with the mask of the one shard each holds, the leader's handle_msg_accept_reply tally peers ascending and --
 * heartbeat != 0 -- the leader's Heartbeat, the followers' heard_heartbeat + Heartbeats back, the leader hearing them; messages
 * cross wavefronts through LDS.  What summerset_amd/rsp_cluster.SteadyLoop does call by call (same handler bodies, same order).
 * lost_dev (may be NULL): [4 * R] device pointers, entry k * R + q (may be NULL) = u8 [G], 1 where the message is lost:
 * k = 0 Accept leader -> q, 1 AcceptReply q -> leader, 2 Heartbeat leader -> q, 3 Heartbeat q -> leader.
 * committed_dev[g] = 1 where the tick's slot committed at the leader.  Replicas: me = index, one population / groups / window /
 * fault_tolerance; they stay the caller's.  (The shard bytes are smr_rs_from_data_encode_fanout's.) */
typedef struct smr_rsp_cluster smr_rsp_cluster;
int smr_rsp_cluster_create(smr_rsp_replica *const *reps, uint32_t n, smr_rsp_cluster **out);
void smr_rsp_cluster_destroy(smr_rsp_cluster *c);
int smr_rsp_cluster_steady_tick(smr_rsp_cluster *c, uint8_t leader, const uint32_t *val_dev, const uint8_t *const *lost_dev, int heartbeat,
                                uint8_t *committed_dev, void *stream);
int smr_rsp_handle_heartbeat(smr_rsp_replica *e, const uint8_t *peer_dev, const smr_rsp_heartbeat *in, uint8_t *reply_dev,
                             const smr_rsp_heartbeat *out, void *stream);
int smr_rsp_bcast_heartbeat(smr_rsp_replica *e, const uint8_t *flags_dev, const smr_rsp_heartbeat *out, void *stream);
/* ---- layout L2 of the RSPaxos replica engine as ONE call per tick (round 6; BASELINE config 4 as written: "1 -> 8 GPU shard over
 * xGMI"; the groups block-partitioned over `world` ranks, replica r of block b on rank (b + r) mod world: block b is led -- replica 0,
 * prepared -- from rank b) ------------------------------------------------------------------------------------------------------
 * What summerset_amd/spread_rsp.py drove from Python in rounds 3-5: the leader's from_data + RS encode with every follower's shard
 * written straight into that follower's slice of the send buffer, handle_req_batch and the Accept headers (segment 0); the
 * followers' handle_msg_accept, reply ballots straight into the backward buffer (1); the leader's tally (2); on a heartbeat tick the
 * Heartbeat out (3), heard_heartbeat + the Heartbeats back (4), the leader hearing them (5).  Exchanges (one all-to-all with static
 * split sizes each; TransportHub::send_msg, server/transport.rs:208-275) behind segments 0, 1, 3 and 4: exchange 0 Accept (header +
 * shard), 1 AcceptReply, 2 Heartbeat, 3 Heartbeat back.
 *   smr_rsp_spread_create    reps[i] = replica rep_id[i] of block rep_block[i], every one that lives on `rank`, preset to leader 0
 *                            (smr_rsp_preset_leader), window and fault_tolerance as the job's; data_len = bytes of a serialized batch
 *   smr_rsp_spread_segment   one segment, for a host that moves the buffers itself (smr_rsp_spread_buffers)
 *   smr_rsp_spread_tick      all of it: segments and smr_comm_exchange calls back to back on `stream` (smr_rsp_spread_bind_comm first)
 * data_dev / val_dev: the led block's batches (u8 [G][data_len]) and tokens (u32 [G], SMR_RSP_NULL: none), NULL on a rank that
 * leads no block; lost_dev (may be NULL): [world * 4 * R] pointers, entry (b * 4 + k) * R + q (may be NULL) = u8 [G_b], 1 where
 * block b's message is lost -- k = 0 Accept leader -> q, 1 AcceptReply q -> leader, 2 Heartbeat leader -> q, 3 Heartbeat q ->
 * leader (as smr_rsp_cluster_steady_tick's); committed_dev: u8 [G] of the led block, 1 where the tick's slot committed. */
typedef struct smr_rsp_spread smr_rsp_spread;
int smr_rsp_spread_create(smr_rsp_replica *const *reps, const uint32_t *rep_block, const uint8_t *rep_id, uint32_t n_reps,
                          const uint32_t *block_groups, uint32_t world, uint32_t rank, uint8_t population, uint32_t window, uint64_t data_len,
                          smr_rsp_spread **out);
void smr_rsp_spread_destroy(smr_rsp_spread *s);
int smr_rsp_spread_buffers(smr_rsp_spread *s, uint32_t exchange, void **send_dev, uint64_t *send_bytes, void **recv_dev, uint64_t *recv_bytes);
int smr_rsp_spread_bind_comm(smr_rsp_spread *s, smr_comm *comm);
int smr_rsp_spread_segment(smr_rsp_spread *s, uint32_t seg, const uint8_t *data_dev, const uint32_t *val_dev, const uint8_t *const *lost_dev,
                           int heartbeat, uint8_t *committed_dev, void *stream);
int smr_rsp_spread_abort_tick(smr_rsp_spread *s);
int smr_rsp_spread_tick(smr_rsp_spread *s, const uint8_t *data_dev, const uint32_t *val_dev, const uint8_t *const *lost_dev, int heartbeat,
                        uint8_t *committed_dev, void *stream);
int smr_rsp_spread_info(const smr_rsp_spread *s, uint64_t out[2]);
/* host buffers: scalars [G], peer_exec_bar [R][G], per slot [W][G] by slot % W (cells outside the ring read as
 * null instances); s_flags: bit0 leader_bk, bit1 replica_bk, bit2 external; counters[4] = commits, commands
 * executed, absorbs of a different token (none in a correct run), redirected batches */
typedef struct {
    uint8_t *leader;
    uint64_t *bal_prep_sent, *bal_prepared, *bal_max_seen;
    uint32_t *len, *commit_bar, *exec_bar, *snap_bar, *peer_exec_bar;
    uint64_t *digest;
    uint64_t *s_bal; uint8_t *s_status; uint32_t *s_val; uint8_t *s_mask; uint64_t *s_vbal; uint32_t *s_vval; uint8_t *s_vmask;
    uint8_t *s_flags; uint32_t *s_ltrig, *s_lendp; uint8_t *s_packs, *s_aacks; uint64_t *s_pmax; uint8_t *s_rsrc;
    uint32_t *s_rtrig, *s_rendp;
    uint64_t *counters;
} smr_rsp_dump_bufs;
int smr_rsp_dump(smr_rsp_replica *e, const smr_rsp_dump_bufs *host_bufs);
/* the commands the LAST handler call executed (state_machine.submit_cmd + its result, durability.rs:166-176,
 * execution.rs:10-65), group-major, in execution order within a group: (group, slot, batch token).  *n_out = how
 * many; the first `cap` are written (NULL host arrays: count only).  The next handler call starts a new list. */
int smr_rsp_exec_poll(smr_rsp_replica *e, uint32_t *group_host, uint32_t *slot_host, uint32_t *val_host, uint64_t cap, uint64_t *n_out);

/* ------------------------------------------------------------------------
 * RSPaxos payload store: the shard BYTES behind a replica's instances, resident in HBM, keyed by (slot, shard) --
 * `inst.reqs_cw` and the voted copy `inst.voted.1` (rspaxos/mod.rs:168-233) as two planes (0 = reqs, 1 = voted), each a
 * ring of W rows x n shards x G groups.  The replica engine above decides which shards exist where (token + mask per
 * instance); the store makes the bytes follow:
 *   smr_rsp_pstore_put       the leader's RSCodeword::from_data + compute_parity (rspaxos/request.rs:71-101,
 *                            rscoding.rs:165-243,447-486) of a tick's batches into the rows smr_rsp_req_batch chose: a_n /
 *                            a_slot / a_val = that call's smr_rsp_accepts (its first list entry), data_dev = uint8
 *                            [G][data_stride] serialized batches, len_dev (may be NULL) = per-group lengths <= data_len
 *   smr_rsp_pstore_follow    after ANY handler call of replica e: for every ring cell of both planes, the shards the
 *                            engine's mask has and the row lacks are taken from the sources (store, plane) that hold the
 *                            SAME token -- the sender's subset_copy + the receiver's `inst.reqs_cw = reqs_cw` /
 *                            absorb_other (rscoding.rs:255-346; messages.rs:180-194,373-380,547-560) -- entries of src may
 *                            be NULL (a list indexed by replica id has an empty seat at my own); sel_dev (may be NULL) = u8
 *                            [G]: in group g only source sel_dev[g] may give shards (the `peer` array of the handler call:
 *                            the message's sender); my own other plane is always a source; what is still missing is
 *                            rebuilt from any d shards present
 *                            (reconstruct_data on commit and at the prepare quorum, compute_parity for the re-Accepts:
 *                            durability.rs:140-160, messages.rs:227-259); a row whose token changed drops its shards.
 *                            Payload identity is the token: shard bytes are a function of (token, shard index).  Token 0
 *                            = ReqBatch::new() (messages.rs:246-252), serialized as the one byte 0x00, is synthesised.
 *   smr_rsp_pstore_get_data  RSCodeword::get_data (rscoding.rs:583-609) for a list of instances: item i = (group_dev[i], or
 *                            i when NULL; slot_dev[i], SMR_RSP_NULL = skip) -> out_dev[i][0 .. len), len_out_dev[i],
 *                            ok_dev[i] = 1 iff the row holds every data shard (and token expect_dev[i], when given)
 * Host-side reads: _dump (tok / avail / dlen [W][G] of a plane; NULL = skip), _read_row (the n x G x group_stride bytes of
 * a slot's row), _layout (device pointer + strides of a plane: shard k of group g of row r at bytes_dev + r * row_stride +
 * k * shard_stride + g * group_stride -- a row is a shard-major batch smr_rs_reconstruct / smr_rs_verify accept),
 * _counters: shards copied, shards rebuilt, shards the engine has that no source could give (0 in a correct run; counted by every
 * follow call while they stay missing -- after an absorb of a DIFFERENT token, which rscoding.rs:296-346 refuses, they always do),
 * rows whose token changed while they held shards.  n_shards = the population (<= 8), n_data_shards = the majority.
 * A vote is the codeword the instance holds at that moment (`inst.voted = (ballot, reqs_cw.clone())`, messages.rs:373-380;
 * request.rs:103-118), so a VOTED shard that equals the REQS row's shard -- same token, present there -- is NOT stored a second
 * time: it is an ALIAS of the REQS row's bytes (counted as copied all the same) until the two part ways (reqs_cw takes another
 * value in the Prepare phase, messages.rs:180-194), when follow first moves it into the VOTED row.  Every call above reads
 * through the aliases (_extract, _emit_accepts, _read_row and a follow that names plane 1 of this store as a source); a host
 * that reads plane 1 through _layout needs _voted_alias: u8 [W][G], bit k = shard k of that cell is read at the SAME offset of
 * plane 0.  _put and an _ingest into plane 0 replace a REQS row: votes that lived in it go with it (the handler in front of
 * either has re-initialised the instance's vote; the next follow re-derives the cell).
 * The calls on one store only enqueue work and must be issued on ONE stream at a time (their scratch list is the store's);
 * a source's rows must not be written by another stream meanwhile.  A message is consumed in the tick that produced it: the
 * sender's row must still hold the token the message named when the receiver's follow runs (or go through _extract / _ingest).
 * ---------------------------------------------------------------------- */
typedef struct smr_rsp_pstore smr_rsp_pstore;
int smr_rsp_pstore_create(uint32_t n_groups, uint32_t n_shards, uint32_t n_data_shards, uint32_t window, uint32_t max_data_len,
                          smr_rsp_pstore **out);
void smr_rsp_pstore_destroy(smr_rsp_pstore *s);
int smr_rsp_pstore_put(smr_rsp_pstore *s, const uint32_t *a_n_dev, const uint32_t *a_slot_dev, const uint32_t *a_val_dev,
                       const uint8_t *data_dev, uint64_t data_stride, const uint32_t *len_dev, uint32_t data_len, void *stream);
int smr_rsp_pstore_follow(smr_rsp_pstore *s, const smr_rsp_replica *e, uint32_t n_src, smr_rsp_pstore *const *src, const uint8_t *src_plane,
                          const uint8_t *sel_dev, void *stream);
/* smr_rsp_pstore_follow for n <= 8 replicas that consumed ONE sender's message (an Accept reaches every follower): stores[k]
 * follows replicas[k], each with the single source (src, src_plane) (src may be NULL: none) -- two launches for all of them
 * instead of two each.  src must not be one of the stores (nothing a store writes in this call is read by another). */
int smr_rsp_pstore_follow_many(uint32_t n, smr_rsp_pstore *const *stores, const smr_rsp_replica *const *replicas, const smr_rsp_pstore *src,
                               int src_plane, void *stream);
/* A co-located leader's tick of the byte path as ONE call and four launches (round 6): smr_rsp_pstore_put(s, ..) +
 * smr_rsp_pstore_follow(s, e, no sources) + smr_rsp_pstore_follow_many(n, stores, replicas, s, SMR_RSP_PLANE_REQS) -- the leader's
 * from_data + compute_parity (rspaxos/request.rs:71-101), its vote (request.rs:103-118) and what its n <= 8 co-located followers'
 * Accept handlers took of it (messages.rs:373-380) -- with the same cells, bytes and counters as the three calls (five launches):
 * the leader's byte launch rides in the followers' plan launch, and the put launch writes the shards the followers' engines took of
 * the new codewords into THEIR rows as well (`subset_copy` at the sender + `inst.reqs_cw = reqs_cw` at the receiver, request.rs:127-142,
 * messages.rs:373-380, without reading the leader's row back; the followers' plan counts them as copied and lists nothing).  n = 0:
 * the first two calls alone.  Call it AFTER the engines' handlers of the tick, as the three calls are. */
int smr_rsp_pstore_put_follow_all(smr_rsp_pstore *s, const smr_rsp_replica *e, const uint32_t *a_n_dev, const uint32_t *a_slot_dev,
                                  const uint32_t *a_val_dev, const uint8_t *data_dev, uint64_t data_stride, const uint32_t *len_dev,
                                  uint32_t data_len, uint32_t n, smr_rsp_pstore *const *stores, const smr_rsp_replica *const *replicas,
                                  void *stream);
int smr_rsp_pstore_get_data(smr_rsp_pstore *s, uint32_t n_items, const uint32_t *group_dev, const uint32_t *slot_dev, const uint32_t *expect_dev,
                            uint8_t *out_dev, uint64_t out_stride, uint32_t *len_out_dev, uint8_t *ok_dev, void *stream);
/* The payload of a message between replicas on DIFFERENT devices / hosts.  A message buffer is laid out like one row: shard k of
 * group g at k * shard_stride + g * group_stride (smr_rsp_pstore_layout), n_shards x G x group_stride bytes.
 *   _extract  sender: RSCodeword::subset_copy (rscoding.rs:255-293) -- per group g with flags_dev[g] != 0 (NULL: all) the shards
 *             mask_dev[g] of row slot_dev[g] of `plane` that the row holds go to out_dev; tok_out / mask_out / dlen_out [G] = the
 *             codeword's header (token SMR_RSP_NULL, mask 0: nothing)
 *   _ingest   receiver: row slot_dev[g] of `plane` of a STAGING store (same geometry as the replica's store) := that header and
 *             those bytes, replacing what it held; smr_rsp_pstore_follow then names (staging, plane) as the source behind the
 *             handler call that consumes the message.  Only enqueue work on `stream`. */
int smr_rsp_pstore_extract(const smr_rsp_pstore *s, int plane, const uint8_t *flags_dev, const uint32_t *slot_dev, const uint8_t *mask_dev,
                           uint8_t *out_dev, uint32_t *tok_out_dev, uint8_t *mask_out_dev, uint32_t *dlen_out_dev, void *stream);
int smr_rsp_pstore_ingest(smr_rsp_pstore *s, int plane, const uint8_t *flags_dev, const uint32_t *slot_dev, const uint32_t *tok_dev,
                          const uint8_t *mask_dev, const uint32_t *dlen_dev, const uint8_t *in_dev, void *stream);
/* The Accepts of a call as the frames TcpTransport writes (safetcp.rs:127-132; PeerMsg::Accept { slot, ballot, reqs_cw },
 * rspaxos/mod.rs:262-270; RSCodeword's Encode, utils/rscoding.rs:43-77), payload included, straight out of the store: frame g =
 * frames_dev[g * stride .. + len_dev[g]) carries the shards mask_dev[g] of row slot_dev[g] of `plane` that the row holds -- byte
 * for byte smr_wire_rsp_accept(smr_wire_rscodeword(..)).  len_dev[g] = 0: nothing to send (flags_dev[g] = 0, a NULL slot, no
 * shard of the mask present); 0xFFFFFFFF: the frame does not fit `stride` (header <= 64 bytes + 11 per shard + the shards). */
int smr_rsp_pstore_emit_accepts(const smr_rsp_pstore *s, int plane, const uint8_t *flags_dev, const uint32_t *slot_dev, const uint64_t *ballot_dev,
                                const uint8_t *mask_dev, uint8_t *frames_dev, uint64_t stride, uint32_t *len_dev, void *stream);
int smr_rsp_pstore_dump(smr_rsp_pstore *s, int plane, uint32_t *tok_host, uint8_t *avail_host, uint32_t *dlen_host);
int smr_rsp_pstore_read_row(smr_rsp_pstore *s, int plane, uint32_t slot, uint8_t *bytes_host);
int smr_rsp_pstore_layout(const smr_rsp_pstore *s, int plane, void **bytes_dev, uint64_t *row_stride, uint64_t *shard_stride,
                          uint64_t *group_stride);
int smr_rsp_pstore_voted_alias(const smr_rsp_pstore *s, const uint8_t **alias_dev, uint8_t *alias_host);   /* either may be NULL */
int smr_rsp_pstore_counters(smr_rsp_pstore *s, uint64_t *out4_host);
/* of `copied`: the shards a sender's put launch wrote straight into this store (smr_*_pstore_put_follow_all, below: a follower's
 * steady-tick copy out of the leader's row is done by the launch that has the leader's bytes in registers); 0 with SMR_PS_DELIVER=0 */
int smr_rsp_pstore_debug_delivered(smr_rsp_pstore *s, uint64_t *out_host);

/* CRaft: the same store keyed by LOG INDEX -- the shard bytes of `LogEntry::reqs_cw` (/root/reference/src/protocols/craft/mod.rs:129-150)
 * behind a CRaft replica (smr_raft_leader with smr_raft_craft_enable: leader or follower).  The engine keeps an entry's codeword as
 * the avail_shards_map of ring cell [slot % window]; the store makes the bytes follow it:
 *   smr_craft_pstore_create  one plane (a log entry has one codeword; the voted plane of the RSPaxos store is not allocated and
 *                            plane 1 is refused by every call); n_data_shards = the majority, n_shards = the population
 *   smr_craft_pstore_put     the leader's RSCodeword::from_data + compute_parity of the batch it appended at slot_dev[g]
 *                            (SMR_RSP_NULL: none) -- craft/request.rs:71-76: the leader holds every shard.  Call it behind
 *                            smr_raft_leader_append / _append_emit with one entry per group
 *   smr_craft_pstore_follow  after ANY handler call of replica e: every held log entry's shards the engine's bitmap has and the row
 *                            lacks are taken from the sources' rows that hold the SAME (slot, term) -- an AppendEntries' entries
 *                            (craft/durability.rs:41-80: one's own shard, or the data shards in full-copy mode,
 *                            leadership.rs:80-141; absorbed at craft/messages.rs:133-146), a ReconstructReply's slots_data
 *                            (messages.rs:665-745); sel_dev (may be NULL) = the message's sender per group, an index into src --
 *                            and what the bitmap gained by reconstruct_data on commit (messages.rs:193-233, 699-737) is rebuilt
 *                            from any majority of shards present.  An entry the log no longer holds (truncated, overwritten by
 *                            another term's, trimmed) drops its shards.
 * An entry's identity is (slot, term) (what the consistency check compares); everything else -- smr_rsp_pstore_get_data for the
 * executed entries, _extract / _ingest for replicas on different devices, _dump, _read_row, _layout, _counters, _destroy -- is the
 * RSPaxos store's, on plane 0. */
int smr_craft_pstore_create(uint32_t n_groups, uint32_t n_shards, uint32_t n_data_shards, uint32_t window, uint32_t max_data_len,
                            smr_rsp_pstore **out);
int smr_craft_pstore_put(smr_rsp_pstore *s, const smr_raft_leader *e, const uint32_t *slot_dev, const uint8_t *data_dev, uint64_t data_stride,
                         const uint32_t *len_dev, uint32_t data_len, void *stream);
int smr_craft_pstore_follow(smr_rsp_pstore *s, const smr_raft_leader *e, uint32_t n_src, smr_rsp_pstore *const *src, const uint8_t *sel_dev,
                            void *stream);
/* smr_craft_pstore_follow for n <= 8 followers that consumed ONE leader's AppendEntries: stores[k] follows replicas[k], each with
 * the single source src (may be NULL; none of the stores) -- three launches for all of them instead of three each */
int smr_craft_pstore_follow_many(uint32_t n, smr_rsp_pstore *const *stores, const smr_raft_leader *const *replicas, const smr_rsp_pstore *src,
                                 void *stream);
/* smr_craft_pstore_put(s, e, ..) + smr_craft_pstore_follow(s, e, no sources) + smr_craft_pstore_follow_many(n, stores, replicas, s)
 * in four launches, as smr_rsp_pstore_put_follow_all: call it after the leader's append AND the followers' AppendEntries handlers
 * (craft/request.rs:71-100, craft/messages.rs:133-146) -- the entries' terms and masks are the engines' */
int smr_craft_pstore_put_follow_all(smr_rsp_pstore *s, const smr_raft_leader *e, const uint32_t *slot_dev, const uint8_t *data_dev,
                                    uint64_t data_stride, const uint32_t *len_dev, uint32_t data_len, uint32_t n,
                                    smr_rsp_pstore *const *stores, const smr_raft_leader *const *replicas, void *stream);

/* ------------------------------------------------------------------------
 * RepNothing (BASELINE config 1) + the KV state machine: host-only plumbing
 * ---------------------------------------------------------------------- */
typedef struct smr_repnothing smr_repnothing;
#define SMR_CMD_GET 0   /* Command::Get { key }        (src/server/statemach.rs:21-27) */
#define SMR_CMD_PUT 1   /* Command::Put { key, value } */

int smr_repnothing_create(smr_repnothing **out);
void smr_repnothing_destroy(smr_repnothing *h);

/* handle_req_batch (src/protocols/rep_nothing/request.rs:11-37) for one batch
 * of n >= 1 requests, followed -- LS-1 rule 0 -- by its WAL completion
 * (durability.rs:10-51) and the execution of its commands in order
 * (statemach.rs:193-202, execution.rs:10-67); one reply per request is queued.
 * value / value_len are read for Put entries only. */
int smr_repnothing_submit_batch(smr_repnothing *h, uint32_t n, const uint64_t *client, const uint64_t *req_id,
                                const uint8_t *kind, const char *const *key, const uint32_t *key_len,
                                const char *const *value, const uint32_t *value_len, uint64_t *inst_idx);

/* Next queued ApiReply::normal (execution.rs:46-52): returns 1 and fills the
 * outputs, 0 when the queue is empty, < 0 on error.  has_value: Get -> value
 * found; Put -> old_value existed. */
int smr_repnothing_poll_reply(smr_repnothing *h, uint64_t *client, uint64_t *req_id, uint8_t *kind, int *has_value,
                              char *value_buf, uint32_t value_cap, uint32_t *value_len);

/* instances logged, WAL offset (framed bincode sizes), commands executed, keys in the state */
int smr_repnothing_stats(smr_repnothing *h, uint64_t *n_insts, uint64_t *wal_offset, uint64_t *n_execed,
                         uint64_t *n_keys);

/* ------------------------------------------------------------------------
 * Wire + WAL formats of the MultiPaxos hot-path messages (host only)
 * frame = 8-byte big-endian length + bincode-standard payload (src/utils/safetcp.rs:46,127-132;
 * src/server/storage.rs:326-346); payload = PeerMessage::Msg { msg: PeerMsg } (src/server/
 * transport.rs:37-52, src/protocols/multipaxos/mod.rs:298-368) resp. WalEntry (mod.rs:261-274).
 * Encoders return the frame's byte count (< 0: error, e.g. buffer too small).
 * ---------------------------------------------------------------------- */
#define SMR_WIRE_PREPARE 0
#define SMR_WIRE_PREPARE_REPLY 1
#define SMR_WIRE_ACCEPT 2
#define SMR_WIRE_ACCEPT_REPLY 3
#define SMR_WIRE_READ_QUERY 4
#define SMR_WIRE_READ_QUERY_REPLY 5
#define SMR_WIRE_HEARTBEAT 6
#define SMR_WIRE_COMMIT_NOTICE 7
#define SMR_WIRE_LEAVE 0xFE      /* PeerMessage::Leave */
#define SMR_WIRE_OTHER 0xFF      /* a frame of another kind (lease traffic): skipped */

/* bincode(ReqBatch) of n Get / Put requests -- the bytes an Accept carries and RSCodeword shards;
 * no frame header.  Arguments as smr_repnothing_submit_batch. */
int64_t smr_wire_reqbatch(uint32_t n, const uint64_t *client, const uint64_t *req_id, const uint8_t *kind,
                          const char *const *key, const uint32_t *key_len, const char *const *value,
                          const uint32_t *value_len, uint8_t *out, uint64_t cap);
int64_t smr_wire_prepare(uint64_t trigger_slot, uint64_t ballot, uint8_t *out, uint64_t cap);
int64_t smr_wire_prepare_reply(uint64_t slot, uint64_t trigger_slot, uint64_t endprep_slot, uint64_t ballot,
                               int has_voted, uint64_t voted_ballot, const uint8_t *voted_reqs, uint64_t voted_reqs_len,
                               uint64_t accept_bar, uint8_t *out, uint64_t cap);
/* reqs = bincode(ReqBatch) bytes (smr_wire_reqbatch) */
int64_t smr_wire_accept(uint64_t slot, uint64_t ballot, const uint8_t *reqs, uint64_t reqs_len, uint8_t *out,
                        uint64_t cap);
int64_t smr_wire_accept_reply(uint64_t slot, uint64_t ballot, uint8_t *out, uint64_t cap);
/* PeerMsg::ReadQuery { reads }: reads = bincode(ReqBatch) of Gets; ReadQueryReply { rq_id, replies, from_leader } with
 * replies[i] = state[i] 0 None / 1 Some((slot, None)) / 2 Some((slot, Some(value))) (multipaxos/mod.rs:344-362) */
int64_t smr_wire_read_query(const uint8_t *reads, uint64_t reads_len, uint8_t *out, uint64_t cap);
int64_t smr_wire_read_query_reply(uint64_t rq_client, uint64_t rq_req_id, uint32_t n, const uint8_t *state, const uint64_t *slot,
                                  const char *const *value, const uint32_t *value_len, int from_leader, uint8_t *out, uint64_t cap);
/* PeerMsg::Heartbeat / CommitNotice (multipaxos/mod.rs:364-378) */
int64_t smr_wire_heartbeat(uint64_t ballot, uint64_t commit_bar, uint64_t exec_bar, uint64_t snap_bar, uint8_t *out, uint64_t cap);
int64_t smr_wire_commit_notice(uint64_t ballot, uint64_t commit_bar, uint8_t *out, uint64_t cap);
/* WalEntry::{PrepareBal, AcceptData, CommitSlot} log records */
int64_t smr_wal_prepare_bal(uint64_t slot, uint64_t ballot, uint8_t *out, uint64_t cap);
int64_t smr_wal_accept_data(uint64_t slot, uint64_t ballot, const uint8_t *reqs, uint64_t reqs_len, uint8_t *out,
                            uint64_t cap);
int64_t smr_wal_commit_slot(uint64_t slot, uint8_t *out, uint64_t cap);

typedef struct {
    uint8_t kind;                  /* SMR_WIRE_* */
    uint8_t has_voted;             /* PrepareReply: voted is Some */
    uint64_t slot, ballot, trigger_slot, endprep_slot, accept_bar, voted_ballot;
    uint64_t reqs_off, reqs_len;   /* Accept / voted / ReadQuery: where in the buffer the bincode(ReqBatch) bytes lie */
    uint64_t commit_bar, exec_bar, snap_bar;          /* Heartbeat, CommitNotice */
    uint64_t rq_client, rq_req_id, n_replies;         /* ReadQueryReply */
    uint64_t replies_off, replies_len;                /* its Vec of replies, for smr_wire_read_query_replies */
    uint8_t from_leader;
} smr_wire_msg;
/* Parses the first TCP frame of buf[0, len): returns the bytes it occupies, 0 if it is not complete
 * yet (safetcp.rs:30-70 reads until it is), < 0 if malformed. */
int64_t smr_wire_decode(const uint8_t *buf, uint64_t len, smr_wire_msg *out);
/* unpacks the replies of a decoded ReadQueryReply (p = buf + replies_off, len = replies_len): state / slot per reply and
 * where its value bytes lie relative to p; returns their number (< 0: malformed, or more than max) */
int64_t smr_wire_read_query_replies(const uint8_t *p, uint64_t len, uint32_t max, uint8_t *state, uint64_t *slot, uint64_t *value_off,
                                    uint64_t *value_len);

/* ---- Raft frames (src/protocols/raft/mod.rs:117-234): PeerMsg::{AppendEntries 0, AppendEntriesReply 1,
 * RequestVote 2, RequestVoteReply 3}; DurEntry::Metadata log record */
int64_t smr_wire_raft_append_entries(uint64_t term, uint64_t prev_slot, uint64_t prev_term, uint32_t n,
                                     const uint64_t *entry_term, const uint8_t *reqs, const uint64_t *reqs_off,
                                     const uint8_t *external, uint64_t leader_commit, uint64_t last_snap, uint8_t *out,
                                     uint64_t cap);
int64_t smr_wire_raft_append_entries_reply(uint64_t term, uint64_t end_slot, int has_conflict, uint64_t conflict_term,
                                           uint64_t conflict_slot, uint8_t *out, uint64_t cap);
int64_t smr_wire_raft_request_vote(uint64_t term, uint64_t last_slot, uint64_t last_term, uint8_t *out, uint64_t cap);
int64_t smr_wire_raft_request_vote_reply(uint64_t term, int granted, uint8_t *out, uint64_t cap);
int64_t smr_wal_raft_metadata(uint64_t curr_term, uint8_t voted_for, uint8_t *out, uint64_t cap);
typedef struct {
    uint8_t kind;                  /* 0..3 as above, SMR_WIRE_LEAVE, SMR_WIRE_OTHER */
    uint8_t has_conflict, granted;
    uint32_t n_entries;
    uint64_t term, prev_slot, prev_term, leader_commit, last_snap, end_slot, conflict_term, conflict_slot, last_slot,
        last_term;
} smr_wire_raft_msg;
/* as smr_wire_decode; the terms of an AppendEntries' entries go to entry_term_out[0 .. max_entries) */
int64_t smr_wire_raft_decode(const uint8_t *buf, uint64_t len, smr_wire_raft_msg *out, uint64_t *entry_term_out,
                             uint32_t max_entries);

/* ---- RSPaxos wire + WAL (src/protocols/rspaxos/mod.rs:207-311) and RSCodeword's own Encode / Decode
 * (src/utils/rscoding.rs:43-109): num_data_shards u8, num_parity_shards u8, data_len, shard_len,
 * Vec<Option<Vec<u8>>> shards, Option<T> data_copy (None on the wire).  PeerMsg variants: Prepare 0, PrepareReply 1,
 * Accept 2, AcceptReply 3 (SMR_WIRE_* above), Reconstruct 4, ReconstructReply 5, Heartbeat 6. */
#define SMR_WIRE_RSP_RECONSTRUCT 4
#define SMR_WIRE_RSP_RECONSTRUCT_REPLY 5
#define SMR_WIRE_RSP_HEARTBEAT 6
/* bincode(RSCodeword) of one codeword (no frame): shard k is read from shards + k * shard_stride if bit k of avail_mask */
int64_t smr_wire_rscodeword(uint8_t d, uint8_t p, uint64_t data_len, uint64_t shard_len, uint32_t avail_mask, const uint8_t *shards,
                            uint64_t shard_stride, uint8_t *out, uint64_t cap);
int64_t smr_wire_rsp_prepare(uint64_t trigger_slot, uint64_t ballot, uint8_t *out, uint64_t cap);
/* cw = bytes from smr_wire_rscodeword */
int64_t smr_wire_rsp_prepare_reply(uint64_t slot, uint64_t trigger_slot, uint64_t endprep_slot, uint64_t ballot, int has_voted,
                                   uint64_t voted_ballot, const uint8_t *cw, uint64_t cw_len, uint8_t *out, uint64_t cap);
int64_t smr_wire_rsp_accept(uint64_t slot, uint64_t ballot, const uint8_t *cw, uint64_t cw_len, uint8_t *out, uint64_t cap);
int64_t smr_wire_rsp_accept_reply(uint64_t slot, uint64_t ballot, uint8_t *out, uint64_t cap);
int64_t smr_wire_rsp_reconstruct(uint32_t n, const uint64_t *slots, uint8_t *out, uint64_t cap);
/* n entries of the slots_data map in the order given; codeword i = cws[cw_off[i] .. cw_off[i + 1]) */
int64_t smr_wire_rsp_reconstruct_reply(uint32_t n, const uint64_t *slots, const uint64_t *ballots, const uint8_t *cws,
                                       const uint64_t *cw_off, uint8_t *out, uint64_t cap);
int64_t smr_wire_rsp_heartbeat(uint64_t ballot, uint64_t commit_bar, uint64_t exec_bar, uint64_t snap_bar, uint8_t *out, uint64_t cap);
/* WalEntry::AcceptData { slot, ballot, reqs_cw }; PrepareBal / CommitSlot are smr_wal_prepare_bal / smr_wal_commit_slot */
int64_t smr_wal_rsp_accept_data(uint64_t slot, uint64_t ballot, const uint8_t *cw, uint64_t cw_len, uint8_t *out, uint64_t cap);

typedef struct {
    uint8_t num_data_shards, num_parity_shards;
    uint32_t avail_mask;            /* shards present */
    uint64_t data_len, shard_len;
    uint64_t shard_off[16];         /* where in the buffer shard k's bytes lie (present shards only) */
} smr_wire_codeword;
typedef struct {
    uint8_t kind;                   /* SMR_WIRE_* / SMR_WIRE_RSP_* */
    uint8_t has_voted;
    uint32_t n_items;               /* codewords (Accept / voted: 1), Reconstruct slots, ReconstructReply entries */
    uint64_t slot, ballot, trigger_slot, endprep_slot, voted_ballot, commit_bar, exec_bar, snap_bar;
} smr_wire_rsp_msg;
/* as smr_wire_decode; codewords go to cws[0 .. max_items), Reconstruct(Reply) slots / ballots to slots / ballots */
int64_t smr_wire_rsp_decode(const uint8_t *buf, uint64_t len, smr_wire_rsp_msg *out, smr_wire_codeword *cws, uint64_t *slots,
                            uint64_t *ballots, uint32_t max_items);

/* ---- EPaxos wire + WAL (src/protocols/epaxos/mod.rs:124,199,254-377): SlotIdx(row u8, col), DepSet = Vec<Option<usize>>
 * (deps[i] == SMR_EP_NONE: None).  kind = PeerMsg variant: */
#define SMR_WIRE_EP_PRE_ACCEPT 0
#define SMR_WIRE_EP_PRE_ACCEPT_REPLY 1
#define SMR_WIRE_EP_ACCEPT 2
#define SMR_WIRE_EP_ACCEPT_REPLY 3
#define SMR_WIRE_EP_COMMIT_NOTICE 4
/* seq / deps are ignored for AcceptReply; reqs = bincode(ReqBatch), read for PreAccept / Accept / CommitNotice only */
int64_t smr_wire_ep_msg(uint8_t kind, uint8_t row, uint64_t col, uint64_t ballot, uint64_t seq, const uint32_t *deps, uint32_t n_deps,
                        const uint8_t *reqs, uint64_t reqs_len, uint8_t *out, uint64_t cap);
/* WalEntry::{PreAcceptSlot 0, AcceptSlot 1, CommitSlot 2} { slot, ballot, seq, deps, reqs } */
int64_t smr_wal_ep_slot(uint8_t kind, uint8_t row, uint64_t col, uint64_t ballot, uint64_t seq, const uint32_t *deps, uint32_t n_deps,
                        const uint8_t *reqs, uint64_t reqs_len, uint8_t *out, uint64_t cap);
typedef struct {
    uint8_t kind, row;              /* SMR_WIRE_EP_* (or SMR_WIRE_LEAVE / SMR_WIRE_OTHER); SlotIdx.0 */
    uint32_t n_deps;
    uint64_t col, ballot, seq;
    uint64_t reqs_off, reqs_len;    /* where in the buffer the bincode(ReqBatch) bytes lie */
} smr_wire_ep_msg_t;
int64_t smr_wire_ep_decode(const uint8_t *buf, uint64_t len, smr_wire_ep_msg_t *out, uint32_t *deps_out, uint32_t max_deps);

/* ---- MultiPaxos peer traffic, parsed on the device (csrc/wire_ingest.hip) ----------------
 * The receive side of `TcpTransport` (src/server/transport.rs:404-470 -> safetcp.rs:30-70) for a batch of connections:
 * buf_dev[conn_off[c] .. conn_off[c + 1]) are the bytes connection c -- replica `conn_peer[c]` of group
 * `conn_group[c]` -- delivered since the last call, frames `[u64 BE length][bincode(PeerMessage)]` back to back, the
 * last one possibly incomplete.  Every complete frame, in stream order, becomes
 *   PeerMsg::AcceptReply { slot, ballot, .. }   an smr_mp_ack { group, slot, ballot, peer }: what smr_mp_deliver_acks takes
 *                                               (a slot above 2^32 - 1 goes to `others`: the engine's slots are u32);
 *   PeerMsg::Heartbeat / CommitNotice           an smr_wire_hb;
 *   anything else (Prepare, PrepareReply, Accept, ReadQuery, ReadQueryReply, PeerMessage::Leave, lease traffic)
 *                                               an smr_wire_other { conn, kind, off, len }: off is the frame's offset
 *                                               in buf_dev, for the host's smr_wire_decode -- located, not validated.
 * Records are written in the order the sequential decoder would produce them (connection by connection, frame by
 * frame); counts_dev[0 .. 3) = their numbers (records past a capacity are counted, not stored), counts_dev[3] = the
 * connections that hit a malformed frame (smr_wire_decode's rules: a length above 10^12, a variant that does not
 * parse, a frame that does not end where its length says) -- status_dev[c] = 1 there and the connection stops at
 * that frame.  consumed_dev[c] = the bytes of c's complete frames: the host keeps the rest for the next call, as
 * safetcp's read buffer does.  buf_dev must be 16-byte aligned; scratch_dev holds smr_wire_ingest_scratch_bytes(n_conn)
 * bytes.  Only enqueues work on `stream`. */
typedef struct {
    uint32_t group, peer;
    uint32_t kind;                 /* SMR_WIRE_HEARTBEAT or SMR_WIRE_COMMIT_NOTICE */
    uint32_t reserved;             /* 0 */
    uint64_t ballot, commit_bar, exec_bar, snap_bar;   /* CommitNotice: exec_bar = snap_bar = 0 */
} smr_wire_hb;                     /* 48 bytes */
typedef struct {
    uint32_t conn, kind;           /* SMR_WIRE_* of the frame (SMR_WIRE_OTHER: a variant this build does not know) */
    uint64_t off, len;             /* the whole frame, header included */
} smr_wire_other;                  /* 24 bytes */
uint64_t smr_wire_ingest_scratch_bytes(uint32_t n_conn);
int smr_wire_ingest_mp(const uint8_t *buf_dev, uint64_t buf_len, const uint64_t *conn_off_dev, const uint32_t *conn_group_dev,
                       const uint8_t *conn_peer_dev, uint32_t n_conn, smr_mp_ack *acks_dev, uint64_t ack_cap, smr_wire_hb *hbs_dev,
                       uint64_t hb_cap, smr_wire_other *others_dev, uint64_t other_cap, uint64_t *counts_dev, uint64_t *consumed_dev,
                       int32_t *status_dev, void *scratch_dev, void *stream);
/* The same parse in ONE pass, for a host that does not need dense lists (round 5).  smr_wire_ingest_mp's lists are dense and in
 * the sequential decoder's order ACROSS connections, which costs a counting parse of the whole buffer in front of the writing one.
 * The reference promises less: a connection's messages in order, connections in whatever order the event loop meets them
 * (transport.rs:404-470).  Here every connection has segments of its own and nothing is counted first:
 *   acks_dev    connection c's AcceptReplies as smr_wire_ack12 { slot, ballot } -- the group and the peer are the connection's --
 *               from record conn_off[c] / 13 on (an AcceptReply frame has >= 13 bytes, so the segments cannot meet; ack_cap >=
 *               buf_len / 13 + 1 records is required); smr_mp_deliver_acks_conn takes them as they are;
 *   hbs_dev     [n_conn][hb_per_conn], others_dev [n_conn][other_per_conn]: c's first Heartbeats / CommitNotices and located frames;
 *               one more than that stops the connection IN FRONT of the frame (status_dev[c] = 2, not malformed: consumed_dev[c]
 *               says where the next call goes on -- what an incomplete frame does too);
 *   cnt_dev     u32 [n_conn][3]: how many of each connection c has (sum them where a total is wanted).
 * Frame rules, consumed_dev / status_dev (1 = malformed) and the records themselves are smr_wire_ingest_mp's; no scratch. */
typedef struct {
    uint32_t slot, ballot_lo, ballot_hi;     /* PeerMsg::AcceptReply { slot, ballot } of the connection the segment belongs to */
} smr_wire_ack12;                            /* 12 bytes */
int smr_wire_ingest_mp_conn(const uint8_t *buf_dev, uint64_t buf_len, const uint64_t *conn_off_dev, const uint32_t *conn_group_dev,
                            const uint8_t *conn_peer_dev, uint32_t n_conn, smr_wire_ack12 *acks_dev, uint64_t ack_cap, smr_wire_hb *hbs_dev,
                            uint32_t hb_per_conn, smr_wire_other *others_dev, uint32_t other_per_conn, uint32_t *cnt_dev, uint64_t *consumed_dev,
                            int32_t *status_dev, void *stream);

/* ---- Raft / EPaxos reply traffic parsed on the device (round 3; csrc/wire_ingest_replies.hip).  The Raft leader and the
 * EPaxos command leader take ONE reply per (peer, group) and call, as arrays [R][G]: these calls fill those arrays from the
 * bytes the leader's connections delivered -- connection c = the stream from peer conn_peer[c] (< population) of group
 * conn_group[c] (< n_groups), frames `[u64 BE length][bincode(PeerMessage)]` back to back in
 * buf_dev[conn_off[c] .. conn_off[c + 1]), the last one possibly incomplete.  Per connection, in stream order: the FIRST
 *   Raft    PeerMsg::AppendEntriesReply { term, end_slot, conflict }   (raft/mod.rs:203-234; what the follower's
 *           handle_msg_append_entries sends, raft/messages.rs:128-240)
 *   EPaxos  PeerMsg::PreAcceptReply { slot, ballot, seq, deps } for MY instance slot == (me, col_dev[group]) with
 *           `population` dependencies (epaxos/mod.rs:306-377; messages.rs:81-88)
 * is written at [peer][group] of the arrays smr_raft_leader_handle_replies / smr_raft_tick resp.
 * smr_ep_handle_pre_accept_replies take (flags bit0 = present, Raft bit1 = `conflict` is Some; deps [R][R][G] = (peer, dep
 * row, group)); the walk STOPS in front of a second one (consumed_dev[c] = where the next call's stream starts;
 * counts_dev[3] counts such connections).  Every other complete frame -- and a reply the arrays cannot hold: a slot
 * beyond u32, a PreAcceptReply for another instance -- is located for the host in others_dev (in no particular order;
 * counts_dev[1] = their number, those past other_cap are counted, not stored).  A frame that breaks the host decoder's
 * rules (smr_wire_raft_decode / smr_wire_ep_decode: a length above 10^12, an unknown Raft variant, a reply that does not
 * end where its length says) makes its connection malformed: status_dev[c] = 1, consumed_dev[c] = 0, nothing of it
 * counts (a reply it delivered before stays in the arrays, with its flag set, and so do the frames of that connection
 * already located in others_dev: a host that re-decodes the connection from byte 0 because consumed_dev[c] = 0 must first
 * drop the others_dev entries whose conn has status 1, or it handles those frames twice); counts_dev[2] = such
 * connections.  counts_dev[0] = the replies taken.  flags_dev is zeroed by the call; the other arrays are only written where flags says so.  Two connections
 * with the same (group, peer): the caller's error (one of them wins).  buf_dev must be 16-byte aligned.  conn_len_dev (may be
 * NULL): the length of every connection's bytes where they do not lie back to back -- connection c is then
 * buf_dev[conn_off[c] .. conn_off[c] + conn_len[c]) and conn_off needs n_conn entries, ascending; this is the layout the emit
 * calls below leave (slot i at i * stride, len_dev[i] bytes), so replies can go from a follower's arrays through frames into
 * the leader's arrays without leaving the device.  Only enqueues work on `stream`. */
int smr_wire_ingest_raft_replies(const uint8_t *buf_dev, uint64_t buf_len, const uint64_t *conn_off_dev, const uint32_t *conn_group_dev,
                                 const uint8_t *conn_peer_dev, const uint8_t *conn_len_dev, uint32_t n_conn, uint32_t n_groups, uint8_t population,
                                 uint64_t *reply_term_dev, uint32_t *end_slot_dev, uint64_t *conflict_term_dev, uint32_t *conflict_slot_dev,
                                 uint8_t *flags_dev, smr_wire_other *others_dev, uint64_t other_cap, uint64_t *counts_dev,
                                 uint64_t *consumed_dev, int32_t *status_dev, void *stream);
int smr_wire_ingest_ep_pre_accept_replies(const uint8_t *buf_dev, uint64_t buf_len, const uint64_t *conn_off_dev, const uint32_t *conn_group_dev,
                                          const uint8_t *conn_peer_dev, const uint8_t *conn_len_dev, uint32_t n_conn, uint32_t n_groups, uint8_t population, uint8_t me,
                                          const uint32_t *col_dev, uint64_t *ballot_dev, uint64_t *seq_dev, uint32_t *deps_dev,
                                          uint8_t *flags_dev, smr_wire_other *others_dev, uint64_t other_cap, uint64_t *counts_dev,
                                          uint64_t *consumed_dev, int32_t *status_dev, void *stream);

int smr_wire_ingest_rsp_accept_replies(const uint8_t *buf_dev, uint64_t buf_len, const uint64_t *conn_off_dev, const uint32_t *conn_group_dev,
                                       const uint8_t *conn_peer_dev, const uint8_t *conn_len_dev, uint32_t n_conn, uint32_t n_groups, uint8_t population, uint32_t *slot_dev,
                                       uint64_t *ballot_dev, uint8_t *flags_dev, smr_wire_other *others_dev, uint64_t other_cap,
                                       uint64_t *counts_dev, uint64_t *consumed_dev, int32_t *status_dev, void *stream);
/* (RSPaxos PeerMsg::AcceptReply { slot, ballot }, rspaxos/mod.rs:262-305 -> the [R][G] arrays smr_rsp_handle_accept_replies takes; same rules) */
/* Round 6: the Raft leader's receive side of a tick in ONE launch -- smr_wire_ingest_raft_replies as the prologue of
 * smr_raft_leader_handle_replies (safetcp.rs:46,127-132 framing, raft/mod.rs:203-234 AppendEntriesReply, raft/messages.rs:222-309
 * the handler): no [R][G] reply arrays in between, nothing to clear in front (the two calls: two memsets, two launches, 13 bytes
 * per connection written and read back).  The connections come DENSE: n_conn == n_groups * (population - 1), connection
 * g * (population - 1) + k is group g's k-th follower, peer ids ascending with the leader's own left out; their bytes as in
 * smr_wire_ingest_raft_replies (conn_off_dev [n_conn + 1], or starts + conn_len_dev).  Frames taken, frames located
 * (others_dev, counts), consumed_dev / status_dev: exactly that call's; what the leader then does with the replies (order_dev:
 * delivery order per group, NULL = peer order): exactly smr_raft_leader_handle_replies'.  counts_dev[4] is WRITTEN by the
 * call's last block (no need to clear it). */
int smr_raft_leader_handle_wire_replies(smr_raft_leader *l, const uint8_t *buf_dev, uint64_t buf_len, const uint64_t *conn_off_dev,
                                        const uint8_t *conn_len_dev, uint32_t n_conn, const uint32_t *order_dev, smr_wire_other *others_dev,
                                        uint64_t other_cap, uint64_t *counts_dev, uint64_t *consumed_dev, int32_t *status_dev, void *stream);

/* ---- reply frames written on the device (round 3; csrc/wire_emit.hip): the send half.  A follower's handler leaves its
 * replies as device arrays; these calls write, for every reply, the frame TcpTransport would send -- `[u64 BE length]
 * [bincode(PeerMessage::Msg { msg })]`, the bytes of smr_wire_accept_reply / smr_wire_raft_append_entries_reply /
 * smr_wire_ep_msg -- into slot i of `frames_dev` (fixed stride, 8-byte aligned) and its length into len_dev[i] (0: no reply):
 *   smr_wire_emit_mp_accept_replies      record i of smr_mp_collect_acks -> PeerMsg::AcceptReply { slot, ballot, reply_ts: None }
 *   smr_wire_emit_raft_replies           group g of smr_raft_replica_handle_append_entries' reply arrays (flags bit0 = sent, bit1 =
 *                                        conflict) -> PeerMsg::AppendEntriesReply { term, end_slot, conflict }
 *   smr_wire_emit_ep_pre_accept_replies  group g of smr_ep_handle_pre_accept's reply (flags bit0 = sent; deps [R][G]) for the instance
 *                                        (row, col[g]) -> PeerMsg::PreAcceptReply { slot, ballot, seq, deps }
 * The socket layer (or an exchange's pack pass) sends len[i] bytes from frames + i * stride.  Only enqueue work on `stream`. */
#define SMR_WIRE_EMIT_MP_STRIDE 32
#define SMR_WIRE_EMIT_RAFT_STRIDE 48
#define SMR_WIRE_EMIT_EP_STRIDE 96
int smr_wire_emit_mp_accept_replies(const smr_mp_ack *acks_dev, uint64_t n, uint8_t *frames_dev, uint8_t *len_dev, void *stream);
int smr_wire_emit_raft_replies(const uint8_t *flags_dev, const uint64_t *term_dev, const uint32_t *end_slot_dev, const uint64_t *conflict_term_dev,
                               const uint32_t *conflict_slot_dev, uint32_t n_groups, uint8_t *frames_dev, uint8_t *len_dev, void *stream);
int smr_wire_emit_ep_pre_accept_replies(const uint8_t *flags_dev, uint8_t row, const uint32_t *col_dev, const uint64_t *ballot_dev,
                                        const uint64_t *seq_dev, const uint32_t *deps_dev, uint32_t n_groups, uint8_t population,
                                        uint8_t *frames_dev, uint8_t *len_dev, void *stream);

/* ---- request batching front-end (host only; src/server/external.rs:323-344 get_req_batch, :697-730 the batch
 * ticker): requests queue per group; one tick turns, for every group with queued requests, up to max_batch_size
 * of them (0 = all) into one ReqBatch, FIFO; groups with an empty queue get no batch. */
typedef struct smr_batcher smr_batcher;
int smr_batcher_create(uint32_t n_groups, uint32_t max_batch_size, smr_batcher **out);
void smr_batcher_destroy(smr_batcher *b);
int smr_batcher_submit(smr_batcher *b, uint32_t group, uint64_t client, uint64_t req_id, uint8_t kind, const char *key, uint32_t key_len,
                       const char *value, uint32_t value_len);
int smr_batcher_pending(smr_batcher *b, uint64_t *n);
/* returns the number n of groups that got a batch (< 0: error, nothing consumed): groups[k], counts[k] requests,
 * bincode(ReqBatch) bytes at bytes[off[k] .. off[k + 1]) */
int64_t smr_batcher_tick(smr_batcher *b, uint32_t *groups, uint32_t *counts, uint64_t *off, uint32_t max_groups, uint8_t *bytes,
                         uint64_t cap);

/* ---- MultiPaxos near quorum reads -------------------------------------
 * One replica per group, G groups.  Replaces MultiPaxosReplica::{refresh_highest_slot, inspect_highest_slot,
 * handle_msg_read_query, handle_msg_read_query_reply} (multipaxos/quorumread.rs:8-26, 30-73, 75-188, 190-346) and the
 * ReadQueryBookkeeping set-up of treat_read_only_reqs (request.rs:55-101).  Model: keys are integers < n_keys, a value
 * is a 32-bit token (0 = none), a request batch = its Put keys + the token they write; an outstanding query
 * (client, request id) is an index q < n_queries.  Option<(slot, Option<value>)> = (state, slot, val) with state
 * 0 None, 1 Some((slot, None)), 2 Some((slot, Some(val))).  All device arrays are [..][G], group fastest. */
typedef struct smr_qread smr_qread;
typedef struct {
    uint32_t n_groups;
    uint8_t population, replica_id, n_keys, max_reads;   /* max_reads = longest Get list of one ReadQuery */
    uint32_t n_queries;
} smr_qread_cfg;
typedef struct { uint8_t *state; uint32_t *slot, *val; } smr_qread_replies;   /* [max_reads][G], or [R][max_reads][G] */
/* the replica's log as inspect_highest_slot sees it: start_slot and log_end = start_slot + insts.len() [G]; Status
 * (Committed = 3) and batch token per slot, rings of `window` slots indexed by slot % window.  mp_layout 0: status is
 * uint8 [window][G], token uint32 [window][G].  mp_layout 1: the arrays of a replica of the MultiPaxos cluster engine as
 * they lie in HBM (smr_mp_replica_log_view): wave-tiled rings, status = the low 3 bits of the 32-bit meta words.
 * run_lo / run_hi [G] (both NULL, or both set): slots in [run_lo[g], run_hi[g]) are Executed whatever the stored status
 * says -- the engine keeps the statuses its followers learn by heartbeat implicit in (run start, commit_bar).
 * run_leader [G] (NULL: none) with run_rep: where run_leader[g] != run_rep the replica is a follower whose run stores no
 * status word at all (round 4) -- a slot of [run_lo[g], log_end[g]) at or above run_hi[g] is then Accepting. */
typedef struct { const uint32_t *start_slot, *log_end; const void *status; const uint32_t *token; uint32_t window, mp_layout;
                 const uint32_t *run_lo, *run_hi; const uint8_t *run_leader; uint32_t run_rep; } smr_qread_log;
/* the log of replica `rep` of a MultiPaxos cluster as smr_qread_handle_read_query reads it, in place (device pointers
 * into the cluster's arena; valid while the cluster lives; read them between ticks) */
int smr_mp_replica_log_view(smr_mp_cluster *c, uint8_t rep, smr_qread_log *out);
int smr_qread_create(const smr_qread_cfg *cfg, smr_qread **out);
void smr_qread_destroy(smr_qread *h);
/* refresh_highest_slot for the batch saved into slot[g] (0xFFFFFFFF = no batch): put_keys[max_reads][G], 0xFF = not a Put */
int smr_qread_refresh_highest_slot(smr_qread *h, const uint32_t *slot_dev, const uint8_t *put_keys_dev, void *stream);
/* handle_msg_read_query: n[g] Gets on keys[max_reads][G] (n = 0: no message).  stable_leader[G] (may be NULL) != 0: the
 * replica is a stable leased leader and answers from the state machine kv[n_keys][G] (from_leader = 1); else every key
 * is answered by inspect_highest_slot. */
int smr_qread_handle_read_query(smr_qread *h, const uint8_t *keys_dev, const uint8_t *n_dev, const uint8_t *stable_leader_dev,
                                const uint32_t *kv_dev, const smr_qread_log *log, const smr_qread_replies *out,
                                uint8_t *from_leader_dev, void *stream);
/* the issuer's bookkeeping of query q (request.rs:62-101): n[g] reads (0 = none issued), own = its own
 * inspect_highest_slot answers (what smr_qread_handle_read_query returns without stable_leader) */
int smr_qread_issue(smr_qread *h, uint32_t q, const uint8_t *n_dev, const smr_qread_replies *own, void *stream);
/* handle_msg_read_query_reply, one reply per (peer, group) to query q: replies [R][max_reads][G], flags[R][G] bit0 a
 * reply is there, bit1 from_leader; peers in order[g] order (ackctl encoding, NULL = identity).  When the query is
 * decided: done[g] = 1 and per read outcome[max_reads][G] 1 = not found, 2 = retry on the slow path (ApiReply::rq_retry),
 * 3 = out_val; 0 elsewhere. */
int smr_qread_handle_replies(smr_qread *h, uint32_t q, const smr_qread_replies *replies, const uint8_t *flags_dev,
                             const uint32_t *order_dev, uint8_t *outcome_dev, uint32_t *out_val_dev, uint8_t *done_dev, void *stream);
/* host arrays: highest_slot[n_keys][G] (0xFFFFFFFF = never seen); live, n, rq_acks [n_queries][G]; max_replies as
 * state / slot / val [n_queries][max_reads][G]; counters[4] = values returned, retries, not found, conflicting values */
int smr_qread_dump(smr_qread *h, uint32_t *highest_slot_host, uint8_t *live_host, uint8_t *n_host, uint8_t *rq_acks_host,
                   uint8_t *mx_state_host, uint32_t *mx_slot_host, uint32_t *mx_val_host, uint64_t *counters_host);

/* ---- device-resident KV state machine ------------------------------------
 * StateMachineExecutorTask::execute (src/server/statemach.rs:193-202) per group, commands in submission order.
 * Keys < n_keys, values 32-bit tokens, 0 = None (the model of smr_qread_* and the EPaxos execution kernel). */
typedef struct smr_kv smr_kv;
int smr_kv_create(uint32_t n_groups, uint32_t n_keys, smr_kv **out);
void smr_kv_destroy(smr_kv *h);
/* n_rows commands per group, row after row: kind[n_rows][G] 0 Get / 1 Put / else none, key, val; res[n_rows][G] =
 * CommandResult: Get { value }, Put { old_value } */
int smr_kv_execute(smr_kv *h, uint32_t n_rows, const uint8_t *kind_dev, const uint8_t *key_dev, const uint32_t *val_dev,
                   uint32_t *res_dev, void *stream);
/* the table kv[n_keys][G] on the device, e.g. for smr_qread_handle_read_query's stable-leader arm */
int smr_kv_table(smr_kv *h, uint32_t **kv_dev);
int smr_kv_dump(smr_kv *h, uint32_t *kv_host);

/* ---- device-resident KV state machine over real keys and values -----------------------
 * `State = HashMap<String, String>` of src/server/statemach.rs:21-63,193-202 per group: an open-addressing table of
 * `slots` entries (power of two) + an append-only heap of heap_bytes per group on the device.  Commands name their key /
 * value bytes by (off, len) into payload_dev (e.g. the decoded ReqBatch bytes); rows are applied in order.  Results:
 * res_state 0 = None, 1 = Some -- the value (Get) / old value (Put) is bytes [res_off, res_off + res_len) of the GROUP's
 * heap strip (smr_skv_heap: strip g starts at g * heap_bytes_per_group; old bytes are never moved or reused) --, 2 = the
 * group's table or heap is full (sticky; smr_skv_stats) or the command's bytes lie outside the payload buffer. */
typedef struct smr_skv smr_skv;
int smr_skv_create(uint32_t n_groups, uint32_t slots, uint64_t heap_bytes, smr_skv **out);
void smr_skv_destroy(smr_skv *h);
int smr_skv_execute(smr_skv *h, uint32_t n_rows, const uint8_t *kind_dev, const uint8_t *payload_dev, uint64_t payload_bytes,
                    const uint32_t *key_off_dev, const uint32_t *key_len_dev, const uint32_t *val_off_dev, const uint32_t *val_len_dev,
                    uint8_t *res_state_dev, uint32_t *res_off_dev, uint32_t *res_len_dev, void *stream);
int smr_skv_heap(smr_skv *h, uint8_t **heap_dev, uint64_t *heap_bytes_per_group);
/* host copy of bytes [off, off + len) of group's heap strip (a device consumer reads them in place) */
int smr_skv_read(smr_skv *h, uint32_t group, uint32_t off, uint32_t len, uint8_t *host_buf);
int smr_skv_stats(smr_skv *h, uint32_t *n_keys_host, uint32_t *heap_used_host, uint8_t *full_host);

/* ---- batched Heartbeater (src/server/heartbeat.rs:26-296) --------------------------------
 * One object = replica `replica_id` of n_groups groups: hear timers, send ticker, reply counters / peer_alive.  Clocks and
 * randomness are explicit: calls take now_ms; a kickoff takes draw[R][G], the u32 each timer's random_range would have
 * drawn (timeout = min + draw mod (max - min + 1)).  Per-group selectors: a peer id, SMR_HB_ALL (the reference's None =
 * every peer) or SMR_NO_REPLICA (no call for that group).  smr_hb_create fails on the configurations new_and_setup
 * rejects (:69-89). */
#define SMR_HB_ALL 0xFE
typedef struct smr_hb smr_hb;
typedef struct {
    uint32_t n_groups;
    uint8_t population, replica_id;
    uint64_t hear_timeout_min_ms, hear_timeout_max_ms, send_interval_ms;
} smr_hb_cfg;
int smr_hb_create(const smr_hb_cfg *cfg, uint64_t now_ms, smr_hb **out);
void smr_hb_destroy(smr_hb *h);
int smr_hb_set_sending(smr_hb *h, const uint8_t *sending_dev, void *stream);                 /* :130-132; 0xFF = leave as is */
int smr_hb_kickoff_hear_timer(smr_hb *h, const uint8_t *peer_dev, uint64_t now_ms, const uint32_t *draw_dev, void *stream);   /* :189-210 */
/* get_event (:134-160) drained: timeouts[R][G] = 1 where HeartbeatEvent::HearTimeout { peer } is delivered now (a timer
 * that exploded and was not re-armed since), send_ticked[G] = 1 where SendTicked is (ticker on, due; late ticks skipped) */
int smr_hb_poll(smr_hb *h, uint64_t now_ms, uint8_t *timeouts_dev, uint8_t *send_ticked_dev, void *stream);
int smr_hb_clear_reply_cnts(smr_hb *h, const uint8_t *peer_dev, void *stream);               /* :223-241 */
int smr_hb_update_bcast_cnts(smr_hb *h, const uint8_t *flags_dev, uint8_t *peer_death_dev, void *stream);   /* :247-281 */
int smr_hb_update_heard_cnt(smr_hb *h, const uint8_t *peer_dev, void *stream);               /* :285-300 */
/* host arrays: deadline / exploded / cnt0 / cnt1 / rep as [R][G], is_sending / next_tick / alive as [G] */
int smr_hb_dump(smr_hb *h, uint64_t *deadline, uint8_t *exploded, uint8_t *is_sending, uint64_t *next_tick, uint64_t *cnt0,
                uint64_t *cnt1, uint8_t *rep, uint8_t *alive);

/* ---- the WAL backer file as a byte image (server/storage.rs:240-432), host only ------------------------------------------
 * StorageHubLoggerTask's file operations -- write_entry (:282), append_entry (:326), read_entry (:240), truncate_log (:351),
 * discard_log (:375) -- on a growable host buffer that stands for the backer file.  Entries are the bincode bytes the
 * smr_wal_* encoders produce WITHOUT their frame header; the log writes `[u64 BE length][bytes]` itself.  `file_size` is the
 * logger's own idea of the log's end and is an argument, as in the reference's functions. */
typedef struct smr_wallog smr_wallog;
int smr_wallog_create(smr_wallog **out);
void smr_wallog_destroy(smr_wallog *l);
int64_t smr_wallog_len(const smr_wallog *l);                                   /* the image's length */
int64_t smr_wallog_bytes(const smr_wallog *l, uint8_t *out, uint64_t cap);     /* the image itself */
int smr_wallog_write(smr_wallog *l, uint64_t file_size, const uint8_t *entry, uint64_t entry_len, uint64_t offset, uint8_t *offset_ok,
                     uint64_t *now_size);
int smr_wallog_append(smr_wallog *l, uint64_t file_size, const uint8_t *entry, uint64_t entry_len, uint64_t *now_size);
/* *entry_len = -1: None (then *end_offset = offset) */
int smr_wallog_read(const smr_wallog *l, uint64_t file_size, uint64_t offset, uint8_t *out, uint64_t cap, int64_t *entry_len,
                    uint64_t *end_offset);
int smr_wallog_truncate(smr_wallog *l, uint64_t file_size, uint64_t offset, uint8_t *ok, uint64_t *now_size);
int smr_wallog_discard(smr_wallog *l, uint64_t file_size, uint64_t offset, uint64_t keep, uint8_t *ok, uint64_t *now_size);

/* ---- LeaseManager (src/server/leaseman.rs:132-935), batched: G groups, one replica id per object -------------------------
 * Time is explicit: one smr_lease_step = the timers that exploded up to now_ms (their GrantTimeout / LeaseTimeout notices,
 * earliest first), then one notice per group, through run()'s lease-number filter (:840-926) and handle_notice (:791-835);
 * out come the actions get_action() (:275-290) would drain.  A notice is three u64 per group -- num (its LeaseNum), meta,
 * bar (accept_bar) -- with meta = kind | peer << 8 | peers << 16 | msg << 24 | held << 32 | has_bar << 40 (peers: a
 * bitmap of replica ids, or SMR_LEASE_ALL for the reference's `None`).  Actions come back in the same shape, slot i of
 * group g at [i * G + g]: num, meta = kind | peer << 8 | peers << 16 | msg << 24 | flag << 32 (flag: `held` of a
 * PromiseReply / RevokeReply / GrantRemoved, or "has accept_bar" of a Guard), bar; only the first act_n[g] slots of a
 * group are written.  smr_lease_create fails on the configurations new_and_setup rejects (:175-193). */
#define SMR_LEASE_ALL 0xFF
#define SMR_LEASE_ACT_CAP 20
enum { SMR_LEASE_N_NONE = 0, SMR_LEASE_N_NEW_GRANTS = 1, SMR_LEASE_N_DO_REVOKE = 2, SMR_LEASE_N_CLEAR_HELD = 3, SMR_LEASE_N_RECV_MSG = 4 };
enum { SMR_LEASE_M_GUARD = 0, SMR_LEASE_M_GUARD_REPLY = 1, SMR_LEASE_M_PROMISE = 2, SMR_LEASE_M_PROMISE_REPLY = 3, SMR_LEASE_M_REVOKE = 4,
       SMR_LEASE_M_REVOKE_REPLY = 5 };
enum { SMR_LEASE_A_SEND = 1, SMR_LEASE_A_BCAST = 2, SMR_LEASE_A_NEXT_REFRESH = 3, SMR_LEASE_A_GRANT_REMOVED = 4, SMR_LEASE_A_LEASE_CLEARED = 5,
       SMR_LEASE_A_GRANT_TIMEOUT = 6, SMR_LEASE_A_LEASE_TIMEOUT = 7, SMR_LEASE_A_HIGHER_NUMBER = 8, SMR_LEASE_A_GUARD_ACCEPT_BAR = 9 };
typedef struct smr_lease smr_lease;
typedef struct {
    uint32_t n_groups;
    uint8_t population, replica_id;
    uint64_t expire_timeout_ms, hb_send_interval_ms;
} smr_lease_cfg;
int smr_lease_create(const smr_lease_cfg *cfg, smr_lease **out);
void smr_lease_destroy(smr_lease *h);
/* meta_dev == NULL: timers only */
int smr_lease_step(smr_lease *h, uint64_t now_ms, const uint64_t *num_dev, const uint64_t *meta_dev, const uint64_t *bar_dev,
                   uint8_t *act_n_dev, uint64_t *act_num_dev, uint64_t *act_meta_dev, uint64_t *act_bar_dev, void *stream);
/* attempt_refresh (:296-317): call[g] != 0 where it is called, peers[g] a bitmap or SMR_LEASE_ALL; to_refresh[g] out */
int smr_lease_attempt_refresh(smr_lease *h, uint64_t now_ms, const uint8_t *call_dev, const uint8_t *peers_dev, uint8_t *to_refresh_dev,
                              void *stream);
/* grant_set (:236-243) / lease_set (:246-253) / lease_cnt (:257-259) per group, as of the last step; any pointer may be NULL */
int smr_lease_sets(smr_lease *h, uint8_t *grant_set_dev, uint8_t *lease_set_dev, uint8_t *lease_cnt_dev, void *stream);
/* host arrays: phase / deadlines as [R][G] (phase bits: 1 guards_sent, 2 guards_held, 4 promises_sent, 8 promises_held) */
int smr_lease_dump(smr_lease *h, uint64_t *active_num, uint8_t *phase, uint64_t *grant_deadline, uint64_t *hold_deadline, uint8_t *refresh_mark);

#ifdef __cplusplus
}
#endif
#endif /* SUMMERSET_HIP_H */
