// What the CRaft payload store (rsp_payload.hip: smr_craft_pstore_*) reads of a Raft / CRaft replica object (raft_engine.hip):
// per ring cell [slot % W][g] the entry's term and -- CRaft -- the avail_shards_map of its codeword (`LogEntry::reqs_cw`,
// craft/mod.rs:129-150), and per group the bounds of the log the ring still holds.  Internal to the library.
#pragma once
#include "smr_common.h"

namespace smr {

struct RaftPeek {
    uint32_t G, W, R, quorum;
    const uint64_t *entry_term;         // [W][G]
    const uint8_t *entry_mask;          // [W][G] (NULL: a plain Raft replica -- no codewords)
    const uint32_t *log_len, *start_slot, *ring_lo;   // [G]: slots [max(start_slot, ring_lo), log_len) are held
};

RaftPeek raft_peek(const smr_raft_leader *l);

}  // namespace smr
