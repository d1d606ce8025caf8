// What the RSPaxos payload store (rsp_payload.hip) reads of an RSPaxos replica object (rsp_engine.hip): the per-instance
// codeword bookkeeping the engine keeps as (token, mask of shards present) -- `inst.reqs_cw` and the voted copy
// `inst.voted.1` of rspaxos/mod.rs:168-233 -- for every ring cell [slot % W][g].  Internal to the library.
#pragma once
#include "smr_common.h"

namespace smr {

struct RspPeek {
    uint32_t G, W, R, me, majority;
    const uint32_t *s_val, *s_vval;     // [W][G] batch tokens (0xFFFFFFFF: a null codeword)
    const uint8_t *s_mask, *s_vmask;    // [W][G] shards present
    // A CRaft replica's log instead (smr_craft_pstore_*; c_len != NULL): one codeword per log entry, its token a function of
    // (slot, term) -- csrc/rsp_payload.hip craft_token -- computed where it is compared; s_val / s_vval are unused, s_mask is the
    // engine's entry_mask and nothing is wanted in the voted plane
    const uint32_t *c_len, *c_start, *c_rlo;   // [G]: slots [max(start_slot, ring_lo), log_len) are held
    const uint64_t *c_term;                    // [W][G] the entries' terms
};

RspPeek rsp_peek(const smr_rsp_replica *e);

}  // namespace smr
