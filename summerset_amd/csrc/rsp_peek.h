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
};

RspPeek rsp_peek(const smr_rsp_replica *e);

}  // namespace smr
