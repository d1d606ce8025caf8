// RepNothing (BASELINE config 1): the reference's no-replication protocol -- one
// replica, "append -> mark durable -> execute in order -> reply" -- plus the KV
// state machine every protocol executes on.  Host-only plumbing (SURVEY.md §8 a19:
// CPU path, no GPU work); it lives in the library so that the C-ABI covers the
// configuration the reference can run on a CPU.
//
// Restates  src/protocols/rep_nothing/request.rs:11-37   handle_req_batch
//           src/protocols/rep_nothing/durability.rs:10-51 handle_log_result
//           src/protocols/rep_nothing/execution.rs:10-67  handle_cmd_result
//           src/protocols/rep_nothing/mod.rs:115-127      command ids (inst << 32 | cmd)
//           src/server/statemach.rs:193-202               execute (Get / Put on a HashMap)
// under LS-1 rule 0: the WAL append and the state-machine
// commands of a batch complete right after the handler that submitted them.
// The WAL is accounted, not written: wal_offset advances by the framed size of
// the entry -- 8-byte length + bincode-standard WalEntry { reqs }
// (src/server/storage.rs:326-346; layout SURVEY.md Appendix C, unpinned).
#include <string.h>

#include <deque>
#include <string>
#include <unordered_map>
#include <vector>

#include "smr_common.h"

namespace smr {

struct RnReq { uint64_t client, req_id; uint8_t kind; std::string key, value; };
struct RnInstance { std::vector<RnReq> reqs; bool durable = false; std::vector<bool> execed; };
struct RnReply { uint64_t client, req_id; uint8_t kind; bool has_value; std::string value; };

static size_t varint_len(uint64_t v) { return v < 251 ? 1 : v < (1ull << 16) ? 3 : v < (1ull << 32) ? 5 : 9; }

// bincode-standard length of Vec<(ClientId, ApiRequest::Req { id, cmd })>
static uint64_t wal_entry_len(const std::vector<RnReq> &reqs) {
    uint64_t n = varint_len(reqs.size());
    for (const RnReq &r : reqs) {
        n += varint_len(r.client) + 1 /* ApiRequest::Req */ + varint_len(r.req_id) + 1 /* Command variant */;
        n += varint_len(r.key.size()) + r.key.size();
        if (r.kind == SMR_CMD_PUT) n += varint_len(r.value.size()) + r.value.size();
    }
    return n;
}

}  // namespace smr

using namespace smr;

struct smr_repnothing {
    std::vector<RnInstance> insts;
    std::unordered_map<std::string, std::string> state;   // statemach.rs: State = HashMap<String, String>
    std::deque<RnReply> replies;
    uint64_t wal_offset = 0, n_execed = 0;
};

extern "C" {

int smr_repnothing_create(smr_repnothing **out) {
    if (!out) return fail(SMR_ERR_ARG, "rep_nothing: null out");
    *out = new smr_repnothing();
    return SMR_OK;
}

void smr_repnothing_destroy(smr_repnothing *h) { delete h; }

int smr_repnothing_submit_batch(smr_repnothing *h, uint32_t n, const uint64_t *client, const uint64_t *req_id,
                                const uint8_t *kind, const char *const *key, const uint32_t *key_len,
                                const char *const *value, const uint32_t *value_len, uint64_t *inst_idx) {
    if (!h || !client || !req_id || !kind || !key || !key_len) return fail(SMR_ERR_ARG, "rep_nothing: null argument");
    if (n == 0) return fail(SMR_ERR_ARG, "rep_nothing: empty batch");            // request.rs:16 debug_assert
    RnInstance inst;                                                              // request.rs:18-24
    inst.reqs.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
        if (kind[i] != SMR_CMD_GET && kind[i] != SMR_CMD_PUT) return fail(SMR_ERR_ARG, "rep_nothing: unknown command kind");
        RnReq r{client[i], req_id[i], kind[i], std::string(key[i], key_len[i]), std::string()};
        if (kind[i] == SMR_CMD_PUT) {
            if (!value || !value_len) return fail(SMR_ERR_ARG, "rep_nothing: Put without value arrays");
            r.value.assign(value[i], value_len[i]);
        }
        inst.reqs.push_back(std::move(r));
    }
    inst.execed.assign(n, false);
    const uint64_t idx = h->insts.size();
    if (idx >> 32) return fail(SMR_ERR_STATE, "rep_nothing: instance index exceeds 32 bits");   // mod.rs:116
    h->insts.push_back(std::move(inst));
    if (inst_idx) *inst_idx = idx;
    // WAL append completes (durability.rs:20-38)
    RnInstance &in = h->insts[idx];
    h->wal_offset += 8 + wal_entry_len(in.reqs);
    in.durable = true;
    // commands execute in submission order (durability.rs:41-49, statemach.rs:193-202) and each
    // result is answered to its client (execution.rs:44-58)
    for (uint32_t c = 0; c < n; c++) {
        const RnReq &r = in.reqs[c];
        RnReply rep{r.client, r.req_id, r.kind, false, std::string()};
        if (r.kind == SMR_CMD_GET) {
            auto it = h->state.find(r.key);
            if (it != h->state.end()) { rep.has_value = true; rep.value = it->second; }
        } else {
            auto it = h->state.find(r.key);
            if (it != h->state.end()) { rep.has_value = true; rep.value = it->second; it->second = r.value; }   // old_value
            else h->state.emplace(r.key, r.value);
        }
        in.execed[c] = true;                                                      // execution.rs:42
        h->n_execed++;
        h->replies.push_back(std::move(rep));
    }
    return SMR_OK;
}

int smr_repnothing_poll_reply(smr_repnothing *h, uint64_t *client, uint64_t *req_id, uint8_t *kind, int *has_value,
                              char *value_buf, uint32_t value_cap, uint32_t *value_len) {
    if (!h) return fail(SMR_ERR_ARG, "rep_nothing: null handle");
    if (h->replies.empty()) return 0;
    const RnReply &r = h->replies.front();
    if (r.has_value && r.value.size() > value_cap) {
        if (value_len) *value_len = (uint32_t)r.value.size();
        return fail(SMR_ERR_ARG, "rep_nothing: value buffer too small");
    }
    if (client) *client = r.client;
    if (req_id) *req_id = r.req_id;
    if (kind) *kind = r.kind;
    if (has_value) *has_value = r.has_value ? 1 : 0;
    if (value_len) *value_len = (uint32_t)r.value.size();
    if (r.has_value && value_buf) memcpy(value_buf, r.value.data(), r.value.size());
    h->replies.pop_front();
    return 1;
}

int smr_repnothing_stats(smr_repnothing *h, uint64_t *n_insts, uint64_t *wal_offset, uint64_t *n_execed,
                         uint64_t *n_keys) {
    if (!h) return fail(SMR_ERR_ARG, "rep_nothing: null handle");
    if (n_insts) *n_insts = h->insts.size();
    if (wal_offset) *wal_offset = h->wal_offset;
    if (n_execed) *n_execed = h->n_execed;
    if (n_keys) *n_keys = h->state.size();
    return SMR_OK;
}

}  // extern "C"
