/*
 * oracle/ep_oracle.c -- CPU restatement of the EPaxos command-leader / acceptor
 * hot path of Summerset over G independent groups, one replica (id `me`) per
 * group: dependency and sequence computation, PreAccept handling, the
 * fast-quorum decision on PreAcceptReplies, the slow-path Accept tally and the
 * commit bars.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/mp_oracle.c header for the rules).
 *
 * Follows src/protocols/epaxos/:
 *   handle_req_batch                 request.rs:10-108
 *   first_null_slot / null_instance  mod.rs:467-496
 *   identify_deps, refresh_highest_cols, max_seq_num, DepSet::union
 *                                    dependency.rs:85-167
 *   fast_quorum_eligibility          dependency.rs:175-240
 *   get_enough_identical             dependency.rs:333-367
 *   handle_msg_pre_accept            messages.rs:10-93
 *   handle_msg_pre_accept_reply      messages.rs:96-270
 *   handle_msg_accept                messages.rs:273-345
 *   handle_msg_accept_reply          messages.rs:348-436
 *   handle_msg_commit_notice         messages.rs:438-508
 *   handle_logged_{pre_accept,accept,commit}_slot   durability.rs:10-163
 *   quorum sizes, default ballot     mod.rs:500-514,693-698
 * WAL completions are inline (LS-1 rule 0).  A request batch is one Put on one
 * key of a small key space (key id < n_keys), which is
 * all identify_deps / refresh_highest_cols look at.  Timers are inputs (the set
 * of peers whose hear timer "exploded" is an argument of the reply handler, a
 * HearTimeout an argument of orc_ep_heartbeat_timeout).
 *
 * Explicit prepare (recovery of a suspected peer's row):
 *   heartbeat_timeout                heartbeat.rs:17-125 (the protocol part: the
 *                                    fast-quorum re-evaluation of my PreAccepting
 *                                    instances, ExpPrepare for the peer's row, my
 *                                    own ExpPrepareReply)
 *   make_greater_ballot              mod.rs:500-508
 *   handle_msg_exp_prepare           messages.rs:511-574
 *   handle_msg_exp_prepare_reply     messages.rs:577-821
 *   exp_prepare_next_step            dependency.rs:249-327
 * exp_prepare_voteds is a HashMap in the reference and exp_prepare_next_step
 * takes "the last Committed / Accepting / PreAccepting reply in iteration
 * order" as its representative (dependency.rs:266-273): arbitrary when two
 * PreAccepting replies at the same ballot differ.  Canonical choice here (and in
 * the engine): iteration in peer-id order, i.e. the HIGHEST peer id of a status.
 * With recovery, a replica leads instances outside its own row and messages
 * name their slot's row apart from their sender: the handlers below take an
 * optional `row` array (NULL: the sender's row for requests, my own row for
 * replies -- the only cases there are without recovery).
 *
 * Dependency-graph execution (execution.rs:25-149 attempt_execution, :152-211
 * handle_cmd_result, durability.rs:136-160 the attempts after a commit-bar
 * advance) is restated literally and switched on with orc_ep_set_execute():
 * the breadth-first walk over deps, the graph it builds, Tarjan's algorithm on
 * it, the per-component sort by seq.  Two facts about the reference that the
 * restatement keeps (tests/test_oracle_ep_exec.py derives them by hand):
 *   (1) the walk adds an edge from the slot popped just BEFORE a new node to
 *       that node (execution.rs:57-59), not from a node to its dependencies.
 *       Every node therefore has at most one incoming edge, made when it joins:
 *       the graph is a forest, every strongly connected component is a single
 *       node (n_multi_scc counts the exceptions: none), and the submission
 *       order is the depth-first post-order of that forest.
 *   (2) GraphMap::add_edge inserts a missing endpoint, so a slot that was
 *       pruned as already executing and is popped right before a new node
 *       re-enters the graph and is submitted AGAIN (n_reexec).
 * The order then depends on three published behaviours of petgraph 0.8 (not
 * vendored, "parity unpinned"): GraphMap keeps nodes and edges in insertion
 * order and into_graph() keeps both orders; Graph::neighbors() walks a node's
 * outgoing edges newest first; tarjan_scc() starts from nodes in index order,
 * recurses over neighbors() and emits components in post-order.
 * The state machine is one Put per instance: kv[key] = token(row, col), result
 * = the old token; the digest chains (token, old token) in submission order.
 * Rule 0: command results arrive right after the handler that
 * submitted them returns, in submission order.
 * Harness guard: an instance that left its row's ring of W columns counts as
 * executed (pruned) and cannot re-enter the graph (n_unheld counts the pops).
 *
 * PARITY STATUS: "parity unpinned" -- the reference has no unit tests or
 * fixtures for these handlers and cannot be built here; pinned by hand-derived
 * traces (tests/test_oracle_ep.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ST_NULL = 0, ST_PREACCEPTING = 1, ST_ACCEPTING = 2, ST_COMMITTED = 3, ST_EXECUTING = 4, ST_EXECUTED = 5 };
#define MAXR 8
#define NONE 0xFFFFFFFFu            /* Option<usize>::None in a DepSet */
#define NO_KEY 0xFF                 /* empty ReqBatch */
#define NO_REP 0xFF

typedef struct { uint32_t c[MAXR]; } DepSet;

typedef struct {
    uint64_t bal, seq;
    uint8_t status, key;
    DepSet deps;
    uint8_t has_lbk, has_rbk, source, avoid_fast_path;
    /* LeaderBookkeeping */
    uint8_t pa_acks, acc_acks;
    uint8_t pa_has[MAXR]; uint64_t pa_seq[MAXR]; DepSet pa_deps[MAXR];   /* pre_accept_replies: HashMap<peer, (seq, deps)> */
    uint8_t xp_acks; uint64_t xp_max_bal;                                 /* exp_prepare_acks, exp_prepare_max_bal */
    uint8_t xp_has[MAXR], xp_status[MAXR], xp_key[MAXR]; uint64_t xp_seq[MAXR]; DepSet xp_deps[MAXR];   /* exp_prepare_voteds */
} Inst;

typedef struct { uint8_t row; uint32_t col; } Slot;

typedef struct {
    uint8_t id, population, simple_q, super_q;
    uint64_t n_xp_commit, n_xp_accept, n_xp_pre_accept, n_xp_noop;   /* explicit-prepare outcomes */
    uint32_t n_keys, W;
    Inst *rows[MAXR]; uint32_t len[MAXR], cap[MAXR];
    uint32_t start_col;
    uint32_t commit_bars[MAXR], exec_bars[MAXR];
    DepSet *highest_cols; uint8_t *hc_present;      /* HashMap<key, DepSet> */
    uint64_t n_fast, n_slow, n_accept_commits;
    /* execution */
    uint8_t execute;
    Slot *execq; uint32_t n_execq, cap_execq;       /* commands submitted to the state machine, results pending */
    Slot *sublog; uint32_t n_sublog, cap_sublog;    /* every submission since the last orc_ep_take_submissions */
    uint64_t *kv;                                   /* [n_keys] token of the last Put, 0 = none */
    uint64_t digest;
    uint64_t n_exec, n_reexec, n_unheld, n_multi_scc, n_attempts, n_aborts;
} EpRep;

typedef struct { uint32_t G; uint8_t R; EpRep *reps; } EpCl;

static DepSet dep_empty(void) { DepSet d; for (int i = 0; i < MAXR; i++) d.c[i] = NONE; return d; }
static void dep_union(DepSet *s, const DepSet *o, int R) {          /* dependency.rs:85-97 */
    for (int i = 0; i < R; i++) {
        if (s->c[i] != NONE) { if (o->c[i] != NONE && o->c[i] > s->c[i]) s->c[i] = o->c[i]; }
        else s->c[i] = o->c[i];
    }
}
static int dep_eq(const DepSet *a, const DepSet *b, int R) {
    for (int i = 0; i < R; i++) if (a->c[i] != b->c[i]) return 0;
    return 1;
}

static Inst null_instance(void) {                                   /* mod.rs:467-480 */
    Inst in; memset(&in, 0, sizeof(in));
    in.status = ST_NULL; in.key = NO_KEY; in.deps = dep_empty(); in.source = NO_REP;
    for (int p = 0; p < MAXR; p++) { in.pa_deps[p] = dep_empty(); in.xp_deps[p] = dep_empty(); in.xp_key[p] = NO_KEY; }
    return in;
}
static void row_push(EpRep *r, int row, Inst in) {
    if (r->len[row] == r->cap[row]) {
        r->cap[row] = r->cap[row] ? r->cap[row] * 2 : 16;
        r->rows[row] = (Inst *)realloc(r->rows[row], sizeof(Inst) * r->cap[row]);
    }
    r->rows[row][r->len[row]++] = in;
}
static Inst *at(EpRep *r, int row, uint32_t col) { return &r->rows[row][col - r->start_col]; }
/* harness guard shared with the engine (rings of W instances per row): is the column still held? */
static int held(const EpRep *r, int row, uint32_t col) {
    uint32_t end = r->start_col + r->len[row];
    return col >= r->start_col && col < end && col + r->W >= end;
}

void *orc_ep_new(uint32_t G, uint8_t R, uint8_t me, uint32_t W, uint32_t n_keys, uint8_t optimized_quorum) {
    EpCl *cl = (EpCl *)calloc(1, sizeof(EpCl));
    cl->G = G; cl->R = R;
    cl->reps = (EpRep *)calloc(G, sizeof(EpRep));
    for (uint32_t g = 0; g < G; g++) {
        EpRep *r = &cl->reps[g];
        r->id = me; r->population = R; r->W = W; r->n_keys = n_keys;
        r->simple_q = (uint8_t)(R / 2 + 1);                          /* mod.rs:693 */
        r->super_q = optimized_quorum ? (uint8_t)(R / 2 + (R / 2 + 1) / 2) : (uint8_t)((R / 2) * 2);   /* :694-698 */
        r->highest_cols = (DepSet *)calloc(n_keys, sizeof(DepSet));
        r->hc_present = (uint8_t *)calloc(n_keys, 1);
        r->kv = (uint64_t *)calloc(n_keys, sizeof(uint64_t));
    }
    return cl;
}
void orc_ep_set_execute(void *h, uint8_t on) {
    EpCl *cl = (EpCl *)h;
    for (uint32_t g = 0; g < cl->G; g++) cl->reps[g].execute = on;
}
void orc_ep_free(void *h) {
    EpCl *cl = (EpCl *)h;
    for (uint32_t g = 0; g < cl->G; g++) {
        for (int i = 0; i < MAXR; i++) free(cl->reps[g].rows[i]);
        free(cl->reps[g].highest_cols); free(cl->reps[g].hc_present);
        free(cl->reps[g].kv); free(cl->reps[g].execq); free(cl->reps[g].sublog);
    }
    free(cl->reps); free(cl);
}

/* dependency.rs:101-109; harness guard shared with the engine: an instance older than the
 * row's last W columns is no longer held (the engine keeps rings of W instances per row) */
static uint64_t max_seq_num(EpRep *r, const DepSet *deps) {
    uint64_t m = 0;
    for (int row = 0; row < r->population; row++) {
        uint32_t c = deps->c[row];
        if (c == NONE) continue;
        if (!held(r, row, c)) continue;
        uint64_t s = at(r, row, c)->seq;
        if (s > m) m = s;
    }
    return m;
}
static DepSet identify_deps(EpRep *r, uint8_t key) {                /* dependency.rs:113-137 */
    DepSet d = dep_empty();
    if (key != NO_KEY && r->hc_present[key]) dep_union(&d, &r->highest_cols[key], r->population);
    return d;
}
static void refresh_highest_cols(EpRep *r, int row, uint32_t col, uint8_t key) {   /* dependency.rs:141-167 */
    if (key == NO_KEY) return;
    if (r->hc_present[key]) {
        uint32_t *hc = &r->highest_cols[key].c[row];
        if (*hc != NONE) { if (col > *hc) *hc = col; } else *hc = col;
    } else {
        r->highest_cols[key] = dep_empty();
        r->highest_cols[key].c[row] = col;
        r->hc_present[key] = 1;
    }
}

/* ---- execution.rs:25-149 ---- */
#define TOKEN(row, col) ((((uint64_t)(row) + 1) << 32) | (uint64_t)(col))
#define DG_MUL 0x100000001B3ull

typedef struct { Slot *v; uint32_t n, cap; } SlotVec;
static void sv_push(SlotVec *s, Slot x) {
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 32; s->v = (Slot *)realloc(s->v, sizeof(Slot) * s->cap); }
    s->v[s->n++] = x;
}
static int sv_find(const SlotVec *s, Slot x) {
    for (uint32_t i = 0; i < s->n; i++) if (s->v[i].row == x.row && s->v[i].col == x.col) return (int)i;
    return -1;
}

/* petgraph::algo::tarjan_scc as of 0.6-0.8 (one index per node that doubles as the component mark;
 * components come out in post-order = reverse topological order) over a Graph given as per-node
 * lists of outgoing edges, newest edge first */
typedef struct {
    uint32_t n; const uint32_t *head, *next, *dst;    /* adjacency: head[v] -> edge, next[edge], dst[edge]; ~0 = end */
    uint64_t *rootindex;                              /* 0 = None */
    uint64_t index, componentcount;
    uint32_t *stack; uint32_t n_stack;
    uint32_t *out; uint32_t n_out;                    /* component members, concatenated */
    uint32_t *out_end; uint32_t n_comp;               /* end offset of each component in out */
} Tarjan;
static void tarjan_visit(Tarjan *t, uint32_t v) {
    int v_is_local_root = 1;
    t->rootindex[v] = t->index++;
    for (uint32_t e = t->head[v]; e != ~0u; e = t->next[e]) {
        uint32_t w = t->dst[e];
        if (t->rootindex[w] == 0) tarjan_visit(t, w);
        if (t->rootindex[w] < t->rootindex[v]) { t->rootindex[v] = t->rootindex[w]; v_is_local_root = 0; }
    }
    if (v_is_local_root) {
        uint64_t indexadjustment = 1, c = t->componentcount;
        uint32_t start = t->n_stack;
        while (start > 0 && !(t->rootindex[v] > t->rootindex[t->stack[start - 1]])) {
            t->rootindex[t->stack[start - 1]] = c; indexadjustment++; start--;
        }
        t->rootindex[v] = c;
        t->stack[t->n_stack++] = v;
        for (uint32_t i = start; i < t->n_stack; i++) t->out[t->n_out++] = t->stack[i];
        t->out_end[t->n_comp++] = t->n_out;
        t->n_stack = start;
        t->index -= indexadjustment;
        t->componentcount--;
    } else t->stack[t->n_stack++] = v;
}

static void handle_cmd_result(EpRep *r, int row, uint32_t col);

/* execution.rs:25-149, sync_exec = false */
static int attempt_execution(EpRep *r, int trow, uint32_t tcol) {
    const int R = r->population;
    SlotVec nodes = {0}, ea = {0}, eb = {0}, queue = {0};          /* GraphMap nodes; edges (a -> b); dep_queue */
    uint32_t qh = 0;
    Slot last = {0, 0}; int has_last = 0;
    r->n_attempts++;
    sv_push(&queue, (Slot){(uint8_t)trow, tcol});
    while (qh < queue.n) {
        Slot s = queue.v[qh++];
        if (s.col >= r->commit_bars[s.row]) {                      /* :41-45 dependency not committed */
            r->n_aborts++;
            free(nodes.v); free(ea.v); free(eb.v); free(queue.v);
            return 0;
        }
        int is_held = held(r, s.row, s.col);
        if (!is_held) r->n_unheld++;
        if (s.col < r->start_col || !is_held || at(r, s.row, s.col)->status >= ST_EXECUTING) {
            /* :46-51 already submitted: prune it and what it depends on */
        } else if (sv_find(&nodes, s) >= 0) {
            /* :52-54 already in the graph */
        } else {
            sv_push(&nodes, s);                                    /* :56-59 */
            if (has_last && held(r, last.row, last.col)) {
                sv_push(&ea, last); sv_push(&eb, s);
                if (sv_find(&nodes, last) < 0) { sv_push(&nodes, last); r->n_reexec++; }   /* GraphMap::add_edge inserts it */
            }
            Inst *in = at(r, s.row, s.col);                        /* :62-77 */
            for (int i = 0; i < R; i++)
                if (in->deps.c[i] != NONE) sv_push(&queue, (Slot){(uint8_t)i, in->deps.c[i]});
            if (s.col > r->start_col) sv_push(&queue, (Slot){s.row, s.col - 1});
        }
        last = s; has_last = 1;
    }
    /* into_graph(): node i = i-th inserted node; edges re-added in insertion order, so each node's
     * list of outgoing edges ends up newest first */
    const uint32_t n = nodes.n, m = ea.n;
    uint32_t *head = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1)), *next = (uint32_t *)malloc(sizeof(uint32_t) * (m + 1));
    uint32_t *dst = (uint32_t *)malloc(sizeof(uint32_t) * (m + 1));
    for (uint32_t i = 0; i < n; i++) head[i] = ~0u;
    for (uint32_t e = 0; e < m; e++) {
        uint32_t a = (uint32_t)sv_find(&nodes, ea.v[e]), b = (uint32_t)sv_find(&nodes, eb.v[e]);
        dst[e] = b; next[e] = head[a]; head[a] = e;
    }
    Tarjan t; memset(&t, 0, sizeof(t));
    t.n = n; t.head = head; t.next = next; t.dst = dst;
    t.rootindex = (uint64_t *)calloc(n + 1, sizeof(uint64_t));
    t.index = 1; t.componentcount = UINT64_MAX;
    t.stack = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    t.out = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    t.out_end = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    for (uint32_t v = 0; v < n; v++) if (t.rootindex[v] == 0) tarjan_visit(&t, v);
    uint32_t lo = 0;
    for (uint32_t ci = 0; ci < t.n_comp; ci++) {
        uint32_t hi = t.out_end[ci];
        if (hi - lo > 1) r->n_multi_scc++;
        for (uint32_t i = lo + 1; i < hi; i++) {                   /* :99-102 sort_by_key(seq), stable */
            uint32_t x = t.out[i]; uint64_t sx = at(r, nodes.v[x].row, nodes.v[x].col)->seq;
            uint32_t j = i;
            while (j > lo && at(r, nodes.v[t.out[j - 1]].row, nodes.v[t.out[j - 1]].col)->seq > sx) { t.out[j] = t.out[j - 1]; j--; }
            t.out[j] = x;
        }
        for (uint32_t i = lo; i < hi; i++) {                       /* :105-142 */
            Slot s = nodes.v[t.out[i]];
            Inst *in = at(r, s.row, s.col);
            if (in->key != NO_KEY) {                               /* submit_cmd: the state machine runs them in order */
                uint64_t tok = TOKEN(s.row, s.col), old = r->kv[in->key];
                r->kv[in->key] = tok;
                r->digest = (r->digest ^ tok) * DG_MUL; r->digest = (r->digest ^ old) * DG_MUL;
                r->n_exec++;
                if (r->n_execq == r->cap_execq) {
                    r->cap_execq = r->cap_execq ? r->cap_execq * 2 : 32;
                    r->execq = (Slot *)realloc(r->execq, sizeof(Slot) * r->cap_execq);
                }
                r->execq[r->n_execq++] = s;
                if (r->n_sublog == r->cap_sublog) {
                    r->cap_sublog = r->cap_sublog ? r->cap_sublog * 2 : 32;
                    r->sublog = (Slot *)realloc(r->sublog, sizeof(Slot) * r->cap_sublog);
                }
                r->sublog[r->n_sublog++] = s;
            }
            in->status = ST_EXECUTING;
        }
        lo = hi;
    }
    free(head); free(next); free(dst); free(t.rootindex); free(t.stack); free(t.out); free(t.out_end);
    free(nodes.v); free(ea.v); free(eb.v); free(queue.v);
    return 1;
}

/* execution.rs:152-211 (one command per instance) */
static void handle_cmd_result(EpRep *r, int row, uint32_t col) {
    if (col < r->start_col || !held(r, row, col)) return;
    at(r, row, col)->status = ST_EXECUTED;
    if (col == r->exec_bars[row]) {
        while (r->exec_bars[row] < r->start_col + r->len[row] && held(r, row, r->exec_bars[row])) {
            if (at(r, row, r->exec_bars[row])->status < ST_EXECUTED) break;
            r->exec_bars[row]++;
        }
    }
}
/* Rule 0: the results of the commands a handler submitted, in submission order, once it has returned */
static void drain_exec(EpRep *r) {
    for (uint32_t i = 0; i < r->n_execq; i++) handle_cmd_result(r, r->execq[i].row, r->execq[i].col);
    r->n_execq = 0;
}

/* durability.rs:104-163 */
static void handle_logged_commit_slot(EpRep *r, int row, uint32_t col) {
    if (col < r->start_col) return;
    if (col == r->commit_bars[row]) {
        int advanced = 0;
        while (r->commit_bars[row] < r->start_col + r->len[row] && held(r, row, r->commit_bars[row])) {
            Inst *in = at(r, row, r->commit_bars[row]);
            if (in->status < ST_COMMITTED) break;
            else if (in->key == NO_KEY) in->status = ST_EXECUTED;
            r->commit_bars[row]++;
            advanced = 1;
        }
        if (advanced && r->execute) {                              /* :136-160 */
            if (attempt_execution(r, row, r->commit_bars[row] - 1)) {
                Slot re[MAXR]; int n_re = 0;
                for (int q = 0; q < r->population; q++) {
                    uint32_t c = r->commit_bars[q];
                    if (c > r->exec_bars[q] && held(r, q, c - 1) && at(r, q, c - 1)->status == ST_COMMITTED)
                        re[n_re++] = (Slot){(uint8_t)q, c - 1};
                }
                for (int i = 0; i < n_re; i++) attempt_execution(r, re[i].row, re[i].col);
            }
        }
    }
}

/* dependency.rs:333-367 over the replies held so far, in peer-id order (the reference iterates a
 * HashMap; the outcome does not depend on the order: classes of equal (seq, deps)) */
static int get_enough_identical(EpRep *r, Inst *in, uint8_t thresh, uint64_t *seq, DepSet *deps, uint8_t *max_cnt) {
    int idx[MAXR], n = 0;
    for (int p = 0; p < r->population; p++) if (in->pa_has[p]) idx[n++] = p;
    uint8_t visited[MAXR] = {0};
    int first = 0; uint8_t mc = 1;
    visited[0] = 1;
    while (first < n) {
        int next_first = n; uint8_t same = 1;
        for (int i = first + 1; i < n; i++) {
            if (!visited[i]) {
                if (in->pa_seq[idx[i]] == in->pa_seq[idx[first]] &&
                    dep_eq(&in->pa_deps[idx[i]], &in->pa_deps[idx[first]], r->population)) { visited[i] = 1; same++; }
                else if (next_first == n) next_first = i;
            }
        }
        if (same >= thresh) { *seq = in->pa_seq[idx[first]]; *deps = in->pa_deps[idx[first]]; *max_cnt = same; return 1; }
        first = next_first;
        if (same > mc) mc = same;
    }
    *max_cnt = mc;
    return 0;
}

/* dependency.rs:175-240: 0 = undecided, else the Status to enter with (seq, deps) */
static int fast_quorum_eligibility(EpRep *r, Inst *in, uint8_t exploded, uint64_t *seq, DepSet *deps) {
    uint8_t all_cnt = (uint8_t)__builtin_popcount(in->pa_acks);
    if (all_cnt < r->simple_q) return 0;
    if (!in->avoid_fast_path) {
        uint8_t bad = 0;
        for (int p = 0; p < r->population; p++)
            if (!((in->pa_acks >> p) & 1) && p != r->id && ((exploded >> p) & 1)) bad++;
        uint8_t max_cnt;
        if (get_enough_identical(r, in, r->super_q, seq, deps, &max_cnt)) return ST_COMMITTED;
        if (max_cnt + (r->population - bad - all_cnt) >= r->super_q) return 0;      /* :221-236 */
    }
    *seq = 0; *deps = dep_empty();                                    /* union of deps, max of seqs */
    for (int p = 0; p < r->population; p++)
        if (in->pa_has[p]) { dep_union(deps, &in->pa_deps[p], r->population); if (in->pa_seq[p] > *seq) *seq = in->pa_seq[p]; }
    return ST_ACCEPTING;
}

static void handle_msg_accept_reply(EpRep *r, uint8_t peer, int row, uint32_t col, uint64_t ballot);

/* messages.rs:96-270; ballot == 0: "failure suspected" re-evaluation */
static void handle_msg_pre_accept_reply(EpRep *r, uint8_t peer, int row, uint32_t col, uint64_t ballot, uint64_t seq,
                                        const DepSet *deps, uint8_t exploded) {
    if (col < r->start_col) return;
    if (col >= r->start_col + r->len[row] || !held(r, row, col)) return;   /* :125-127 */
    Inst *in = at(r, row, col);
    if (in->status != ST_PREACCEPTING || (ballot > 0 && in->bal != ballot) || !in->has_lbk) return;   /* :129-134 */
    if ((in->pa_acks >> peer) & 1) return;                          /* :136-138 */
    if (ballot > 0) {                                                /* :141-144 */
        in->pa_has[peer] = 1; in->pa_seq[peer] = seq; in->pa_deps[peer] = *deps;
        in->pa_acks |= (uint8_t)(1u << peer);
    }
    uint64_t dseq; DepSet ddeps;
    int next = fast_quorum_eligibility(r, in, exploded, &dseq, &ddeps);
    if (next == ST_COMMITTED) {                                      /* :158-206 */
        in->status = ST_COMMITTED; in->seq = dseq; in->deps = ddeps;
        r->n_fast++;
        handle_logged_commit_slot(r, row, col);
    } else if (next == ST_ACCEPTING) {                               /* :209-262 */
        in->status = ST_ACCEPTING; in->seq = dseq; in->deps = ddeps;
        r->n_slow++;
        handle_msg_accept_reply(r, r->id, row, col, in->bal);         /* durability.rs:78-83: my own AcceptSlot completion */
    }
}

/* messages.rs:348-436 */
static void handle_msg_accept_reply(EpRep *r, uint8_t peer, int row, uint32_t col, uint64_t ballot) {
    if (col < r->start_col) return;
    if (col >= r->start_col + r->len[row] || !held(r, row, col)) return;
    Inst *in = at(r, row, col);
    if (in->status != ST_ACCEPTING || in->bal != ballot || !in->has_lbk) return;     /* :371-376 */
    if ((in->acc_acks >> peer) & 1) return;
    in->acc_acks |= (uint8_t)(1u << peer);
    if (__builtin_popcount(in->acc_acks) >= r->simple_q) {           /* :386 */
        in->status = ST_COMMITTED;
        r->n_accept_commits++;
        handle_logged_commit_slot(r, row, col);
    }
}

/* LeaderBookkeeping { .. } as request.rs:48-57 and heartbeat.rs:88-97 make it */
static void fresh_leader_bk(Inst *in) {
    in->has_lbk = 1; in->pa_acks = 0; in->acc_acks = 0; in->xp_acks = 0; in->xp_max_bal = 0;
    for (int p = 0; p < MAXR; p++) { in->pa_has[p] = 0; in->xp_has[p] = 0; }
}

/* request.rs:10-108 + the command leader's own PreAcceptSlot completion (durability.rs:25-35).
 * Output: the PreAccept message it broadcasts. */
void orc_ep_propose(void *h, const uint8_t *key, const uint8_t *exploded, uint8_t *m_flags, uint32_t *m_col,
                    uint64_t *m_seq, uint32_t *m_deps) {
    EpCl *cl = (EpCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        EpRep *r = &cl->reps[g];
        m_flags[g] = 0; m_col[g] = 0; m_seq[g] = 0;
        for (int i = 0; i < cl->R; i++) m_deps[(size_t)i * G + g] = NONE;
        if (key[g] == NO_KEY) continue;
        int row = r->id;
        uint32_t col = NONE;                                         /* mod.rs:485-496 first_null_slot */
        for (uint32_t c = r->exec_bars[row]; c < r->start_col + r->len[row]; c++)
            if (held(r, row, c) && at(r, row, c)->status == ST_NULL) { col = c; break; }
        if (col == NONE) { row_push(r, row, null_instance()); col = r->start_col + r->len[row] - 1; }
        DepSet deps = identify_deps(r, key[g]);
        uint64_t seq = 1 + max_seq_num(r, &deps);
        Inst *in = at(r, row, col);
        in->bal = (uint64_t)(r->id + 1);                              /* make_default_ballot: (0 << 8) | (id + 1) */
        in->seq = seq; in->deps = deps; in->key = key[g];
        refresh_highest_cols(r, row, col, key[g]);
        fresh_leader_bk(in);
        in->status = ST_PREACCEPTING;
        m_flags[g] = 1; m_col[g] = col; m_seq[g] = seq;
        for (int i = 0; i < cl->R; i++) m_deps[(size_t)i * G + g] = deps.c[i];
        handle_msg_pre_accept_reply(r, r->id, row, col, in->bal, seq, &deps, exploded ? exploded[g] : 0); drain_exec(r);
    }
}

/* messages.rs:10-93 + the acceptor's PreAcceptSlot completion (durability.rs:36-54): the reply */
void orc_ep_handle_pre_accept(void *h, const uint8_t *flags, const uint8_t *peer, const uint32_t *col,
                              const uint64_t *ballot, const uint64_t *seq, const uint32_t *deps, const uint8_t *key,
                              uint8_t *r_flags, uint64_t *r_ballot, uint64_t *r_seq, uint32_t *r_deps, const uint8_t *rows) {
    EpCl *cl = (EpCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        EpRep *r = &cl->reps[g];
        r_flags[g] = 0; r_ballot[g] = 0; r_seq[g] = 0;
        for (int i = 0; i < cl->R; i++) r_deps[(size_t)i * G + g] = NONE;
        if (!(flags[g] & 1)) continue;
        int row = rows ? rows[g] : peer[g];                          /* (without recovery: the command leader's own row) */
        uint32_t c = col[g];
        if (c < r->start_col || (c < r->start_col + r->len[row] && !held(r, row, c))) continue;
        while (r->start_col + r->len[row] <= c) row_push(r, row, null_instance());   /* :33-36 */
        Inst *in = at(r, row, c);
        if (ballot[g] >= in->bal) {                                  /* :40 */
            DepSet d; for (int i = 0; i < MAXR; i++) d.c[i] = i < cl->R ? deps[(size_t)i * G + g] : NONE;
            DepSet my = identify_deps(r, key[g]);
            dep_union(&d, &my, r->population);
            uint64_t s = seq[g], ms = 1 + max_seq_num(r, &my);
            if (ms > s) s = ms;
            in->bal = ballot[g]; in->status = ST_PREACCEPTING; in->seq = s; in->deps = d; in->key = key[g];
            refresh_highest_cols(r, row, c, key[g]);
            in->has_rbk = 1; in->source = peer[g];
            /* WAL completion: leader_bk takes precedence (durability.rs:25), else reply to source */
            if (in->has_lbk) { handle_msg_pre_accept_reply(r, r->id, row, c, in->bal, in->seq, &in->deps, 0); drain_exec(r); }
            else {
                r_flags[g] = 1; r_ballot[g] = in->bal; r_seq[g] = in->seq;
                for (int i = 0; i < cl->R; i++) r_deps[(size_t)i * G + g] = in->deps.c[i];
            }
        }
    }
}

static uint32_t ctl_order(uint32_t ctl, int i) { return (ctl >> (3 * i)) & 7u; }
#define CTL_IDENTITY 0x00FAC688u

/* PreAcceptReplies to my instance (me, col[g]): per peer [R][G] ballot, seq, deps[R][R][G];
 * flags[R][G] bit0 = present; peers in `order` order (ackctl encoding).  decision: 0 / ST_* */
void orc_ep_handle_pre_accept_replies(void *h, const uint32_t *col, const uint64_t *ballot, const uint64_t *seq,
                                      const uint32_t *deps, const uint8_t *flags, const uint32_t *order,
                                      const uint8_t *exploded, uint8_t *decision, uint64_t *d_seq, uint32_t *d_deps,
                                      const uint8_t *rows) {
    EpCl *cl = (EpCl *)h;
    const uint32_t G = cl->G; const int R = cl->R;
    for (uint32_t g = 0; g < G; g++) {
        EpRep *r = &cl->reps[g];
        int row = rows ? rows[g] : r->id;
        uint32_t ctl = order ? order[g] : CTL_IDENTITY;
        uint8_t before = 0;
        if (held(r, row, col[g])) before = at(r, row, col[g])->status;
        for (int oi = 0; oi < R; oi++) {
            int p = (int)ctl_order(ctl, oi);
            if (p == r->id || p >= R) continue;
            size_t o = (size_t)p * G + g;
            if (!(flags[o] & 1)) continue;
            DepSet d = dep_empty();
            for (int i = 0; i < R; i++) d.c[i] = deps[((size_t)p * R + i) * G + g];
            handle_msg_pre_accept_reply(r, (uint8_t)p, row, col[g], ballot[o], seq[o], &d, exploded ? exploded[g] : 0); drain_exec(r);
        }
        decision[g] = 0; d_seq[g] = 0;
        for (int i = 0; i < R; i++) d_deps[(size_t)i * G + g] = NONE;
        if (held(r, row, col[g])) {
            Inst *in = at(r, row, col[g]);
            if (before == ST_PREACCEPTING && in->status != ST_PREACCEPTING) {
                decision[g] = in->status >= ST_COMMITTED ? ST_COMMITTED : ST_ACCEPTING;
                if (in->status == ST_ACCEPTING || in->status >= ST_COMMITTED) {
                    d_seq[g] = in->seq;
                    for (int i = 0; i < R; i++) d_deps[(size_t)i * G + g] = in->deps.c[i];
                }
                /* an instance that went Accepting and on to Committed by my own ack alone (simple_q == 1) would
                 * read Committed here; with R >= 3 the slow path needs peers */
                if (in->status == ST_ACCEPTING) decision[g] = ST_ACCEPTING;
            }
        }
    }
}

/* messages.rs:273-345 + the acceptor's AcceptSlot completion (durability.rs:84-100) */
void orc_ep_handle_accept(void *h, const uint8_t *flags, const uint8_t *peer, const uint32_t *col, const uint64_t *ballot,
                          const uint64_t *seq, const uint32_t *deps, const uint8_t *key, uint8_t *r_flags,
                          uint64_t *r_ballot, const uint8_t *rows) {
    EpCl *cl = (EpCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        EpRep *r = &cl->reps[g];
        r_flags[g] = 0; r_ballot[g] = 0;
        if (!(flags[g] & 1)) continue;
        int row = rows ? rows[g] : peer[g];
        uint32_t c = col[g];
        if (c < r->start_col || (c < r->start_col + r->len[row] && !held(r, row, c))) continue;
        while (r->start_col + r->len[row] <= c) row_push(r, row, null_instance());
        Inst *in = at(r, row, c);
        if (ballot[g] >= in->bal) {
            in->bal = ballot[g]; in->status = ST_ACCEPTING; in->seq = seq[g]; in->key = key[g];
            for (int i = 0; i < MAXR; i++) in->deps.c[i] = i < cl->R ? deps[(size_t)i * G + g] : NONE;
            refresh_highest_cols(r, row, c, key[g]);
            in->has_rbk = 1; in->source = peer[g];
            if (in->has_lbk) { handle_msg_accept_reply(r, r->id, row, c, in->bal); drain_exec(r); }
            else { r_flags[g] = 1; r_ballot[g] = in->bal; }
        }
    }
}

/* AcceptReplies to my instance (me, col[g]): ballot[R][G], flags[R][G]; committed[g] = 1 if it commits here */
void orc_ep_handle_accept_replies(void *h, const uint32_t *col, const uint64_t *ballot, const uint8_t *flags,
                                  const uint32_t *order, uint8_t *committed, const uint8_t *rows) {
    EpCl *cl = (EpCl *)h;
    const uint32_t G = cl->G; const int R = cl->R;
    for (uint32_t g = 0; g < G; g++) {
        EpRep *r = &cl->reps[g];
        int row = rows ? rows[g] : r->id;
        uint32_t ctl = order ? order[g] : CTL_IDENTITY;
        uint8_t before = 0;
        if (held(r, row, col[g])) before = at(r, row, col[g])->status;
        for (int oi = 0; oi < R; oi++) {
            int p = (int)ctl_order(ctl, oi);
            if (p == r->id || p >= R) continue;
            size_t o = (size_t)p * G + g;
            if (!(flags[o] & 1)) continue;
            handle_msg_accept_reply(r, (uint8_t)p, row, col[g], ballot[o]); drain_exec(r);
        }
        committed[g] = 0;
        if (held(r, row, col[g]))
            committed[g] = (before == ST_ACCEPTING && at(r, row, col[g])->status >= ST_COMMITTED) ? 1 : 0;
    }
}

/* messages.rs:438-508 + handle_logged_commit_slot */
void orc_ep_handle_commit_notice(void *h, const uint8_t *flags, const uint8_t *peer, const uint32_t *col,
                                 const uint64_t *ballot, const uint64_t *seq, const uint32_t *deps, const uint8_t *key,
                                 const uint8_t *rows) {
    EpCl *cl = (EpCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        EpRep *r = &cl->reps[g];
        if (!(flags[g] & 1)) continue;
        int row = rows ? rows[g] : peer[g];
        uint32_t c = col[g];
        if (c < r->start_col || (c < r->start_col + r->len[row] && !held(r, row, c))) continue;
        while (r->start_col + r->len[row] <= c) row_push(r, row, null_instance());   /* :462-465 */
        Inst *in = at(r, row, c);
        if (ballot[g] >= in->bal) {                                  /* :469 */
            in->bal = ballot[g]; in->status = ST_COMMITTED; in->seq = seq[g]; in->key = key[g];
            for (int i = 0; i < MAXR; i++) in->deps.c[i] = i < cl->R ? deps[(size_t)i * G + g] : NONE;
            refresh_highest_cols(r, row, c, key[g]);
            handle_logged_commit_slot(r, row, c); drain_exec(r);
        }
    }
}

/* ---- explicit prepare ---------------------------------------------------------------------------------------------- */
static uint64_t make_greater_ballot(uint8_t id, uint64_t bal) { return (((bal >> 8) + 1) << 8) | (uint64_t)(id + 1); }   /* mod.rs:500-508 */

/* dependency.rs:249-327: 0 = cannot decide yet, else the Status of the next phase with the instance state to feed it */
static int exp_prepare_next_step(EpRep *r, int slot_row, Inst *in, uint64_t *seq, DepSet *deps, uint8_t *key) {
    if (__builtin_popcount(in->xp_acks) < r->simple_q) return 0;     /* :257-260 */
    int has_commit = -1, has_accept = -1, has_pre_accept = -1;       /* :264-273, peers in id order (see the header) */
    for (int p = 0; p < r->population; p++) {
        if (!in->xp_has[p]) continue;
        if (in->xp_status[p] == ST_COMMITTED) has_commit = p;
        else if (in->xp_status[p] == ST_ACCEPTING) has_accept = p;
        else if (in->xp_status[p] == ST_PREACCEPTING) has_pre_accept = p;
    }
    if (has_commit >= 0) { *seq = in->xp_seq[has_commit]; *deps = in->xp_deps[has_commit]; *key = in->xp_key[has_commit]; return ST_COMMITTED; }
    if (has_accept >= 0) { *seq = in->xp_seq[has_accept]; *deps = in->xp_deps[has_accept]; *key = in->xp_key[has_accept]; return ST_ACCEPTING; }
    /* :286-311 at least N/2 identical PreAccepting replies under the row's default ballot, none from the row's owner */
    if (in->xp_max_bal == (uint64_t)(slot_row + 1)) {
        int idx[MAXR], n = 0;
        for (int p = 0; p < r->population; p++)
            if (in->xp_has[p] && p != slot_row && in->xp_status[p] == ST_PREACCEPTING) idx[n++] = p;
        if (n >= r->simple_q - 1 && n > 0) {                         /* :302-306 (get_enough_identical's thresh is simple_quorum_cnt itself) */
            uint8_t visited[MAXR] = {0};
            int first = 0;
            visited[0] = 1;
            while (first < n) {                                      /* :333-367 */
                int next_first = n, same = 1;
                for (int i = first + 1; i < n; i++) {
                    if (visited[i]) continue;
                    if (in->xp_seq[idx[i]] == in->xp_seq[idx[first]] && in->xp_key[idx[i]] == in->xp_key[idx[first]] &&
                        dep_eq(&in->xp_deps[idx[i]], &in->xp_deps[idx[first]], r->population)) { visited[i] = 1; same++; }
                    else if (next_first == n) next_first = i;
                }
                if (same >= r->simple_q) {
                    *seq = in->xp_seq[idx[first]]; *deps = in->xp_deps[idx[first]]; *key = in->xp_key[idx[first]];
                    return ST_ACCEPTING;                             /* :313-315 */
                }
                first = next_first;
            }
        }
    }
    if (has_pre_accept >= 0) {                                       /* :316-319 */
        *seq = in->xp_seq[has_pre_accept]; *deps = in->xp_deps[has_pre_accept]; *key = in->xp_key[has_pre_accept];
        return ST_PREACCEPTING;
    }
    *seq = 1; *deps = dep_empty(); *key = NO_KEY;                    /* :320-327 no-op */
    return ST_PREACCEPTING;
}

/* messages.rs:577-821 with the WAL completions of what it logs (durability.rs:25-35, 78-83, 104-); returns the Status of
 * the message it broadcasts (CommitNotice / Accept / PreAccept for (row, col) under new_ballot with inst's seq / deps /
 * reqs) or 0 */
static int handle_msg_exp_prepare_reply(EpRep *r, uint8_t peer, int row, uint32_t col, uint64_t new_ballot, uint64_t voted_bal,
                                        uint8_t voted_status, uint64_t voted_seq, const DepSet *voted_deps, uint8_t voted_key) {
    if (col < r->start_col) return 0;
    if (col >= r->start_col + r->len[row] || !held(r, row, col)) return 0;   /* :599-601 */
    Inst *in = at(r, row, col);
    if (new_ballot <= in->bal || !in->has_lbk) return 0;             /* :603-605 */
    if ((in->xp_acks >> peer) & 1) return 0;                         /* :607-609 */
    if (voted_bal > in->xp_max_bal) {                                /* :612-615 */
        for (int p = 0; p < MAXR; p++) in->xp_has[p] = 0;
        in->xp_max_bal = voted_bal;
    }
    if (voted_bal >= in->xp_max_bal) {                               /* :616-621 */
        in->xp_has[peer] = 1; in->xp_status[peer] = voted_status; in->xp_seq[peer] = voted_seq; in->xp_deps[peer] = *voted_deps;
        in->xp_key[peer] = voted_key;
    }
    in->xp_acks |= (uint8_t)(1u << peer);
    uint64_t seq; DepSet deps; uint8_t key;
    int next = exp_prepare_next_step(r, row, in, &seq, &deps, &key);
    if (!next) return 0;
    in->bal = new_ballot; in->status = (uint8_t)next; in->seq = seq; in->deps = deps; in->key = key;
    refresh_highest_cols(r, row, col, key);
    if (next == ST_COMMITTED) {                                      /* :637-690 */
        r->n_xp_commit++;
        handle_logged_commit_slot(r, row, col);
    } else if (next == ST_ACCEPTING) {                               /* :692-745 */
        r->n_xp_accept++;
        handle_msg_accept_reply(r, r->id, row, col, in->bal);
    } else {                                                         /* :747-815 */
        in->avoid_fast_path = 1;
        if (key == NO_KEY) r->n_xp_noop++; else r->n_xp_pre_accept++;
        handle_msg_pre_accept_reply(r, r->id, row, col, in->bal, in->seq, &in->deps, 0);
    }
    return next;
}

/* heartbeat.rs:17-125 for HearTimeout { peer = src[g] } (NO_REP: none in this group).  Out: the ExpPrepare { slot, new_ballot }
 * broadcasts of the call, in column order: n[g], col / bal [W][G]. */
void orc_ep_heartbeat_timeout(void *h, const uint8_t *src, const uint8_t *exploded, uint32_t *out_n, uint32_t *out_col,
                              uint64_t *out_bal) {
    EpCl *cl = (EpCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        EpRep *r = &cl->reps[g];
        out_n[g] = 0;
        if (src[g] == NO_REP || src[g] >= r->population || src[g] == r->id) continue;
        const uint8_t ts = src[g];
        /* :35-60 the instances I lead that sit in PreAccept phase: "reply" with ballot 0 */
        SlotVec mk = {0, 0, 0};
        for (int row = 0; row < r->population; row++)
            for (uint32_t c = r->commit_bars[row]; c < r->start_col + r->len[row]; c++) {
                if (!held(r, row, c)) continue;
                Inst *in = at(r, row, c);
                if (in->status == ST_PREACCEPTING && in->has_lbk) sv_push(&mk, (Slot){(uint8_t)row, c});
            }
        DepSet none = dep_empty();
        for (uint32_t i = 0; i < mk.n; i++) {
            handle_msg_pre_accept_reply(r, ts, mk.v[i].row, mk.v[i].col, 0, 0, &none, exploded ? exploded[g] : 0); drain_exec(r);
        }
        free(mk.v);
        /* :62-107 ExpPrepare for every in-progress instance of that peer's row */
        const int row = ts;
        uint32_t n = 0;
        for (uint32_t c = r->exec_bars[row]; c < r->start_col + r->len[row]; c++) {
            if (!held(r, row, c)) continue;
            Inst *in = at(r, row, c);
            if (in->status >= ST_EXECUTING || (in->has_rbk && in->source != ts)) continue;   /* :73-80 */
            if (in->status == ST_COMMITTED) continue;                /* :82-84 (`external` is not modelled) */
            const uint64_t nb = make_greater_ballot(r->id, in->bal);
            fresh_leader_bk(in);
            if (n < r->W) { out_col[(size_t)n * G + g] = c; out_bal[(size_t)n * G + g] = nb; }
            n++;
        }
        out_n[g] = n;
        /* :110-123 my own ExpPrepareReplies */
        for (uint32_t i = 0; i < n && i < r->W; i++) {
            const uint32_t c = out_col[(size_t)i * G + g];
            Inst *in = at(r, row, c);
            DepSet d = in->deps;
            handle_msg_exp_prepare_reply(r, r->id, row, c, out_bal[(size_t)i * G + g], in->bal, in->status, in->seq, &d, in->key);
            drain_exec(r);
        }
    }
}

/* messages.rs:511-574: one ExpPrepare { slot = (row, col), new_ballot } from `peer` per group; the ExpPrepareReply back */
void orc_ep_handle_exp_prepare(void *h, const uint8_t *flags, const uint8_t *peer, const uint8_t *rows, const uint32_t *col,
                               const uint64_t *new_ballot, uint8_t *r_flags, uint64_t *r_voted_bal, uint8_t *r_status,
                               uint64_t *r_seq, uint32_t *r_deps, uint8_t *r_key) {
    EpCl *cl = (EpCl *)h;
    const uint32_t G = cl->G;
    for (uint32_t g = 0; g < G; g++) {
        EpRep *r = &cl->reps[g];
        r_flags[g] = 0; r_voted_bal[g] = 0; r_status[g] = 0; r_seq[g] = 0; r_key[g] = NO_KEY;
        for (int i = 0; i < cl->R; i++) r_deps[(size_t)i * G + g] = NONE;
        if (!(flags[g] & 1)) continue;
        int row = rows[g];
        uint32_t c = col[g];
        if (c < r->start_col || (c < r->start_col + r->len[row] && !held(r, row, c))) continue;
        while (r->start_col + r->len[row] <= c) row_push(r, row, null_instance());   /* :530-533 */
        Inst *in = at(r, row, c);
        if (new_ballot[g] > in->bal) {                               /* :537 */
            in->has_rbk = 1; in->source = peer[g];
            r_flags[g] = 1; r_voted_bal[g] = in->bal; r_status[g] = in->status; r_seq[g] = in->seq; r_key[g] = in->key;
            for (int i = 0; i < cl->R; i++) r_deps[(size_t)i * G + g] = in->deps.c[i];
        }
    }
}

/* The ExpPrepareReplies to the instance (rows[g], col[g]) I am preparing: per peer [R][G] new_ballot, voted_bal, voted_status,
 * voted_seq, voted_key, voted_deps [R][R][G]; flags[R][G] bit0 = present; peers in `order` order.  Out: decision[g] = the
 * Status of the message broadcast here (0: none) with its ballot / seq / deps / key. */
void orc_ep_handle_exp_prepare_replies(void *h, const uint8_t *rows, const uint32_t *col, const uint64_t *new_ballot,
                                       const uint64_t *voted_bal, const uint8_t *voted_status, const uint64_t *voted_seq,
                                       const uint32_t *voted_deps, const uint8_t *voted_key, const uint8_t *flags,
                                       const uint32_t *order, uint8_t *decision, uint64_t *d_ballot, uint64_t *d_seq,
                                       uint32_t *d_deps, uint8_t *d_key) {
    EpCl *cl = (EpCl *)h;
    const uint32_t G = cl->G; const int R = cl->R;
    for (uint32_t g = 0; g < G; g++) {
        EpRep *r = &cl->reps[g];
        const int row = rows[g];
        uint32_t ctl = order ? order[g] : CTL_IDENTITY;
        decision[g] = 0; d_ballot[g] = 0; d_seq[g] = 0; d_key[g] = NO_KEY;
        for (int i = 0; i < R; i++) d_deps[(size_t)i * G + g] = NONE;
        for (int oi = 0; oi < R; oi++) {
            int p = (int)ctl_order(ctl, oi);
            if (p == r->id || p >= R) continue;
            size_t o = (size_t)p * G + g;
            if (!(flags[o] & 1)) continue;
            DepSet d = dep_empty();
            for (int i = 0; i < R; i++) d.c[i] = voted_deps[((size_t)p * R + i) * G + g];
            int next = handle_msg_exp_prepare_reply(r, (uint8_t)p, row, col[g], new_ballot[o], voted_bal[o], voted_status[o],
                                                    voted_seq[o], &d, voted_key[o]);
            drain_exec(r);
            if (next) {
                Inst *in = at(r, row, col[g]);
                decision[g] = (uint8_t)next; d_ballot[g] = new_ballot[o]; d_seq[g] = in->seq; d_key[g] = in->key;
                for (int i = 0; i < R; i++) d_deps[(size_t)i * G + g] = in->deps.c[i];
            }
        }
    }
}

/* explicit-prepare bookkeeping, [R][W][G] by col % W like orc_ep_dump: acks, max_bal, avoid_fast_path, the peers with a
 * voted entry (bitmap) and those entries [R][W][R][G] (+ deps [R][W][R][R][G]); counters[4] = decisions Committed, Accepting,
 * PreAccepting with a command, PreAccepting as a no-op */
void orc_ep_xp_dump(void *h, uint8_t *acks, uint64_t *max_bal, uint8_t *avoid, uint8_t *has, uint8_t *vstatus, uint64_t *vseq,
                    uint8_t *vkey, uint32_t *vdeps, uint64_t *counters) {
    EpCl *cl = (EpCl *)h;
    const uint32_t G = cl->G; const int R = cl->R;
    for (int k = 0; k < 4; k++) counters[k] = 0;
    for (uint32_t g = 0; g < G; g++) {
        EpRep *r = &cl->reps[g];
        const uint32_t W = r->W;
        counters[0] += r->n_xp_commit; counters[1] += r->n_xp_accept; counters[2] += r->n_xp_pre_accept; counters[3] += r->n_xp_noop;
        for (int row = 0; row < R; row++) {
            for (uint32_t w = 0; w < W; w++) {
                size_t o = ((size_t)row * W + w) * G + g;
                acks[o] = 0; max_bal[o] = 0; avoid[o] = 0; has[o] = 0;
                for (int p = 0; p < R; p++) {
                    size_t q = (((size_t)row * W + w) * R + p) * G + g;
                    vstatus[q] = 0; vseq[q] = 0; vkey[q] = NO_KEY;
                    for (int i = 0; i < R; i++) vdeps[((((size_t)row * W + w) * R + p) * R + i) * G + g] = NONE;
                }
            }
            uint32_t end = r->start_col + r->len[row], lo = end > W ? end - W : r->start_col;
            for (uint32_t c = lo; c < end; c++) {
                Inst *in = at(r, row, c);
                size_t o = ((size_t)row * W + (c % W)) * G + g;
                avoid[o] = in->avoid_fast_path;
                if (!in->has_lbk) continue;
                acks[o] = in->xp_acks; max_bal[o] = in->xp_max_bal;
                for (int p = 0; p < R; p++) {
                    if (!in->xp_has[p]) continue;
                    has[o] |= (uint8_t)(1u << p);
                    size_t q = (((size_t)row * W + (c % W)) * R + p) * G + g;
                    vstatus[q] = in->xp_status[p]; vseq[q] = in->xp_seq[p]; vkey[q] = in->xp_key[p];
                    for (int i = 0; i < R; i++) vdeps[((((size_t)row * W + (c % W)) * R + p) * R + i) * G + g] = in->xp_deps[p].c[i];
                }
            }
        }
    }
}

/* canonical dump: rows x the last W columns (by col % W), bars, highest_cols, counters */
void orc_ep_dump(void *h, uint32_t *len, uint32_t *commit_bars, uint64_t *bal, uint64_t *seq, uint8_t *status,
                 uint8_t *key, uint32_t *deps, uint8_t *pa_acks, uint8_t *acc_acks, uint8_t *bk, uint32_t *highest_cols,
                 uint64_t *counters) {
    EpCl *cl = (EpCl *)h;
    const uint32_t G = cl->G; const int R = cl->R;
    counters[0] = counters[1] = counters[2] = 0;
    for (uint32_t g = 0; g < G; g++) {
        EpRep *r = &cl->reps[g];
        const uint32_t W = r->W;
        counters[0] += r->n_fast; counters[1] += r->n_slow; counters[2] += r->n_accept_commits;
        for (int row = 0; row < R; row++) {
            len[(size_t)row * G + g] = r->start_col + r->len[row];
            commit_bars[(size_t)row * G + g] = r->commit_bars[row];
            for (uint32_t w = 0; w < W; w++) {
                size_t o = ((size_t)row * W + w) * G + g;
                bal[o] = 0; seq[o] = 0; status[o] = 0; key[o] = NO_KEY; pa_acks[o] = 0; acc_acks[o] = 0; bk[o] = 0;
                for (int i = 0; i < R; i++) deps[(o * R) + i] = NONE;
            }
            uint32_t end = r->start_col + r->len[row], lo = end > W ? end - W : r->start_col;
            for (uint32_t c = lo; c < end; c++) {
                Inst *in = at(r, row, c);
                size_t o = ((size_t)row * W + (c % W)) * G + g;
                bal[o] = in->bal; seq[o] = in->seq; status[o] = in->status; key[o] = in->key;
                pa_acks[o] = in->pa_acks; acc_acks[o] = in->acc_acks;
                bk[o] = (uint8_t)(in->has_lbk | (in->has_rbk << 1) | ((in->has_rbk ? in->source : 0) << 2));
                for (int i = 0; i < R; i++) deps[(o * R) + i] = in->deps.c[i];
            }
        }
        for (uint32_t k = 0; k < r->n_keys; k++)
            for (int i = 0; i < R; i++)
                highest_cols[((size_t)k * R + i) * G + g] = r->hc_present[k] ? r->highest_cols[k].c[i] : NONE;
    }
}

/* execution state: exec_bars[R][G], kv[n_keys][G], digest[G], counters[6] = commands submitted, of them
 * re-submissions of an already executing instance, pops of an instance no longer held, components with
 * more than one node, attempts, aborted attempts */
void orc_ep_exec_dump(void *h, uint32_t *exec_bars, uint64_t *kv, uint64_t *digest, uint64_t *counters) {
    EpCl *cl = (EpCl *)h;
    const uint32_t G = cl->G;
    for (int k = 0; k < 6; k++) counters[k] = 0;
    for (uint32_t g = 0; g < G; g++) {
        EpRep *r = &cl->reps[g];
        for (int row = 0; row < cl->R; row++) exec_bars[(size_t)row * G + g] = r->exec_bars[row];
        for (uint32_t k = 0; k < r->n_keys; k++) kv[(size_t)k * G + g] = r->kv[k];
        digest[g] = r->digest;
        counters[0] += r->n_exec; counters[1] += r->n_reexec; counters[2] += r->n_unheld;
        counters[3] += r->n_multi_scc; counters[4] += r->n_attempts; counters[5] += r->n_aborts;
    }
}

/* the commands submitted since the last call, group-major, in submission order within a group: (group, row, col);
 * returns how many there were (only the first cap are written) */
uint64_t orc_ep_take_submissions(void *h, uint32_t *group, uint8_t *row, uint32_t *col, uint64_t cap) {
    EpCl *cl = (EpCl *)h;
    uint64_t n = 0;
    for (uint32_t g = 0; g < cl->G; g++) {
        EpRep *r = &cl->reps[g];
        for (uint32_t i = 0; i < r->n_sublog; i++, n++)
            if (n < cap) { group[n] = g; row[n] = r->sublog[i].row; col[n] = r->sublog[i].col; }
        r->n_sublog = 0;
    }
    return n;
}
