"""RSPaxos handlers of the CPU oracle against hand-derived traces of the reference code
(src/protocols/rspaxos/{request,messages,durability,leadership,execution}.rs), R = 5: majority 3, RS(3, 2),
data shards 0-2 (mask 0b00111), all shards 0b11111."""
import numpy as np

NULL, NO_REP = 0xFFFFFFFF, 0xFF
NUL, PREPARING, ACCEPTING, COMMITTED, EXECUTED = 0, 1, 2, 3, 4
B0 = (1 << 8) | 1                                               # make_unique_ballot(1) of replica 0
u8, u32, u64 = (lambda *v: np.array(v, np.uint8)), (lambda *v: np.array(v, np.uint32)), (lambda *v: np.array(v, np.uint64))


def _oracle_factory(oracle):
    def make(me, ft=0, W=8):
        o = oracle.RspOracle(1, 5, me=me, W=W, fault_tolerance=ft)
        o.preset_leader(0)
        return o
    return make


TRACES = ("trace_leader_append_tally_and_execution", "trace_follower_holds_one_shard_and_cannot_execute",
          "trace_step_up_prepare_merge_and_reconstruct", "trace_all_replies_without_enough_shards_choose_the_empty_batch",
          "trace_follower_answers_prepare_and_reconstruct")


def test_traces(oracle):
    for name in TRACES:
        globals()[name](_oracle_factory(oracle))


def _slot(o, s):
    d = o.dump()
    w = s % o.W
    return {k[2:]: int(d[k][w, 0]) for k in d if k.startswith("s_")}


def _acks(o, slot, peers, ballot=B0):
    b = np.zeros((5, 1), np.uint64); f = np.zeros((5, 1), np.uint8)
    for p in peers:
        b[p] = ballot; f[p] = 1
    return int(o.accept_replies(u32(slot), b, f)["committed"][0])


def trace_leader_append_tally_and_execution(_new):
    for ft, need in ((0, 2), (1, 3)):                            # peers needed besides my own ack: majority + ft - 1
        o = _new(0, ft)
        a = o.req_batch(u32(77))
        assert (int(a["a_n"][0]), int(a["a_slot"][0, 0]), int(a["a_val"][0, 0]), int(a["a_ballot"][0])) == (1, 0, 77, B0)
        s = _slot(o, 0)
        # request.rs:71-107: the leader holds all five shards, votes for its own one, and its AcceptData completion acks
        assert (s["status"], s["val"], s["mask"], s["vbal"], s["vmask"], s["aacks"]) == (ACCEPTING, 77, 0b11111, B0, 0b00001, 0b00001)
        assert _acks(o, 0, list(range(1, need))) == 0            # one short of majority + fault_tolerance (messages.rs:437-440)
        assert _acks(o, 0, [need]) == 1
        d = o.dump()
        # durability.rs:140-181: enough shards -> submitted, executed (rule 0), both bars move
        assert (int(d["commit_bar"][0]), int(d["exec_bar"][0]), _slot(o, 0)["status"]) == (1, 1, EXECUTED)
        assert int(d["digest"][0]) == ((0 ^ ((0 << 32) | 77)) * 0x100000001B3) & ((1 << 64) - 1)
        assert _acks(o, 0, [4]) == 0                             # late reply: not Accepting any more


def trace_follower_holds_one_shard_and_cannot_execute(_new):
    o = _new(2)
    r = o.accept(u8(1), u8(0), u32(0), u64(B0), u32(77), u8(1 << 2))
    assert (int(r["r_ballot"][0]), int(r["r_slot"][0])) == (B0, 0)
    s = _slot(o, 0)
    assert (s["status"], s["mask"], s["vbal"], s["vmask"], s["rsrc"], s["flags"] & 2) == (ACCEPTING, 0b00100, B0, 0b00100, 0, 2)
    # the leader's heartbeat says commit_bar = 1 (leadership.rs:281-322): Committed, but one shard < majority:
    # the commit-bar run stops in front of it (durability.rs:146-154)
    h = o.heartbeat(u8(1), u8(0), u64(B0), u32(1), u32(0), u32(0))
    assert int(h["reply"][0]) == 1 and int(h["commit_bar"][0]) == 0   # my Heartbeat goes back before the learning
    d = o.dump()
    assert (_slot(o, 0)["status"], int(d["commit_bar"][0]), int(d["exec_bar"][0])) == (COMMITTED, 0, 0)
    # an Accept with a stale ballot is ignored (messages.rs:360)
    assert int(o.accept(u8(1), u8(1), u32(1), u64(2), u32(5), u8(4))["r_ballot"][0]) == 0


def trace_step_up_prepare_merge_and_reconstruct(_new):
    """Replica 1 takes over.  Its log: slot 0 Committed with only its own shard (heartbeat), slot 1 Accepting with
    its own shard.  become_a_leader: Reconstruct for slot 0, Prepare from slot 1."""
    o = _new(1)
    o.accept(u8(1), u8(0), u32(0), u64(B0), u32(70), u8(1 << 1))
    o.accept(u8(1), u8(0), u32(1), u64(B0), u32(71), u8(1 << 1))
    o.heartbeat(u8(1), u8(0), u64(B0), u32(1), u32(0), u32(0))
    bl = o.become_leader(u8(0))
    b1 = (2 << 8) | 2                                            # make_greater_ballot(0x101) by replica 1
    assert (int(bl["hb_flags"][0]), int(bl["hb_ballot"][0])) == (1, B0)          # the step-up Heartbeat still carries the old ballot
    assert (int(bl["p_flags"][0]), int(bl["p_trig"][0]), int(bl["p_ballot"][0])) == (1, 1, b1)
    assert (int(bl["rc_n"][0]), int(bl["rc_slot"][0, 0])) == (1, 0)              # leadership.rs:142-148
    s1 = _slot(o, 1)
    # my own PrepareBal completion is a Prepare reply from myself (durability.rs:27-46): voted (B0, shard 1) taken
    assert (s1["status"], s1["bal"], s1["ltrig"], s1["lendp"], s1["packs"], s1["pmax"], s1["mask"]) == (PREPARING, b1, 1, 1, 0b00010, B0, 0b00010)
    d = o.dump()
    assert (int(d["leader"][0]), int(d["bal_prepared"][0]), int(d["bal_prep_sent"][0])) == (1, 0, b1)

    def reply(peer, vbal, vval, vmask):
        W = o.W
        vb = np.zeros((W, 1), np.uint64); vv = np.full((W, 1), NULL, np.uint32); vm = np.zeros((W, 1), np.uint8)
        vb[0] = vbal; vv[0] = vval; vm[0] = vmask
        return o.prepare_replies(u8(peer), u32(1), u32(1), u32(1), u64(b1), vb, vv, vm)

    a = reply(2, B0, 71, 1 << 2)                                 # same voted ballot: shards merge (messages.rs:189-194)
    assert _slot(o, 1)["mask"] == 0b00110 and int(a["a_n"][0]) == 0          # 2 acks < majority
    a = reply(3, 0, NULL, 0)                                     # never voted: third ack = quorum, but only 2 shards
    # quorum 3 < population - fault_tolerance = 5: "not yet for this instance" (messages.rs:240-252)
    assert int(a["a_n"][0]) == 0 and _slot(o, 1)["status"] == PREPARING and int(o.dump()["bal_prepared"][0]) == b1
    a = reply(4, B0, 71, 1 << 4)                                 # a third shard: reconstruct, compute parity, re-Accept
    assert (int(a["a_n"][0]), int(a["a_slot"][0, 0]), int(a["a_val"][0, 0]), int(a["a_ballot"][0])) == (1, 1, 71, b1)
    s1 = _slot(o, 1)
    assert (s1["status"], s1["mask"], s1["vbal"], s1["vmask"], s1["aacks"]) == (ACCEPTING, 0b11111, b1, 0b00010, 0b00010)
    # reconstruction read for slot 0: two more shards arrive -> the run reconstructs and executes it
    W = o.W
    for peer in (2, 3):
        rs = np.zeros((W, 1), np.uint32); rb = np.zeros((W, 1), np.uint64); rv = np.full((W, 1), NULL, np.uint32); rm = np.zeros((W, 1), np.uint8)
        rb[0] = B0; rv[0] = 70; rm[0] = 1 << peer
        o.reconstruct_reply(u8(1), u32(1), rs, rb, rv, rm)
    d = o.dump()
    s0 = _slot(o, 0)
    assert (s0["status"], s0["mask"], int(d["commit_bar"][0]), int(d["exec_bar"][0])) == (EXECUTED, 0b01111, 1, 1)


def trace_all_replies_without_enough_shards_choose_the_empty_batch(_new):
    o = _new(1)
    o.accept(u8(1), u8(0), u32(0), u64(B0), u32(70), u8(1 << 1))
    bl = o.become_leader(u8(0))
    b1 = int(bl["p_ballot"][0])
    W = o.W
    nv = (np.zeros((W, 1), np.uint64), np.full((W, 1), NULL, np.uint32), np.zeros((W, 1), np.uint8))
    for peer in (0, 2, 3):
        a = o.prepare_replies(u8(peer), u32(1), u32(0), u32(0), u64(b1), *nv)
        assert int(a["a_n"][0]) == 0                             # 4 acks < 5 = population - fault_tolerance
    a = o.prepare_replies(u8(4), u32(1), u32(0), u32(0), u64(b1), *nv)
    # messages.rs:240-252: everybody answered and one shard is all there is: any value may be chosen: the empty batch
    assert (int(a["a_n"][0]), int(a["a_slot"][0, 0]), int(a["a_val"][0, 0])) == (1, 0, 0)
    assert _slot(o, 0)["mask"] == 0b11111


def trace_follower_answers_prepare_and_reconstruct(_new):
    o = _new(3)
    o.accept(u8(1), u8(0), u32(0), u64(B0), u32(70), u8(1 << 3))
    o.accept(u8(1), u8(0), u32(2), u64(B0), u32(72), u8(1 << 3))                 # slot 1 stays a null hole
    b1 = (2 << 8) | 2
    pr = o.prepare(u8(1), u8(1), u32(1), u64(b1))
    # messages.rs:40-48: endprep = last non-null slot = 2; rows for slots 1, 2: slot 1 never voted, slot 2 voted (B0, shard 3)
    assert [int(pr[k][0]) for k in ("pr_n", "pr_trig", "pr_endp", "pr_ballot")] == [2, 1, 2, b1]
    assert [int(x) for x in pr["pr_vbal"][:2, 0]] == [0, B0] and int(pr["pr_vval"][1, 0]) == 72 and int(pr["pr_vmask"][1, 0]) == 1 << 3
    d = o.dump()
    assert int(d["leader"][0]) == 1 and int(d["bal_max_seen"][0]) == b1 and _slot(o, 0)["status"] == ACCEPTING and _slot(o, 2)["status"] == PREPARING
    rc_slot = np.zeros((o.W, 1), np.uint32); rc_slot[1] = 1; rc_slot[2] = 5
    rr = o.reconstruct(u8(1), u32(3), rc_slot)                   # slots 0, 1, 5: only slot 0 has something to send (messages.rs:494-499)
    assert int(rr["rr_n"][0]) == 1 and (int(rr["rr_slot"][0, 0]), int(rr["rr_bal"][0, 0]), int(rr["rr_val"][0, 0]), int(rr["rr_mask"][0, 0])) == (0, B0, 70, 1 << 3)
    assert int(o.dump()["len"][0]) == 6                          # padded with nulls up to slot 5 (:489-491)
