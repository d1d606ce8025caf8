"""RSPaxos end to end with real bytes: the RSPaxos replica engine decides WHICH shards exist where (token + shard
mask per instance), the RS kernels hold the BYTES (RSCodewordBatch: from_data / compute_parity / subset_copy /
absorb_other / reconstruct_data, rscoding.rs:165-537).  The test carries a codeword batch next to every engine
message and applies to it exactly what the engine did to the mask; wherever the engine says an instance is
executable the bytes must be the batch the old leader encoded.

Steady state under leader 0 (one shard per follower), then replica 1 takes over: instances still open are rebuilt
from the PrepareReplies' voted shards (any 3 of 5: reconstruct + re-encode), instances the old leader had committed
from reconstruction reads.  Sorts last like the other first-run device tests."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
NULL, NO_REP = 0xFFFFFFFF, 0xFF


def _payload(tokens, L):
    """the serialized request batch behind a token: L bytes, a function of the token alone"""
    t = tokens.astype(np.uint64)[:, None]
    i = np.arange(L, dtype=np.uint64)[None, :]
    return (((t * np.uint64(2654435761) + i * np.uint64(40503)) >> np.uint64(7)) & np.uint64(0xFF)).astype(np.uint8)


def test_tokens_are_real_shard_bytes(cuda, oracle):
    import torch
    import rsp_cluster as rc
    from summerset_amd import RSCodewordBatch, RSPaxosReplicaGroup
    G, R, W, L, T = 48, 5, 16, 1000, 6
    u8 = lambda v: np.full(G, v, np.uint8)
    reps = [rc.NumpyEngine(RSPaxosReplicaGroup(G, R, me=r, window=W, fault_tolerance=1), cuda) for r in range(R)]
    for r in reps:
        r.preset_leader(0)
    held = [dict() for _ in range(R)]                            # held[r][slot] = the codeword batch replica r holds
    tokens = {}

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)

    # ---- steady state: leader 0 encodes, every follower gets its own shard -----------------------
    for t in range(T):
        tok = (1 + t * G + np.arange(G)).astype(np.uint32)
        a = reps[0].req_batch(tok)
        assert (a["a_n"] == 1).all() and (a["a_slot"][0] == t).all()
        cw = RSCodewordBatch.from_data(dev(_payload(tok, L)), 3, 2)
        cw.compute_parity()
        held[0][t], tokens[t] = cw, tok
        ballot = np.zeros((R, G), np.uint64); flags = np.zeros((R, G), np.uint8)
        # the last two instances reach only follower 1 and 2: still open when the leader goes away
        for q in range(1, R if t < T - 2 else 3):
            ar = reps[q].accept(u8(1), u8(0), a["a_slot"][0], a["a_ballot"], a["a_val"][0], u8(1 << q))
            held[q][t] = cw.subset_copy(1 << q)                  # what the Accept carried
            flags[q] = 1; ballot[q] = ar["r_ballot"]
        c = reps[0].accept_replies(a["a_slot"][0], ballot, flags)
        assert (c["committed"] == (1 if t < T - 2 else 0)).all()  # majority + fault_tolerance = 4 acks
    d0 = reps[0].dump()
    assert (d0["commit_bar"] == T - 2).all() and (d0["exec_bar"] == T - 2).all()
    hb = reps[0].bcast_heartbeat(u8(1))                          # followers learn the commits, cannot run them (one shard)
    for q in range(1, R):
        reps[q].heartbeat(u8(1), u8(0), hb["ballot"], hb["commit_bar"], hb["exec_bar"], hb["snap_bar"])
        assert not reps[q].dump()["commit_bar"].any()

    # ---- replica 1 takes over -------------------------------------------------------------------------
    bl = reps[1].become_leader(u8(0))
    assert (bl["p_flags"] == 1).all() and (bl["p_trig"] == T - 2).all() and (bl["rc_n"] == T - 2).all()
    mine = held[1]                                               # the new leader's own shards
    # Prepare phase: the voted shards of replicas 2 and 3 come back (replica 3 never saw the open instances)
    for q in (2, 3):
        pr = reps[q].prepare(u8(1), u8(1), bl["p_trig"], bl["p_ballot"])
        acc = reps[1].prepare_replies(u8(q), **pr)
        for k in range(int(pr["pr_n"].max())):
            slot = T - 2 + k
            if (pr["pr_vbal"][k] > 0).all():                     # the reply row carries the peer's voted shard
                mine[slot].absorb_other(held[q][slot].subset_copy(int(pr["pr_vmask"][k][0])))
    d1 = reps[1].dump()
    # quorum of 3 reached with 2 shards ({1, 2}) per open instance and fewer than population - f = 4 replies: keep waiting
    assert (acc["a_n"] == 0).all() and (d1["s_status"][(T - 2) % W] == 1).all() and (d1["s_mask"][(T - 2) % W] == 0b00110).all()
    assert all(mine[s].avail == 0b00110 for s in (T - 2, T - 1))
    # replica 4 answers (never voted for them): 4 replies >= population - f, still 2 shards: the empty batch is chosen
    pr = reps[4].prepare(u8(1), u8(1), bl["p_trig"], bl["p_ballot"])
    acc = reps[1].prepare_replies(u8(4), **pr)
    assert (acc["a_n"] == 2).all() and (acc["a_val"][:2] == 0).all()
    # ... which is what RS(3,2) can promise with f = 1: an instance on fewer than 3 replicas is not recoverable, and it
    # was never committed (needed 4 acks).  The committed ones are: reconstruction reads bring their shards in.
    for q in (2, 3):
        rr = reps[q].reconstruct(u8(1), bl["rc_n"], bl["rc_slot"])
        assert (rr["rr_n"] == T - 2).all()
        for k in range(T - 2):
            slot = int(rr["rr_slot"][k][0])
            assert (rr["rr_val"][k] == tokens[slot]).all() and (rr["rr_mask"][k] == 1 << q).all()
            mine[slot].absorb_other(held[q][slot].subset_copy(1 << q))
        reps[1].reconstruct_reply(u8(1), **rr)
    d1 = reps[1].dump()
    # three shards {1, 2, 3} of every committed instance: the commit-bar run reconstructed and executed them
    assert (d1["commit_bar"] == T - 2).all() and (d1["exec_bar"] == T - 2).all()
    assert np.array_equal(d1["digest"], d0["digest"])            # same commands in the same order as the old leader ran
    for slot in range(T - 2):
        assert (d1["s_val"][slot % W] == tokens[slot]).all() and (d1["s_mask"][slot % W] == 0b01111).all()
        cw = mine[slot]
        assert cw.avail == 0b01110                               # bytes: shards 1, 2, 3 -- one data shard is missing
        cw.reconstruct_data()                                    # what the engine's mask |= data_mask stands for
        assert cw.avail == 0b01111
        assert np.array_equal(cw.get_data().cpu().numpy(), _payload(tokens[slot], L)), slot
        cw.compute_parity()
        assert cw.avail == 0b11111 and cw.verify_parity().all()
        assert np.array_equal(cw.buf.cpu().numpy(), held[0][slot].buf.cpu().numpy())   # byte for byte the old leader's codeword
