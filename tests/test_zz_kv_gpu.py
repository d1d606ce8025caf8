"""Device-resident KV state machine: `kv_execute_kernel` through the C-ABI against the reference's own
state-machine tests (src/server/statemach.rs:229-337: get_empty, put_one_get_one, put_twice, put_rand_get_rand -- keys and
values as tokens) and, per group, against a Python dict on random command lists; then a stable leader's ReadQuery
answered from that table (multipaxos/quorumread.rs:99-147)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GET, PUT, NOP = 0, 1, 0xFF


def _exec(sm, cuda, cmds):
    """cmds: list of rows, each (kind, key, val) broadcast to every group, or arrays [G]"""
    import torch
    G = sm.G
    col = lambda x, dt: np.broadcast_to(np.asarray(x, dt), (G,)).copy()
    kind = np.stack([col(c[0], np.uint8) for c in cmds]); key = np.stack([col(c[1], np.uint8) for c in cmds])
    val = np.stack([col(c[2], np.uint32) for c in cmds])
    res = sm.execute(torch.from_numpy(kind).to(cuda), torch.from_numpy(key).to(cuda), torch.from_numpy(val.view(np.int32)).to(cuda))
    return res.cpu().numpy().view(np.uint32)


def test_reference_state_machine_tests(cuda):
    from summerset_amd import KvStateMachine
    JOSE = 7
    sm = KvStateMachine(3, 16)
    assert (_exec(sm, cuda, [(GET, JOSE, 0)]) == 0).all()                             # get_empty: value None
    sm = KvStateMachine(3, 16)
    r = _exec(sm, cuda, [(PUT, JOSE, 180), (GET, JOSE, 0)])                           # put_one_get_one
    assert (r[0] == 0).all() and (r[1] == 180).all()
    sm = KvStateMachine(3, 16)
    r = _exec(sm, cuda, [(PUT, JOSE, 180), (PUT, JOSE, 185)])                         # put_twice: old_value Some("180")
    assert (r[0] == 0).all() and (r[1] == 180).all() and (sm.dump()[JOSE] == 185).all()


def test_put_rand_get_rand_per_group(cuda):
    """the reference's random test, one independent state per group, rows with no command in between"""
    from summerset_amd import KvStateMachine
    rng = np.random.default_rng(4)
    G, K, rows = 300, 36, 60
    sm = KvStateMachine(G, K)
    ref = [dict() for _ in range(G)]
    for call in range(5):
        kind = rng.choice(np.array([GET, PUT, PUT, NOP], np.uint8), (rows, G))
        key = rng.integers(0, K, (rows, G)).astype(np.uint8)
        val = rng.integers(1, 1 << 31, (rows, G)).astype(np.uint32)
        res = _exec(sm, cuda, [(kind[i], key[i], val[i]) for i in range(rows)])
        for g in range(G):
            for i in range(rows):
                k = int(key[i, g])
                if kind[i, g] == GET:
                    want = ref[g].get(k, 0)
                elif kind[i, g] == PUT:
                    want = ref[g].get(k, 0); ref[g][k] = int(val[i, g])
                else:
                    want = 0
                assert res[i, g] == want, (call, i, g)
    table = sm.dump()
    for g in range(0, G, 17):
        assert {k: int(table[k, g]) for k in range(K) if table[k, g]} == ref[g]


def test_stable_leader_reads_the_executed_state(cuda, oracle):
    """committed Puts executed on the device, then ReadQueries answered by the stable leader from the same table"""
    import torch
    from summerset_amd import KvStateMachine, QuorumReadGroup
    G, K, B = 64, 8, 3
    sm = KvStateMachine(G, K)
    _exec(sm, cuda, [(PUT, 2, np.arange(100, 100 + G)), (PUT, 5, 77), (PUT, 2, np.arange(500, 500 + G))])
    q = QuorumReadGroup(G, 5, 0, K, B, 1)
    keys = torch.from_numpy(np.stack([np.full(G, 2, np.uint8), np.full(G, 6, np.uint8), np.full(G, 5, np.uint8)])).to(cuda)
    n = torch.full((G,), B, dtype=torch.uint8, device=cuda)
    z32 = lambda *s: torch.zeros(s, dtype=torch.int32, device=cuda)
    log = dict(start_slot=z32(G), log_end=z32(G), status=torch.zeros((8, G), dtype=torch.uint8, device=cuda), token=z32(8, G))
    out, fl = q.handle_msg_read_query(keys, n, log, stable_leader=torch.ones(G, dtype=torch.uint8, device=cuda), kv=sm.table_ptr())
    assert (fl.cpu().numpy() == 1).all()
    st, vl = out["state"].cpu().numpy(), out["val"].cpu().numpy()
    assert (st[0] == 2).all() and (vl[0] == np.arange(500, 500 + G)).all()
    assert (st[1] == 0).all() and (st[2] == 2).all() and (vl[2] == 77).all()
    o = oracle.QrOracle(G, 5, 0, K, B, 1)
    lg = dict(start_slot=np.zeros(G, np.uint32), log_end=np.zeros(G, np.uint32), status=np.zeros((8, G), np.uint8), token=np.zeros((8, G), np.uint32))
    want, _ = o.handle_read_query(keys.cpu().numpy(), n.cpu().numpy(), lg, np.ones(G, np.uint8), sm.dump())
    assert np.array_equal(want["state"], st) and np.array_equal(want["val"], vl.view(np.uint32))
