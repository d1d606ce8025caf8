"""The string-key device KV state machine (summerset_amd/csrc/skv_exec.hip, `smr_skv_*`) against the reference's own
state-machine tests (src/server/statemach.rs:229-337: get_empty, put_one_get_one, put_twice, put_rand_get_rand with real
strings), against a Python dict per group on random command lists, and against the HOST state machine of the RepNothing
path (`smr_repnothing_*`, the same `HashMap<String, String>` semantics in C++) fed the same commands."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GET, PUT, NOP = 0, 1, 0xFF


def _run(sm, dev, rows):
    """rows: list of lists (one entry per group) of ("get", key) | ("put", key, value) | None -> results as Python values"""
    import torch
    G = sm.G
    blob, kind = bytearray(), np.full((len(rows), G), NOP, np.uint8)
    ko, kl, vo, vl = (np.zeros((len(rows), G), np.int32) for _ in range(4))
    for i, row in enumerate(rows):
        for g, c in enumerate(row):
            if c is None:
                continue
            kind[i, g] = PUT if c[0] == "put" else GET
            ko[i, g], kl[i, g] = len(blob), len(c[1]); blob += c[1]
            if c[0] == "put":
                vo[i, g], vl[i, g] = len(blob), len(c[2]); blob += c[2]
    t = lambda a: torch.from_numpy(a).to(dev)
    payload = torch.from_numpy(np.frombuffer(bytes(blob) or b"\0", np.uint8).copy()).to(dev)
    st, off, ln = sm.execute(t(kind), payload, t(ko), t(kl), t(vo), t(vl))
    st, off, ln = st.cpu().numpy(), off.cpu().numpy(), ln.cpu().numpy()
    out = []
    for i in range(len(rows)):
        out.append([None if st[i, g] == 0 else ("FULL" if st[i, g] == 2 else sm.heap_bytes_of(g, off[i, g], ln[i, g])) for g in range(G)])
    return out


def test_reference_state_machine_tests(cuda):
    from summerset_amd import StringKvStateMachine
    G = 3
    every = lambda c: [c] * G
    sm = StringKvStateMachine(G)
    assert _run(sm, cuda, [every(("get", b"Jose"))]) == [[None] * G]                                     # get_empty
    sm = StringKvStateMachine(G)
    r = _run(sm, cuda, [every(("put", b"Jose", b"180")), every(("get", b"Jose"))])                        # put_one_get_one
    assert r == [[None] * G, [b"180"] * G]
    sm = StringKvStateMachine(G)
    r = _run(sm, cuda, [every(("put", b"Jose", b"180")), every(("put", b"Jose", b"185")), every(("get", b"Jose"))])   # put_twice
    assert r == [[None] * G, [b"180"] * G, [b"185"] * G]
    assert (sm.stats()["n_keys"] == 1).all()


def test_put_rand_get_rand_per_group_and_the_host_state_machine(cuda):
    """the reference's random test (:292-337): random alphanumeric keys and values, one independent state per group;
    the same commands go through the RepNothing host executor for group 0"""
    from summerset_amd import RepNothingReplica, StringKvStateMachine
    rng = np.random.default_rng(11)
    G, rows = 96, 40
    alnum = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789", np.uint8)
    rs = lambda n: alnum[rng.integers(0, 62, n)].tobytes()
    keys = [rs(int(rng.integers(1, 12))) for _ in range(24)] + [b"", b"k0000003"]
    sm = StringKvStateMachine(G, slots=64, heap_bytes=1 << 15)
    ref = [dict() for _ in range(G)]
    host = RepNothingReplica()
    req = 0
    for call in range(4):
        cmds = []
        for i in range(rows):
            row = []
            for g in range(G):
                u = rng.random()
                k = keys[int(rng.integers(len(keys)))]
                row.append(None if u < 0.15 else (("get", k) if u < 0.5 else ("put", k, rs(int(rng.integers(0, 40))))))
            cmds.append(row)
        got = _run(sm, cuda, cmds)
        for i in range(rows):
            for g in range(G):
                c = cmds[i][g]
                if c is None:
                    assert got[i][g] is None
                    continue
                want = ref[g].get(c[1])
                if c[0] == "put":
                    ref[g][c[1]] = c[2]
                assert got[i][g] == want, (call, i, g, c)
                if g == 0:                                          # the host state machine agrees (statemach.rs semantics in C++)
                    req += 1
                    _, reps = host.handle_req_batch([(1, req, c)])
                    assert len(reps) == 1 and reps[0][3] == want, (call, i, c, reps)
    st = sm.stats()
    assert not st["full"].any() and [int(x) for x in st["n_keys"]] == [len(d) for d in ref]


def test_full_table_and_heap_are_sticky_not_silent(cuda):
    from summerset_amd import StringKvStateMachine
    sm = StringKvStateMachine(2, slots=4, heap_bytes=64)
    puts = [[("put", bytes([65 + i]), b"v")] * 2 for i in range(4)]
    assert _run(sm, cuda, puts) == [[None, None]] * 4                       # four keys fill the four entries
    r = _run(sm, cuda, [[("put", b"Z", b"v"), ("get", b"A")]])
    assert r == [["FULL", b"v"]]                                           # a fifth key: refused, the group is frozen
    assert _run(sm, cuda, [[("get", b"A"), ("put", b"A", b"x" * 100)]]) == [["FULL", "FULL"]]   # group 1's heap cannot take 100 bytes
    assert sm.stats()["full"].tolist() == [1, 1]
    # bytes outside the payload buffer are refused, not read
    import torch
    one = lambda v, dt: torch.tensor([[v, v]], dtype=dt, device=cuda)
    sm2 = StringKvStateMachine(2)
    st, _, _ = sm2.execute(one(GET, torch.uint8), torch.zeros(4, dtype=torch.uint8, device=cuda), one(2, torch.int32), one(3, torch.int32),
                           one(0, torch.int32), one(0, torch.int32))
    assert st.cpu().numpy().tolist() == [[2, 2]]
