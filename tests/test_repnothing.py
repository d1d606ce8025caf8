"""RepNothing + KV state machine (BASELINE config 1, CPU only): the reference's own
state-machine tests (src/server/statemach.rs:229-290) restated through the C-ABI, the
tester's old-value chain (summerset_client tester.rs:401-410), and the config-1 stream shape
(Put, 5 keys "k%07d", batches of 1)."""
import numpy as np

from summerset_amd.repnothing import RepNothingReplica


def test_get_empty():                                   # statemach.rs:229-239
    r = RepNothingReplica()
    _, rep = r.handle_req_batch([(1, 0, ("get", "Jose"))])
    assert rep == [(1, 0, "get", None)]


def test_put_one_get_one():                             # statemach.rs:241-263
    r = RepNothingReplica()
    assert r.handle_req_batch([(1, 0, ("put", "Jose", "180"))])[1] == [(1, 0, "put", None)]
    assert r.handle_req_batch([(1, 1, ("get", "Jose"))])[1] == [(1, 1, "get", b"180")]


def test_put_twice():                                   # statemach.rs:265-290
    r = RepNothingReplica()
    assert r.handle_req_batch([(1, 0, ("put", "Jose", "180"))])[1] == [(1, 0, "put", None)]
    assert r.handle_req_batch([(1, 1, ("put", "Jose", "185"))])[1] == [(1, 1, "put", b"180")]


def test_batch_order_and_instance_index():
    r = RepNothingReplica()
    i0, rep = r.handle_req_batch([(7, 0, ("put", "a", "1")), (8, 0, ("get", "a")), (7, 1, ("put", "a", "2")),
                                  (8, 1, ("get", "b"))])
    assert i0 == 0
    assert rep == [(7, 0, "put", None), (8, 0, "get", b"1"), (7, 1, "put", b"1"), (8, 1, "get", None)]
    i1, _ = r.handle_req_batch([(9, 0, ("get", "a"))])
    assert i1 == 1 and r.stats()["instances"] == 2 and r.stats()["executed"] == 5 and r.stats()["keys"] == 1


def test_config1_stream_old_value_chain():
    rng = np.random.default_rng(0x5EED5EED)
    r = RepNothingReplica()
    alnum = np.frombuffer(b"0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", np.uint8)
    last, wal = {}, 0
    for i in range(2000):
        key = "k%07d" % (i % 5)
        val = alnum[rng.integers(0, 62, 1024)].tobytes()
        _, rep = r.handle_req_batch([(3, i, ("put", key, val))])
        assert rep == [(3, i, "put", last.get(key))]
        last[key] = val
        # framed bincode WalEntry: 8 | vec len 1 | client 1 | Req 1 | id varint | Put 1 | key 1+8 | value 3+1024
        wal += 8 + 1 + 1 + 1 + (1 if i < 251 else 3) + 1 + 9 + 1027
    s = r.stats()
    assert s == {"instances": 2000, "wal_offset": wal, "executed": 2000, "keys": 5}


def test_large_value_roundtrip():
    r = RepNothingReplica()
    big = bytes(range(256)) * 1024                      # 256 KiB > the default reply buffer
    r.handle_req_batch([(1, 0, ("put", "big", big))])
    assert r.handle_req_batch([(1, 1, ("get", "big"))])[1] == [(1, 1, "get", big)]
