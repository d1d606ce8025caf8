"""The reference's own `Bitmap` unit tests (src/utils/bitmap.rs:312-387), restated one for one against the oracle's
restatement of the type (oracle/bitmap_oracle.c), plus the convention every engine and oracle of this repository relies
on: the integer mask of a bitmap has bit i set iff id i is true -- accept_acks, prepare_acks, rq_acks,
avail_shards_map, peer_alive are all held that way."""
import numpy as np
import pytest


@pytest.fixture()
def B(oracle):
    return oracle.Bitmap


def test_new_invalid(B):                       # :316-320  #[should_panic = "invalid bitmap size 0"]
    with pytest.raises(AssertionError, match="invalid bitmap size 0"):
        B(0, True)


def test_conversions(B):                       # :322-337
    ref_map = B.from_ones(5, range(1, 4))      # Bitmap::from((5, 1..4))
    assert B.from_ones(5, [1, 2, 3]) == ref_map
    assert B.from_ones(5, {1, 2, 3}) == ref_map
    assert ref_map.to_vec() == [1, 2, 3]
    assert set(ref_map.to_vec()) == {1, 2, 3}


def test_bitmap_set_get(B):                    # :339-351
    m = B(7, False)
    m.set(0, True)
    m.set(1, False)
    m.set(2, True)
    with pytest.raises(ValueError):
        m.set(7, True)
    assert m.get(0) is True and m.get(1) is False and m.get(2) is True and m.get(3) is False
    with pytest.raises(ValueError):
        m.get(7)


def test_bitmap_flip(B):                       # :353-359
    m = B(5, False)
    m.set(1, True)
    m.flip()
    assert m == B.from_ones(5, [0, 2, 3, 4])


def test_bitmap_union(B):                      # :361-367
    a, b = B.from_ones(5, [0, 1, 3]), B.from_ones(5, [0, 4])
    a.union(b)
    assert a == B.from_ones(5, [0, 1, 3, 4])
    with pytest.raises(ValueError):             # :124-133 sizes mismatch
        a.union(B(6, False))


def test_bitmap_count(B):                      # :369-377
    m = B(7, False)
    assert m.count() == 0
    for i in (0, 2, 3):
        m.set(i, True)
    assert m.count() == 3


def test_bitmap_iter(B):                       # :379-388
    ref = [True, True, False, True, True]
    m = B(5, True)
    m.set(2, False)
    for i, flag in m.iter():
        assert ref[i] == flag
    assert m.to_vec() == [0, 1, 3, 4]


def test_bincode_encode_decode(B):             # :390-403: a round trip in the reference; here the bytes themselves
    m = B.from_ones(10, [0, 2, 3, 9])
    m.set(5, True)
    enc = m.bincode()                          # usize len, then the block slice: 0A | 01 | varint(0b10_0010_1101 = 0x22D)
    assert enc == bytes([10, 1, 0xFB, 0x2D, 0x02])
    assert B.from_ones(5, [0, 3]).bincode() == bytes([5, 1, 9])


def test_bincode_length_bits(B):               # :404-419: 24 bits, a round trip in the reference; here the bytes and the bits back
    ones = [1, 5, 7, 12, 17, 23]
    m = B(24, False)
    for i in ones:
        m.set(i, True)
    word = sum(1 << i for i in ones)           # one 32-bit block: 0x008210A2
    enc = m.bincode()                          # usize len 24 | one block | varint(u32 >= 2^16: 0xFC + 4 bytes LE)
    assert enc == bytes([24, 1, 0xFC]) + word.to_bytes(4, "little")
    assert m.size() == 24 and [m.get(i) for i in range(24)] == [i in ones for i in range(24)]


def test_mask_convention_and_quorum_freeze(B):
    """bit i of the mask = id i; the accept-ack bitmap of handle_msg_accept_reply (multipaxos/messages.rs:404-412)
    freezes at exactly quorum_cnt bits in arrival order -- the same statement in Bitmap terms as the engine's tally"""
    rng = np.random.default_rng(5)
    for _ in range(200):
        size = int(rng.integers(1, 9))
        ones = [int(i) for i in range(size) if rng.random() < 0.5]
        m = B.from_ones(size, ones)
        assert m.mask() == sum(1 << i for i in ones) and m.count() == bin(m.mask()).count("1")
    population, quorum = 5, 3
    for _ in range(50):
        order = [int(x) for x in rng.permutation(population)]
        acks, committed = B(population, False), False
        for peer in order:
            if committed or acks.get(peer):     # :394-406: not Accepting any more / duplicate
                continue
            acks.set(peer, True)
            committed = acks.count() >= quorum  # :412
        assert acks.count() == quorum and acks.to_vec() == sorted(order[:quorum])
