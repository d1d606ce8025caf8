"""Synthetic stream generator: determinism and encoding (CPU only)."""
import numpy as np

from summerset_amd import stream


def test_splitmix64_reference_values():
    # SplitMix64 with state 0: first outputs of the canonical generator (seed advanced by the golden gamma)
    x = np.uint64(0)
    outs = []
    for _ in range(3):
        outs.append(int(stream.splitmix64(np.array([x]))[0]))
        with np.errstate(over="ignore"):
            x = x + np.uint64(0x9E3779B97F4A7C15)
    assert outs == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_ackctl_is_a_permutation_with_drop_mask():
    w = stream.random_ackctl(1, 7, 5, 200, 5, 0.1)
    assert w.shape == (5, 200) and w.dtype == np.uint32
    for word in w.reshape(-1)[:300]:
        ids = [(int(word) >> (3 * i)) & 7 for i in range(8)]
        assert sorted(ids[:5]) == [0, 1, 2, 3, 4] and ids[5:] == [5, 6, 7]
        assert (int(word) >> 24) < 32
    drops = np.unpackbits((w >> 24).astype(np.uint8)[..., None], axis=-1).mean() * 8 / 5
    assert 0.05 < drops < 0.15
    padded = stream.random_ackctl(1, 7, 2, 8, 5, 0.1, cap=6)
    assert (padded[2:] == stream.CTL_IDENTITY).all()


def test_stream_is_deterministic_and_follows_the_leader():
    a = stream.MultiPaxosStream(64, 5, 2, cap=12, n_ticks=32, timeout_frac=0.5, seed=5)
    b = stream.MultiPaxosStream(64, 5, 2, cap=12, n_ticks=32, timeout_frac=0.5, seed=5)
    for t in (0, 9, 31):
        x, y = a.tick(t), b.tick(t)
        for k in x:
            assert np.array_equal(x[k], y[k])
        assert (x["req_val"] != 0).all()
    tt = a.timeout_tick
    g = int(np.nonzero(tt >= 0)[0][0])
    at = a.tick(int(tt[g]))
    assert at["timeout_rep"][g] == 1 and at["timeout_src"][g] == 0 and at["req_target"][g] == 0
    assert a.tick(int(tt[g]) + 1)["req_target"][g] == 1
    assert [a.heartbeat(t) for t in range(8)] == [False, False, False, True] * 2
