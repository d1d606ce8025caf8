"""EPaxos explicit prepare on the engine (summerset_amd/csrc/ep_engine.hip: ep_heartbeat_timeout_kernel, ep_exp_prepare_kernel,
ep_exp_prepare_replies_kernel and the row-general reply handlers, through the C-ABI): the hand-derived traces of
tests/test_oracle_ep_recovery.py run on it, and a crash-and-recovery run of a five-replica cluster where every output of
every call and the full state (instances, leader bookkeeping, exp_prepare_voteds) equal the oracle cluster's -- bit-exact."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import ep_cluster as ec  # noqa: E402
import test_oracle_ep_recovery as tr  # noqa: E402

pytestmark = pytest.mark.gpu


class _EngineAsOracle:
    """what the traces construct as `oracle.EpOracle(G, R, me=, W=, n_keys=)`"""

    def __init__(self, cuda):
        self.cuda = cuda

    def EpOracle(self, G, R=5, me=0, W=32, n_keys=64, optimized_quorum=True, execute=False):
        from summerset_amd import EPaxosReplicaGroup
        return ec.NumpyEngine(EPaxosReplicaGroup(G, R, me=me, window=W, n_keys=n_keys, optimized_quorum=optimized_quorum, recovery=True,
                                                 execute=execute), self.cuda)


TRACES = [tr.test_heartbeat_timeout_starts_exp_prepare_on_the_peers_row, tr.test_heartbeat_timeout_skips_committed_executed_and_foreign_instances,
          tr.test_exp_prepare_acceptor, tr.test_next_step_needs_a_simple_quorum, tr.test_next_step_committed_reply_wins,
          tr.test_next_step_accepting_reply_and_higher_voted_ballot, tr.test_next_step_enough_identical_pre_accepts_go_to_accept,
          tr.test_next_step_differing_pre_accepts_start_over_and_avoid_the_fast_path, tr.test_next_step_nothing_voted_is_a_noop,
          tr.test_suspected_peer_releases_my_own_waiting_instance]


@pytest.mark.parametrize("trace", TRACES, ids=lambda f: f.__name__[5:])
def test_trace_on_the_engine(cuda, trace):
    trace(_EngineAsOracle(cuda))


@pytest.mark.parametrize("G,seed,loss", [(700, 0, 0.0), (700, 3, 0.2), (2100, 4, 0.1)])
def test_crash_and_recovery_matches_the_oracle_cluster(cuda, oracle, G, seed, loss):
    W = 16
    te, to = [], []
    eng = _EngineAsOracle(cuda)
    re_, live, cut_e, tal_e = tr.run_crash_and_recovery(lambda G, R, r, W, K: eng.EpOracle(G, R, me=r, W=W, n_keys=K), seed, loss, G, W, trace=te)
    ro, _, cut_o, tal_o = tr.run_crash_and_recovery(lambda G, R, r, W, K: oracle.EpOracle(G, R, me=r, W=W, n_keys=K), seed, loss, G, W, trace=to)
    assert (cut_e == cut_o).all() and tal_e == tal_o and len(te) == len(to)
    for i, (a, b) in enumerate(zip(te, to)):
        assert a[0] == b[0]
        for x, y in zip(a[1:], b[1:]):
            assert np.array_equal(x, y), (i, a[0])
    for q in range(5):
        a, b = re_[q].dump(), ro[q].dump()
        for n in b:
            assert np.array_equal(a[n], b[n]), (q, n)
        a, b = re_[q].xp_dump(), ro[q].xp_dump()
        for n in b:
            assert np.array_equal(a[n], b[n]), (q, "xp", n)
    assert ec.check_agreement(re_, live, 0, G) > 0
    assert sum(t[1] for t in tal_o) > 0 and sum(t[2] for t in tal_o) > 0 and sum(t[3] for t in tal_o) > 0


def test_recovery_needs_the_flag(cuda):
    import torch
    from summerset_amd import EPaxosReplicaGroup, SummersetError
    e = EPaxosReplicaGroup(8, 5, me=1, window=8, n_keys=4)
    src = torch.zeros(8, dtype=torch.uint8, device=cuda)
    with pytest.raises(SummersetError):
        e.heartbeat_timeout(src)
    EPaxosReplicaGroup(8, 5, me=1, window=8, n_keys=4, execute=True, recovery=True).close()   # (round 2: they run together)
