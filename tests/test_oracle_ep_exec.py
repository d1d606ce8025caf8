"""EPaxos dependency-graph execution of the CPU oracle against hand-derived traces of the reference
(src/protocols/epaxos/execution.rs:25-149 attempt_execution, :152-211 handle_cmd_result,
durability.rs:104-163 the attempts after a commit-bar advance).

The replica under test is id 4; rows 0-3 fill through CommitNotices, so every instance is
placed with the (seq, deps) the trace needs.  The expected SUBMISSION ORDER of each step is derived
by hand in the comments; kv / digest are then computed from that order by the few lines of
`_Model` (kv[key] = token, result = old token)."""
import numpy as np

N = 0xFFFFFFFF
COMMITTED, EXECUTING, EXECUTED = 3, 4, 5
MUL = 0x100000001B3
M64 = (1 << 64) - 1


def tok(row, col):
    return ((row + 1) << 32) | col


class _Model:
    def __init__(self):
        self.kv, self.digest, self.n, self.pending = {}, 0, 0, []

    def run(self, order):
        self.pending += [(row, col) for row, col, _ in order]
        for row, col, key in order:
            t, old = tok(row, col), self.kv.get(key, 0)
            self.kv[key] = t
            self.digest = ((self.digest ^ t) * MUL) & M64
            self.digest = ((self.digest ^ old) * MUL) & M64
            self.n += 1


def _commit(o, row, col, key, seq, *deps, R=5):
    d = np.full((R, 1), N, np.uint32)
    for r, c in deps:
        d[r, 0] = c
    o.handle_commit_notice(flags=np.array([1], np.uint8), peer=np.array([row], np.uint8), col=np.array([col], np.uint32),
                           ballot=np.array([row + 1], np.uint64), seq=np.array([seq], np.uint64), deps=d,
                           key=np.array([key], np.uint8))


def _check(o, m, exec_bars, **counters):
    _, r, c = o.take_submissions()                             # the hand-derived submission order itself
    assert [(int(a), int(b)) for a, b in zip(r, c)] == m.pending, (list(zip(r, c)), m.pending)
    m.pending = []
    x = o.exec_dump()
    assert [int(v) for v in x["exec_bars"][:, 0]] == exec_bars, x["exec_bars"][:, 0]
    assert int(x["digest"][0]) == m.digest
    for k in range(o.n_keys):
        assert int(x["kv"][k, 0]) == m.kv.get(k, 0), k
    names = ("n_exec", "n_reexec", "n_unheld", "n_multi_scc", "n_attempts", "n_aborts")
    got = dict(zip(names, (int(v) for v in x["counters"])))
    assert got["n_exec"] == m.n and got["n_multi_scc"] == 0
    for k, v in counters.items():
        assert got[k] == v, (k, got)


def _st(o, row, col):
    return int(o.dump()["status"][row, col % o.W, 0])


def _new(oracle):
    return oracle.EpOracle(1, 5, me=4, W=8, n_keys=4, execute=True)


def test_execution_traces(oracle):
    trace_execution(_new(oracle))


def test_path_order_and_sibling_order(oracle):
    trace_path_order(_new(oracle))


def test_two_children_newest_edge_first(oracle):
    trace_two_children(_new(oracle))


def test_instance_outside_the_ring_is_pruned(oracle):
    trace_outside_the_ring(_new(oracle))


TRACES = ("trace_execution", "trace_path_order", "trace_two_children", "trace_outside_the_ring")


def trace_execution(o):
    m = _Model()
    # A. (0,0) alone: commit bar of row 0 -> 1, attempt on tail (0,0): one node, submitted; its result
    #    arrives once the handler is back (rule 0): Executed, exec bar of row 0 -> 1
    _commit(o, 0, 0, 1, 1)
    m.run([(0, 0, 1)])
    _check(o, m, [1, 0, 0, 0, 0], n_attempts=1, n_aborts=0)
    assert _st(o, 0, 0) == EXECUTED

    # B. (1,0) depends on (0,1), which is not committed: the walk pops (1,0), then (0,1) whose column is
    #    not below row 0's commit bar -> the attempt is abandoned, nothing changes (execution.rs:41-45)
    _commit(o, 1, 0, 1, 2, (0, 1))
    _check(o, m, [1, 0, 0, 0, 0], n_attempts=2, n_aborts=1)
    assert _st(o, 1, 0) == COMMITTED
    #    (0,1) depends on (1,0).  Pops: (0,1) new, pushes (1,0) and its row predecessor (0,0);
    #    (1,0) new, edge (0,1)->(1,0), pushes (0,1); (0,0) executed: pruned; (0,1) already a node.
    #    Forest: (0,1) -> (1,0).  Post-order: (1,0), (0,1).
    #    Re-attempts (durability.rs:141-159): no row's tail is still Committed.  Results: both Executed,
    #    exec bars of rows 1 and 0 move.
    _commit(o, 0, 1, 1, 2, (1, 0))
    m.run([(1, 0, 1), (0, 1, 1)])
    _check(o, m, [2, 1, 0, 0, 0], n_attempts=3, n_aborts=1, n_reexec=0)

    # C. (2,0) depends on (0,2), uncommitted: abandoned.
    _commit(o, 2, 0, 1, 3, (0, 2))
    _check(o, m, [2, 1, 0, 0, 0], n_attempts=4, n_aborts=2)
    #    (0,2) depends on (0,1) and (2,0).  Pops:
    #      (0,2) new [node 0], pushes (0,1), (2,0), row predecessor (0,1)
    #      (0,1) Executed: pruned                                  last = (0,1)
    #      (2,0) new [node 1]; edge last->(2,0) = (0,1)->(2,0): GraphMap::add_edge inserts the missing
    #            endpoint (0,1) [node 2]; pushes (0,2)
    #      (0,1) pruned; (0,2) already a node
    #    Graph: nodes (0,2), (2,0), (0,1); one edge (0,1)->(2,0).  tarjan_scc starts from nodes in index
    #    order: (0,2) has no outgoing edge -> first; (2,0) -> second; (0,1) -> last.
    #    So the instance is submitted BEFORE the one it depends on, and (0,1) runs a second time, leaving
    #    its value in the store.
    _commit(o, 0, 2, 1, 3, (0, 1), (2, 0))
    m.run([(0, 2, 1), (2, 0, 1), (0, 1, 1)])
    _check(o, m, [3, 1, 1, 0, 0], n_attempts=5, n_aborts=2, n_reexec=1)
    assert m.kv[1] == tok(0, 1)
    assert _st(o, 0, 1) == EXECUTED and _st(o, 2, 0) == EXECUTED and _st(o, 0, 2) == EXECUTED

    # D. (3,0) on key 2 depends on (1,1), uncommitted: abandoned.  Then (1,1) on key 3, no deps:
    #    attempt on (1,1): pops (1,1) new, row predecessor (1,0) Executed: pruned -> submits (1,1).
    #    Re-attempts: row 3's tail (3,0) is Committed and its commit bar 1 > exec bar 0 -> attempt on
    #    (3,0): pops (3,0) new, (1,1) Executing: pruned -> submits (3,0).  Results in that order.
    _commit(o, 3, 0, 2, 1, (1, 1))
    _check(o, m, [3, 1, 1, 0, 0], n_attempts=6, n_aborts=3)
    _commit(o, 1, 1, 3, 1)
    m.run([(1, 1, 3), (3, 0, 2)])
    _check(o, m, [3, 2, 1, 1, 0], n_attempts=8, n_aborts=3, n_reexec=1)


def trace_path_order(o):
    """T = (0,1) with three unexecuted dependencies: each new node hangs under the slot popped before it, so
    the forest is the path T -> d1 -> d2 -> d3 and the post-order is d3, d2, d1, T -- unless a pruned
    pop sits in between, which starts a new tree."""
    m = _Model()
    # three committed instances that cannot run: each waits for (0,1)
    for row in (1, 2, 3):
        _commit(o, row, 0, row, 1, (0, 1))
    _check(o, m, [0, 0, 0, 0, 0], n_attempts=3, n_aborts=3)
    _commit(o, 0, 0, 0, 1)                                     # (0,0): runs alone
    m.run([(0, 0, 0)])
    # (0,1) depends on (1,0), (2,0), (3,0).  Pops: (0,1) new [pushes (1,0), (2,0), (3,0), (0,0)];
    # (1,0) new, edge (0,1)->(1,0) [pushes (0,1)]; (2,0) new, edge (1,0)->(2,0) [pushes (0,1)];
    # (3,0) new, edge (2,0)->(3,0) [pushes (0,1)]; (0,0) pruned; (0,1) x3 already nodes.
    _commit(o, 0, 1, 0, 2, (1, 0), (2, 0), (3, 0))
    m.run([(3, 0, 3), (2, 0, 2), (1, 0, 1), (0, 1, 0)])
    _check(o, m, [2, 1, 1, 1, 0], n_reexec=0)


def trace_two_children(o):
    """A node that is popped twice can get two children; Graph::neighbors walks the newer edge first.
    (0,2) depends on (1,0) and (1,1)... built so that (0,2) is `last` for two different new nodes."""
    m = _Model()
    _commit(o, 0, 0, 0, 1)
    m.run([(0, 0, 0)])
    # blocked on (0,1): (1,0) [deps (0,1)], (2,0) [deps (0,1)]
    _commit(o, 1, 0, 1, 1, (0, 1))
    _commit(o, 2, 0, 2, 1, (0, 1))
    # (0,1) depends on (1,0) only.  Pops:
    #   (0,1) new [n0]; pushes (1,0), row predecessor (0,0)
    #   (1,0) new [n1], edge (0,1)->(1,0); pushes (0,1)
    #   (0,0) pruned; (0,1) already a node                       last = (0,1)
    # one tree (0,1)->(1,0): submits (1,0), (0,1).  Then the re-attempts: row 2's tail (2,0) is still
    # Committed -> attempt on (2,0): pops (2,0) new, (0,1) Executing: pruned -> submits (2,0).
    _commit(o, 0, 1, 0, 2, (1, 0))
    m.run([(1, 0, 1), (0, 1, 0), (2, 0, 2)])
    _check(o, m, [2, 1, 1, 0, 0], n_reexec=0)
    # now a tail whose dependency list names the same unexecuted slot from two sides:
    # (3,0) and (3,1) wait for (0,2); (0,2) depends on (3,1) and (1,1); (1,1) waits for (0,2) too.
    _commit(o, 3, 0, 3, 1, (0, 2))
    _commit(o, 3, 1, 3, 2, (0, 2))
    _commit(o, 1, 1, 1, 2, (0, 2))
    # Pops for tail (0,2):
    #   (0,2) new [n0]; pushes (1,1), (3,1), row predecessor (0,1)
    #   (1,1) new [n1], edge (0,2)->(1,1); pushes (0,2), row predecessor (1,0)
    #   (3,1) new [n2], edge (1,1)->(3,1); pushes (0,2), row predecessor (3,0)
    #   (0,1) pruned                                             last = (0,1)
    #   (0,2) already a node                                     last = (0,2)
    #   (1,0) pruned                                             last = (1,0)
    #   (0,2) already a node                                     last = (0,2)
    #   (3,0) new [n3], edge (0,2)->(3,0); pushes (0,2)
    #   (0,2) already a node
    # (0,2) has two outgoing edges: to (1,1) [older] and to (3,0) [newer]: neighbors() yields (3,0) first.
    # Post-order from n0: (3,0); then (1,1)'s subtree: (3,1), (1,1); then (0,2).
    _commit(o, 0, 2, 0, 3, (1, 1), (3, 1))
    m.run([(3, 0, 3), (3, 1, 3), (1, 1, 1), (0, 2, 0)])
    _check(o, m, [3, 2, 1, 2, 0], n_reexec=0)


def trace_outside_the_ring(o):
    """harness guard: a dependency that left its row's ring of W columns counts as executed"""
    m = _Model()
    for c in range(10):                                        # row 0 runs to column 9: columns 0, 1 leave the ring
        _commit(o, 0, c, 0, c + 1)
        m.run([(0, c, 0)])
    _commit(o, 1, 0, 1, 20, (0, 0))                            # depends on (0,0), no longer held
    m.run([(1, 0, 1)])
    x = o.exec_dump()
    assert int(x["counters"][2]) >= 1
    _check(o, m, [10, 1, 0, 0, 0])


def test_execute_off_leaves_everything_alone(oracle):
    o = oracle.EpOracle(1, 5, me=4, W=8, n_keys=4)
    _commit(o, 0, 0, 1, 1)
    x = o.exec_dump()
    assert not x["exec_bars"].any() and not x["kv"].any() and not x["counters"].any()
    assert _st(o, 0, 0) == COMMITTED
