"""The reference's own LeaseManager unit tests (src/server/leaseman.rs:1079-2301) restated as explicit-time traces that run
against anything with the batched lease-manager interface (`step_arrays`, `attempt_refresh_arrays`, `dump`): the CPU oracle
(tests/test_oracle_lease.py: this pins the oracle) and the HIP object (tests/test_zz_lease_gpu.py, tests/test_hostsim.py).

The reference sleeps on tokio timers between steps; here each `sleep(ms)` advances an explicit clock shared by the nodes
and message passing between the two test nodes takes no time, which keeps every asserted interval on the same side of
every deadline as in the reference's test (margins there are >= 30 ms)."""
import numpy as np

N_NONE, N_NEW_GRANTS, N_DO_REVOKE, N_CLEAR_HELD, N_RECV_MSG = range(5)
GUARD, GUARD_REPLY, PROMISE, PROMISE_REPLY, REVOKE, REVOKE_REPLY = range(6)
A_SEND, A_BCAST, A_NEXT_REFRESH, A_GRANT_REMOVED, A_LEASE_CLEARED, A_GRANT_TIMEOUT, A_LEASE_TIMEOUT, A_HIGHER_NUMBER, \
    A_GUARD_ACCEPT_BAR = range(1, 10)
ALL = 0xFF


# actions as the reference spells them: (lease_num, LeaseAction)
def HigherNumber(n): return (n, "HigherNumber")
def Send(n, peer, msg, held=None, bar=None): return (n, "SendLeaseMsg", peer, msg, held, bar)
def Bcast(n, peers, msg, bar=None): return (n, "BcastLeaseMsgs", frozenset(peers), msg, bar)
def NextRefresh(n, peer): return (n, "NextRefresh", peer)
def GrantRemoved(n, peer, held): return (n, "GrantRemoved", peer, held)
def LeaseCleared(n): return (n, "LeaseCleared")
def GrantTimeout(n, peer): return (n, "GrantTimeout", peer)
def LeaseTimeout(n, peer): return (n, "LeaseTimeout", peer)
def GuardAcceptBar(n, peer, bar): return (n, "GuardAcceptBar", peer, bar)


def _bits(mask, R):
    return frozenset(p for p in range(R) if (mask >> p) & 1)


def decode_actions(n, out, g, R):
    acts = []
    for i in range(int(n[g])):
        k, num, peer, mask = int(out["kind"][i, g]), int(out["num"][i, g]), int(out["peer"][i, g]), int(out["mask"][i, g])
        msg, flag, bar = int(out["msg"][i, g]), int(out["flag"][i, g]), int(out["bar"][i, g])
        if k == A_SEND:
            held = bool(flag) if msg in (PROMISE_REPLY, REVOKE_REPLY) else None
            acts.append(Send(num, peer, msg, held))
        elif k == A_BCAST:
            acts.append(Bcast(num, _bits(mask, R), msg, bar if (msg == GUARD and flag) else None))
        elif k == A_NEXT_REFRESH:
            acts.append(NextRefresh(num, peer))
        elif k == A_GRANT_REMOVED:
            acts.append(GrantRemoved(num, peer, bool(flag)))
        elif k == A_LEASE_CLEARED:
            acts.append(LeaseCleared(num))
        elif k == A_GRANT_TIMEOUT:
            acts.append(GrantTimeout(num, peer))
        elif k == A_LEASE_TIMEOUT:
            acts.append(LeaseTimeout(num, peer))
        elif k == A_HIGHER_NUMBER:
            acts.append(HigherNumber(num))
        elif k == A_GUARD_ACCEPT_BAR:
            acts.append(GuardAcceptBar(num, peer, bar))
        else:
            raise AssertionError("unknown action kind %d" % k)
    return acts


class Clock:
    def __init__(self):
        self.now = 1000

    def sleep(self, ms):
        self.now += ms


class Node:
    """one replica's manager; every group of the batch is given the same notices and must answer the same"""

    def __init__(self, impl, clock):
        self.m, self.clock = impl, clock
        self.G, self.R = impl.G, impl.R
        self.pending = []        # actions not yet taken by get_action()

    def _step(self, kind=N_NONE, num=0, peer=0, peers=0, msg=0, held=0, bar=None):
        G = self.G
        f = lambda v, dt: np.full(G, v, dt)
        n, out = self.m.step_arrays(self.clock.now, f(kind, np.uint8), f(num, np.uint64), f(peer, np.uint8), f(peers, np.uint8),
                                    f(msg, np.uint8), f(held, np.uint8), f(0 if bar is None else 1, np.uint8),
                                    f(0 if bar is None else bar, np.uint64))
        acts = decode_actions(n, out, 0, self.R)
        for g in range(1, G):
            assert decode_actions(n, out, g, self.R) == acts
        self.pending += acts

    def _mask(self, peers):
        return ALL if peers is None else sum(1 << p for p in peers)

    # LeaseNotice
    def new_grants(self, num, peers, accept_bar=None): self._step(N_NEW_GRANTS, num, peers=self._mask(peers), bar=accept_bar)
    def do_revoke(self, num, peers): self._step(N_DO_REVOKE, num, peers=self._mask(peers))
    def clear_held(self, num): self._step(N_CLEAR_HELD, num)
    def recv(self, num, peer, msg, held=None, bar=None): self._step(N_RECV_MSG, num, peer=peer, msg=msg, held=1 if held else 0, bar=bar)

    def get_action(self):
        """the next action; like the reference's awaiting get_action it also sees timers that fired up to now"""
        if not self.pending:
            self._step()
        assert self.pending, "no action pending"
        return self.pending.pop(0)

    def no_action(self):
        self._step()
        return not self.pending

    def _dump(self, key):
        self._step()                 # timers up to now
        d = self.m.dump()[key]
        assert (d == d[0]).all()
        return int(d[0])

    def grant_set(self): return _bits(self._dump("grant_set"), self.R)
    def lease_set(self): return _bits(self._dump("lease_set"), self.R)
    def lease_cnt(self): return self._dump("lease_cnt")
    def active_num(self): return self._dump("active_num")

    def attempt_refresh(self, peers):
        self._step()
        o = self.m.attempt_refresh_arrays(self.clock.now, np.ones(self.G, np.uint8), np.full(self.G, self._mask(peers), np.uint8))
        assert (o == o[0]).all()
        return _bits(int(o[0]), self.R)


def _grant_until_promise(n0, n1, num=7, first=True):
    """the shared opening of the tests: 0 grants to {1} under `num`, 1 answers the Guard, 0 sends the Promise"""
    n0.new_grants(num, {1})
    if first:
        assert n0.get_action() == HigherNumber(num)
    assert n0.get_action() == Bcast(num, {1}, GUARD)
    n1.recv(num, 0, GUARD)
    if first:
        assert n1.get_action() == HigherNumber(num)
    assert n1.get_action() == Send(num, 0, GUARD_REPLY)


def guard_expired(mk):
    """leaseman.rs:1079-1168"""
    ck = Clock()
    n0, n1 = Node(mk(2, 0, 600), ck), Node(mk(2, 1, 600), ck)
    _grant_until_promise(n0, n1)
    # the GuardReply is deliberately not sent
    ck.sleep(30)
    assert n0.grant_set() == frozenset()
    ck.sleep(30)
    assert n1.lease_cnt() == 1
    ck.sleep(660)
    assert n1.get_action() == LeaseTimeout(7, 0)
    assert n1.lease_cnt() == 1
    assert n0.grant_set() == frozenset() and n0.no_action()


def promise_expired(mk):
    """leaseman.rs:1170-1313"""
    ck = Clock()
    n0, n1 = Node(mk(2, 0, 600), ck), Node(mk(2, 1, 600), ck)
    _grant_until_promise(n0, n1)
    n0.recv(7, 1, GUARD_REPLY)
    assert n0.get_action() == Send(7, 1, PROMISE)
    n1.recv(7, 0, PROMISE)
    assert n1.get_action() == Send(7, 0, PROMISE_REPLY, held=True)
    # the PromiseReply is deliberately not sent: the grantor waits T_guard + T_lease, the holder T_lease
    ck.sleep(30)
    assert n0.grant_set() == {1}
    ck.sleep(30)
    assert n1.lease_cnt() == 2
    ck.sleep(630)                                   # 0: +690 in all, 1: +720
    assert n0.grant_set() == {1}
    ck.sleep(30)
    assert n1.get_action() == LeaseTimeout(7, 0)
    assert n1.lease_cnt() == 1
    ck.sleep(630)                                   # 0: +1350
    assert n0.get_action() == GrantTimeout(7, 1)
    assert n0.grant_set() == frozenset()


def promise_refresh(mk):
    """leaseman.rs:1315-1556"""
    ck = Clock()
    n0, n1 = Node(mk(2, 0, 600), ck), Node(mk(2, 1, 600), ck)
    _grant_until_promise(n0, n1)
    assert n1.lease_cnt() == 1
    assert n0.grant_set() == frozenset()
    n0.recv(7, 1, GUARD_REPLY)
    assert n0.get_action() == Send(7, 1, PROMISE)
    assert n0.grant_set() == {1}
    for _ in range(2):
        n1.recv(7, 0, PROMISE)
        assert n1.get_action() == Send(7, 0, PROMISE_REPLY, held=True)
        assert n1.lease_cnt() == 2
        n0.recv(7, 1, PROMISE_REPLY, held=True)
        assert n0.get_action() == NextRefresh(7, 1)
        ck.sleep(120)
        assert n0.grant_set() == {1}
        assert n0.attempt_refresh({1}) == {1}
    n1.recv(7, 0, PROMISE)
    assert n1.get_action() == Send(7, 0, PROMISE_REPLY, held=True)
    # this PromiseReply is not sent: granted for 2 * T_lease since the last promise on 0, T_lease on 1
    ck.sleep(30)
    assert n0.grant_set() == {1}
    ck.sleep(30)
    assert n1.lease_cnt() == 2
    ck.sleep(630)
    assert n0.grant_set() == {1}
    ck.sleep(30)
    assert n1.get_action() == LeaseTimeout(7, 0)
    assert n1.lease_cnt() == 1
    ck.sleep(630)
    assert n0.get_action() == GrantTimeout(7, 1)
    assert n0.grant_set() == frozenset()


def _granted_and_refreshing(mk):
    ck = Clock()
    n0, n1 = Node(mk(2, 0, 600), ck), Node(mk(2, 1, 600), ck)
    _grant_until_promise(n0, n1)
    n0.recv(7, 1, GUARD_REPLY)
    assert n0.get_action() == Send(7, 1, PROMISE)
    assert n0.grant_set() == {1}
    n1.recv(7, 0, PROMISE)
    assert n1.get_action() == Send(7, 0, PROMISE_REPLY, held=True)
    assert n1.lease_cnt() == 2
    n0.recv(7, 1, PROMISE_REPLY, held=True)
    assert n0.get_action() == NextRefresh(7, 1)
    ck.sleep(120)
    assert n0.grant_set() == {1}
    return ck, n0, n1


def revoke_replied(mk):
    """leaseman.rs:1558-1745"""
    ck, n0, n1 = _granted_and_refreshing(mk)
    n0.do_revoke(7, {1})
    assert n0.get_action() == Bcast(7, {1}, REVOKE)
    assert n0.grant_set() == {1}
    n1.recv(7, 0, REVOKE)
    assert n1.get_action() == Send(7, 0, REVOKE_REPLY, held=True)
    assert n1.lease_cnt() == 1
    n0.recv(7, 1, REVOKE_REPLY, held=True)
    ck.sleep(60)
    assert n0.get_action() == GrantRemoved(7, 1, True)
    assert n0.grant_set() == frozenset()


def revoke_expired(mk):
    """leaseman.rs:1747-1927"""
    ck, n0, n1 = _granted_and_refreshing(mk)
    n0.do_revoke(7, {1})
    assert n0.get_action() == Bcast(7, {1}, REVOKE)
    n1.recv(7, 0, REVOKE)
    assert n1.get_action() == Send(7, 0, REVOKE_REPLY, held=True)
    assert n1.lease_cnt() == 1
    # the RevokeReply is not sent: 0 has to wait for the grant to time out
    assert n0.grant_set() == {1}
    ck.sleep(660)
    assert n0.grant_set() == frozenset()
    assert n0.get_action() == GrantTimeout(7, 1)
    assert n0.attempt_refresh({1}) == frozenset()     # and no further refreshes are scheduled


def regrant_higher(mk):
    """leaseman.rs:1929-2171"""
    ck, n0, n1 = _granted_and_refreshing(mk)
    n0.new_grants(8, {1})
    assert n0.get_action() == HigherNumber(8)
    assert n0.get_action() == Bcast(8, {1}, GUARD)
    assert n0.grant_set() == frozenset()
    n1.recv(8, 0, GUARD)
    assert n1.get_action() == HigherNumber(8)
    assert n1.get_action() == Send(8, 0, GUARD_REPLY)
    assert n1.lease_cnt() == 1
    n0.recv(8, 1, GUARD_REPLY)
    assert n0.get_action() == Send(8, 1, PROMISE)
    assert n0.grant_set() == {1}
    n1.recv(8, 0, PROMISE)
    assert n1.get_action() == Send(8, 0, PROMISE_REPLY, held=True)
    assert n1.lease_cnt() == 2


def mutual_leases(mk, population=5, expire=1200, order_seed=0):
    """leaseman.rs:2173-2300: every replica grants to every other under number 7; the test's loop (take an action -> put the
    messages on the wire; take a message -> hand it to the manager) as a message pump with a seeded delivery order; every
    node must come to hold everyone's lease without one removal or timeout"""
    import random
    rnd = random.Random(order_seed)
    ck = Clock()
    nodes = [Node(mk(population, i, expire), ck) for i in range(population)]
    wire = []                                          # (src, dst, num, msg, held)
    for n in nodes:
        n.new_grants(7, None)
    full = set()
    for it in range(10000):
        for i, n in enumerate(nodes):
            while n.pending:
                a = n.pending.pop(0)
                if a[1] == "SendLeaseMsg":
                    wire.append((i, a[2], a[0], a[3], a[4]))
                elif a[1] == "BcastLeaseMsgs":
                    wire += [(i, p, a[0], a[3], None) for p in sorted(a[2]) if p != i]
                elif a[1] == "HigherNumber":
                    assert a[0] <= 7
                else:
                    assert a[1] in ("LeaseCleared", "NextRefresh", "GuardAcceptBar"), "removal or timeout happened on %d" % i
            if n.lease_cnt() == population:
                full.add(i)
        if len(full) == population:
            return it
        assert wire, "quiescent before every node held every lease"
        ck.sleep(1)
        src, dst, num, msg, held = wire.pop(rnd.randrange(len(wire)))
        nodes[dst].recv(num, src, msg, held=held)
    raise AssertionError("did not converge")


def beyond_the_reference_tests(mk):
    """the branches the reference's tests do not reach, hand-derived from the cited lines"""
    ck = Clock()
    n0, n1, n2 = Node(mk(3, 0, 600), ck), Node(mk(3, 1, 600), ck), Node(mk(3, 2, 600), ck)
    # accept_bar rides the Guard and comes out as GuardAcceptBar before the GuardReply (:526-541)
    n0.new_grants(3, None, accept_bar=41)
    assert n0.get_action() == HigherNumber(3)
    assert n0.get_action() == Bcast(3, {1, 2}, GUARD, bar=41)
    n1.recv(3, 0, GUARD, bar=41)
    assert n1.get_action() == HigherNumber(3)
    assert n1.get_action() == GuardAcceptBar(3, 0, 41)
    assert n1.get_action() == Send(3, 0, GUARD_REPLY)
    # a Promise nobody guarded for: PromiseReply { held: false } (:634-643), and the grantor drops the grant (:659-669)
    n0.recv(3, 2, GUARD_REPLY)
    assert n0.get_action() == Send(3, 2, PROMISE)
    assert n0.grant_set() == {2}
    n2.recv(3, 0, PROMISE)
    assert n2.get_action() == HigherNumber(3)
    assert n2.get_action() == Send(3, 0, PROMISE_REPLY, held=False)
    assert n2.lease_cnt() == 1
    n0.recv(3, 2, PROMISE_REPLY, held=False)
    assert n0.get_action() == GrantRemoved(3, 2, False)
    assert n0.grant_set() == frozenset()
    # finish 1's grant; a second NewGrants for everyone only guards those not already promised (:399-403)
    n0.recv(3, 1, GUARD_REPLY)
    assert n0.get_action() == Send(3, 1, PROMISE)
    n1.recv(3, 0, PROMISE)
    assert n1.get_action() == Send(3, 0, PROMISE_REPLY, held=True)
    n0.recv(3, 1, PROMISE_REPLY, held=True)
    assert n0.get_action() == NextRefresh(3, 1)
    n0.new_grants(3, None)
    assert n0.get_action() == Bcast(3, {2}, GUARD)
    # a second Guard while the promise is held is ignored (:512-516); a duplicate GuardReply too (:563-566)
    n1.recv(3, 0, GUARD)
    assert n1.no_action()
    n0.recv(3, 1, GUARD_REPLY)
    assert n0.no_action()
    # attempt_refresh only returns marked peers, once (:296-317)
    assert n0.attempt_refresh(None) == {1}
    assert n0.attempt_refresh(None) == frozenset()
    # DoRevoke for everyone: guards to 2 dropped, Revoke only to those promised (:456-480); with nothing promised: no action
    n0.do_revoke(3, None)
    assert n0.get_action() == Bcast(3, {1}, REVOKE)
    n0.recv(3, 2, GUARD_REPLY)                         # the guard is gone: ignored
    assert n0.no_action()
    # ClearHeld on the holder (:484-498), then the Revoke finds nothing held
    n1.clear_held(3)
    assert n1.get_action() == LeaseCleared(3)
    assert n1.lease_cnt() == 1
    n1.recv(3, 0, REVOKE)
    assert n1.get_action() == Send(3, 0, REVOKE_REPLY, held=False)
    n0.recv(3, 1, REVOKE_REPLY, held=False)
    assert n0.get_action() == GrantRemoved(3, 1, False)
    n0.do_revoke(3, None)
    assert n0.no_action()
    # outdated numbers: ignored, except a Revoke which is answered RevokeReply { held: false } under ITS number (:843-877)
    n1.recv(2, 0, GUARD)
    assert n1.no_action()
    n1.recv(2, 0, REVOKE)
    assert n1.get_action() == Send(2, 0, REVOKE_REPLY, held=False)
    assert n1.active_num() == 3
    # a new number on both sides
    n0.new_grants(4, {1})
    assert n0.get_action() == HigherNumber(4)
    assert n0.get_action() == Bcast(4, {1}, GUARD)
    n1.recv(4, 0, GUARD)
    assert n1.get_action() == HigherNumber(4)
    assert n1.get_action() == Send(4, 0, GUARD_REPLY)
    n0.recv(4, 1, GUARD_REPLY)
    assert n0.get_action() == Send(4, 1, PROMISE)
    n1.recv(4, 0, PROMISE)
    assert n1.get_action() == Send(4, 0, PROMISE_REPLY, held=True)
    assert n1.lease_cnt() == 2 and n1.lease_set() == {0}
    # a refresh late in the window extends from the deadline, not from now (timer.rs:94-115)
    n0.recv(4, 1, PROMISE_REPLY, held=True)            # deadline now + 600
    assert n0.get_action() == NextRefresh(4, 1)
    ck.sleep(500)
    assert n0.attempt_refresh({1}) == {1}              # deadline -> +1200 from the reply
    # a higher number drops what is held (:880-915)
    n1.recv(9, 2, PROMISE)
    assert n1.get_action() == HigherNumber(9)
    assert n1.get_action() == Send(9, 2, PROMISE_REPLY, held=False)
    assert n1.lease_cnt() == 1
    ck.sleep(650)                                      # +1150
    assert n0.grant_set() == {1}
    ck.sleep(100)
    assert n0.get_action() == GrantTimeout(4, 1)
    ck.sleep(5000)
    assert n1.no_action()                              # the old number's timers went with it


ALL_TRACES = [guard_expired, promise_expired, promise_refresh, revoke_replied, revoke_expired, regrant_higher,
              beyond_the_reference_tests]


class DeviceAdapter:
    """summerset_amd.leaseman.LeaseManager behind the array interface the traces (and the oracle) use"""

    def __init__(self, G, R, me, expire, hb=20):
        import torch
        from summerset_amd.leaseman import LeaseManager
        self.torch = torch
        self.m = LeaseManager(G, R, me, expire, hb)
        self.G, self.R, self.me = G, R, me
        self.dev = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")

    def _t(self, a, dt=np.int64):
        return self.torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(self.dev)

    def step_arrays(self, now, kind, num, peer, peers, msg, held, has_bar, bar):
        from summerset_amd.leaseman import pack_notice, unpack_actions
        u = lambda a: np.asarray(a).astype(np.int64)
        meta = pack_notice(u(kind), u(peer), u(peers), u(msg), u(held), u(has_bar))
        n, anum, ameta, abar = self.m.step(now, self._t(np.asarray(num).astype(np.uint64).view(np.int64)), self._t(meta),
                                           self._t(np.asarray(bar).astype(np.uint64).view(np.int64)))
        self.torch.cuda.synchronize() if self.dev.type == "cuda" else None
        n = n.cpu().numpy()
        live = np.arange(anum.shape[0])[:, None] < n[None, :]          # slots past act_n are not written
        m = np.where(live, ameta.cpu().numpy(), 0)
        out = {k: v.astype(np.uint8) for k, v in unpack_actions(m).items()}
        out["num"] = np.where(live, anum.cpu().numpy(), 0).view(np.uint64)
        out["bar"] = np.where(live, abar.cpu().numpy(), 0).view(np.uint64)
        return n, out

    def attempt_refresh_arrays(self, now, call, peers):
        o = self.m.attempt_refresh(now, self._t(call, np.uint8), self._t(peers, np.uint8))
        return o.cpu().numpy()

    def dump(self):
        return self.m.dump()


def random_differential(mk_dev, mk_orc, G=512, R=5, me=2, steps=120, seed=0, expire=600):
    """one manager, random notices (mostly plausible ones: messages under the current or the next number, grants,
    revokes, the odd stale number), random time steps around the deadlines; actions and state equal after every call"""
    rng = np.random.default_rng(seed)
    dev, orc = mk_dev(G, R, me, expire), mk_orc(G, R, me, expire)
    now = 5000
    num = np.full(G, 1, np.uint64)
    n_granted = n_held = 0
    for it in range(steps):
        now += int(rng.choice([0, 1, 7, 90, 250, 610, 1300], p=[.1, .2, .2, .2, .15, .1, .05]))
        kind = rng.choice([N_NONE, N_NEW_GRANTS, N_DO_REVOKE, N_CLEAR_HELD, N_RECV_MSG], size=G, p=[.1, .15, .07, .03, .65]).astype(np.uint8)
        bump = rng.random(G)
        num = num + (bump < 0.03).astype(np.uint64)
        nn = np.where(bump > 0.97, np.maximum(num, 1) - 1, num).astype(np.uint64)      # now and then a stale number
        peer = rng.integers(0, R + 1, G).astype(np.uint8)                               # R = out of range, me = myself: both ignored
        peers = np.where(rng.random(G) < 0.3, ALL, rng.integers(0, 1 << R, G)).astype(np.uint8)
        msg = rng.integers(0, 6, G).astype(np.uint8)
        held = (rng.random(G) < 0.8).astype(np.uint8)
        has_bar = (rng.random(G) < 0.3).astype(np.uint8)
        bar = rng.integers(0, 1 << 40, G).astype(np.uint64)
        a = dev.step_arrays(now, kind, nn, peer, peers, msg, held, has_bar, bar)
        b = orc.step_arrays(now, kind, nn, peer, peers, msg, held, has_bar, bar)
        assert (a[0] == b[0]).all(), (it, np.nonzero(a[0] != b[0])[0][:5])
        for k in b[1]:
            assert (a[1][k] == b[1][k]).all(), (it, k, np.nonzero(a[1][k] != b[1][k]))
        if it % 3 == 0:
            call = (rng.random(G) < 0.5).astype(np.uint8)
            pp = np.where(rng.random(G) < 0.5, ALL, rng.integers(0, 1 << R, G)).astype(np.uint8)
            assert (dev.attempt_refresh_arrays(now, call, pp) == orc.attempt_refresh_arrays(now, call, pp)).all(), it
        da, db = dev.dump(), orc.dump()
        for k in db:
            assert (da[k] == db[k]).all(), (it, k)
        n_granted += int(db["grant_set"].astype(bool).sum())
        n_held += int(db["lease_set"].astype(bool).sum())
    return n_granted, n_held
