"""Generates tests/golden/*.npz.

The reference (Rust) cannot be built or imported here, and its own tests hold no
byte-level vectors for this path, so these fixtures are produced
by the KAT-pinned CPU oracle (oracle/rs_oracle.c, oracle/mp_oracle.c; late_golden.npz: raft_oracle.c's CRaft
variant, qr_oracle.c) and frozen:
they pin BOTH the oracle and the HIP path against silent drift.  Re-run only on a
deliberate semantic change:   python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402
from summerset_amd import stream  # noqa: E402

ALNUM = np.frombuffer(b"0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz", np.uint8)


def alnum(seed, n):
    h = stream.splitmix64(np.arange(n, dtype=np.uint64) ^ np.uint64(seed))
    return ALNUM[(h % np.uint64(62)).astype(np.int64)]


def rs_fixture():
    out = {}
    # rse_bench-equivalent: bincode(String of 4096 alnum bytes) -> L = 4099, shard_len 1367
    val = alnum(0x5EED5EED, 4096)
    ser = O.bincode_string(val.tobytes())
    out["bench4k_data"] = ser
    out["bench4k_parity"] = O.rs_encode(3, 2, ser)
    # RSPaxos ReqBatch with one Put (client 7, req 42, key k0000003)
    rb = O.bincode_reqbatch_put(7, 42, b"k0000003", val.tobytes())
    out["reqbatch_data"] = rb
    out["reqbatch_parity"] = O.rs_encode(3, 2, rb)
    # the reference test's "interesting_value" TestData(String)
    iv = O.bincode_string(b"interesting_value")
    out["iv_data"] = iv
    out["iv_parity"] = O.rs_encode(3, 2, iv)
    # ragged lengths 1..48, scheme (3,2); and other schemes on one length
    for L in (1, 2, 3, 4, 15, 16, 17, 47, 48, 49):
        d = alnum(L, L)
        out["len%d_data" % L] = d
        out["len%d_parity" % L] = O.rs_encode(3, 2, d)
    for (d, p) in ((6, 4), (9, 6), (12, 8), (5, 5)):
        x = alnum(d * 100 + p, 1000)
        out["s%d_%d_data" % (d, p)] = x
        out["s%d_%d_parity" % (d, p)] = O.rs_encode(d, p, x)
    out["matrix_3_2"] = O.rs_matrix(3, 2)
    out["matrix_5_5"] = O.rs_matrix(5, 5)
    np.savez_compressed(os.path.join(HERE, "rs_golden.npz"), **out)


def mp_fixture():
    """Final state of a small MultiPaxos run with drops + a leader change everywhere."""
    G, R, S, W, T = 48, 5, 2, 32, 40
    cap = W + 4
    m = O.MpOracle(G, R, W, cap=cap)
    m.preset_leader(0)
    st = stream.MultiPaxosStream(G, R, S, cap=cap, n_ticks=T, drop_p=0.1, timeout_frac=0.5, hb_every=4)
    for t in range(T):
        m.tick(**st.tick(t))
    out = {"params": np.array([G, R, S, W, T], np.int64)}
    for r in range(R):
        d = m.dump(r)
        for k, v in d.items():
            out["r%d_%s" % (r, k)] = v
        out["r%d_commits" % r] = np.array([m.total_commits(r)], np.uint64)
    np.savez_compressed(os.path.join(HERE, "mp_golden.npz"), **out)


def cluster_states(oracle):
    """final states of the five oracles of the closed-loop EPaxos (with execution, 10 % lost PreAccepts) and RSPaxos
    (f = 1, 10 % loss) clusters of tests/test_oracle_{ep,rsp}_cluster.py ("ep_*", "rsp_*")"""
    sys.path.insert(0, os.path.dirname(HERE))
    import rsp_scenarios as sc
    import test_oracle_ep_cluster as epc
    import test_oracle_rsp_cluster as rspc
    out = {}
    reps, _ = epc._run(oracle, 48, 20, 6, seed=41, drop_p=0.1, execute=True)
    for r, rep in enumerate(reps):
        for k, v in rep.dump().items():
            if isinstance(v, np.ndarray):
                out["ep_r%d_%s" % (r, k)] = v
        for k, v in rep.exec_dump().items():
            if isinstance(v, np.ndarray):
                out["ep_r%d_x_%s" % (r, k)] = v
    reps = rspc._cluster(oracle, 40, 1)
    sc.run(reps, 40, 24, seed=8, loss=0.1)
    for r, rep in enumerate(reps):
        for k, v in rep.dump().items():
            if isinstance(v, np.ndarray):
                out["rsp_r%d_%s" % (r, k)] = v
    return out


def late_fixture():
    """Final states of the frozen runs of the engines that came after the MultiPaxos cluster: the CRaft leader
    (oracle/raft_oracle.c, orc_craft_*) and the quorum reads of five replicas (oracle/qr_oracle.c), on the seeded streams
    of tests/test_zz_{craft,qread}_gpu.py -- the device tests hold the engines equal to the oracles after every call, the
    CPU tests hold the oracles' final states equal to this file."""
    sys.path.insert(0, os.path.dirname(HERE))
    import test_zz_craft_gpu as craft
    import test_zz_qread_gpu as qread
    out = {}
    for k, v in craft._run(None, O, **craft.GOLDEN_RUN).items():
        out["craft_" + k] = v
    for k, v in qread._run(None, O, **qread.GOLDEN_RUN).items():
        out["qr_" + k] = v
    out.update(cluster_states(O))
    np.savez_compressed(os.path.join(HERE, "late_golden.npz"), **out)


if __name__ == "__main__":
    if "--late-only" not in sys.argv:
        rs_fixture()
        mp_fixture()
    late_fixture()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
