"""`smr_rsp_pstore_emit_accepts` (csrc/rsp_payload.hip): the Accept frames of an RSPaxos leader, shard payload included, written on
the device out of the payload store -- against the host encoder's bytes for the oracle's codewords.  Sorts LAST like the other
first-run device tests: it was written when the round's GPU budget was spent (tests/test_hostsim.py runs it on the kernel-source
emulator), so its first device run is the driver's at round end, behind everything else."""
import numpy as np
import pytest

from summerset_amd.workloads import payload_batch_bytes as batch_bytes, payload_batch_len as batch_len

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300), pytest.mark.stage(9)]


def test_accept_frames_with_their_payload_are_the_host_encoder_s(cuda, oracle):
    """`smr_rsp_pstore_emit_accepts`: the Accept frames a leader sends, shard bytes included, written by a kernel straight out of
    the store -- byte for byte what the host encoder (`smr_wire_rsp_accept` around `smr_wire_rscodeword`, the reference's frame:
    safetcp.rs:127-132, rspaxos/mod.rs:262-270, rscoding.rs:43-77) writes from the ORACLE's codeword: ragged batch lengths (1-,
    3-byte varints), ballots up to 2^40 (9-byte varints), every subset of shards, groups with nothing to send, a slot too short.
    (Written when the round's GPU budget was spent: tests/test_hostsim.py runs it on the kernel-source emulator; its first device
    run is the driver's at round end.)"""
    import torch
    from summerset_amd import RSPaxosPayloadStore, RSPaxosReplicaGroup, wire
    from summerset_amd.rsp_payload import REQS, VOTED
    G, R, W, L = 96, 5, 8, 700
    rep, st = RSPaxosReplicaGroup(G, R, me=0, window=W), RSPaxosPayloadStore(G, R, W, max_data_len=L)
    rep.preset_leader(0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    rng = np.random.default_rng(11)
    tok = (5 + 37 * np.arange(G)).astype(np.uint32)
    lens = batch_len(tok, L)
    lens[:4] = (1, 2, 3, L)                                      # shard lengths 1, 1, 1 and the longest
    acc = rep.req_batch(t(tok.view(np.int32)))
    data = batch_bytes(tok, L)
    st.put(acc, t(data), t(lens.view(np.int32)))
    st.follow(rep)
    slot = np.zeros(G, np.uint32); slot[9] = 0xFFFFFFFF
    ballot = rng.choice(np.array([7, 250, 251, 65535, 65536, 2 ** 32 - 1, 2 ** 32, 2 ** 40], np.uint64), G)
    mask = (np.arange(G) % 32).astype(np.uint8)                  # every subset (0: nothing to send)
    flags = np.ones(G, np.uint8); flags[11] = 0
    stride = 64 + 11 * R + R * 240
    frames, ln = st.emit_accepts(t(slot.view(np.int32)), t(ballot.view(np.int64)), t(mask), stride, REQS, t(flags))
    frames, ln = frames.cpu().numpy(), ln.cpu().numpy()
    n_sent = 0
    for g in range(G):
        if not (flags[g] and slot[g] == 0 and mask[g]):
            assert ln[g] == 0, g
            continue
        d = data[g, :lens[g]]
        sl = oracle.rs_shard_len(d.size, 3)
        cw = np.zeros((R, sl), np.uint8)
        cw[:3].reshape(-1)[:d.size] = d
        cw[3:] = oracle.rs_encode(3, 2, d)
        want = wire.rsp_accept(0, int(ballot[g]), wire.rscodeword(3, 2, int(d.size), [cw[k].tobytes() if (mask[g] >> k) & 1 else None for k in range(R)]))
        assert ln[g] == len(want) and frames[g, :ln[g]].tobytes() == want, (g, int(ln[g]), len(want))
        n_sent += 1
    assert n_sent > 80
    # the leader's own vote out of the VOTED plane: shard 0 alone, whatever the mask asks for
    frames, ln = st.emit_accepts(t(np.zeros(G, np.int32)), t(ballot.view(np.int64)), t(np.full(G, 31, np.uint8)), stride, VOTED)
    frames, ln = frames.cpu().numpy(), ln.cpu().numpy()
    for g in (0, 3, 50):
        d = data[g, :lens[g]]
        sl = oracle.rs_shard_len(d.size, 3)
        x = np.zeros(3 * sl, np.uint8); x[:d.size] = d
        want = wire.rsp_accept(0, int(ballot[g]), wire.rscodeword(3, 2, int(d.size), [x[:sl].tobytes(), None, None, None, None]))
        assert frames[g, :ln[g]].tobytes() == want, g
    # a slot too short for the frame is said, not overrun
    short, ln = st.emit_accepts(t(np.zeros(G, np.int32)), t(ballot.view(np.int64)), t(np.full(G, 31, np.uint8)), 64, REQS)
    ln = ln.cpu().numpy()
    assert (ln[lens > 30] == -1).all() and not short.cpu().numpy()[lens > 30].any()
