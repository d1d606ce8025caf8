"""The engine's kernels -- the shipped .hip sources, compiled for the host by tests/hostsim, every lane a
fiber, cross-lane operations meeting per wavefront / block -- against the CPU oracle, through the shipped
C-ABI and Python mirror.  This reruns scenarios of tests/test_*_gpu.py (at smaller shapes) where no GPU
is at hand: it checks the kernels' logic, not their behaviour on the device (the gpu-marked tests do
that)."""
import os

import pytest


@pytest.fixture(scope="module")
def sim():
    import hostsim
    hostsim.build()
    return hostsim


def test_emulation_selftest(sim):
    """tests/hostsim/selftest.cpp: wave-aggregated appends from divergent callers, counter flushes, lock-step
    memory order inside a wavefront, the LDS hand-off across a block, a loop that lanes leave early -- and
    an access into an arena guard gap must abort"""
    import subprocess
    exe = sim.build_selftest()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.stdout, r.stderr)
    r = subprocess.run([exe, "guard"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "arena guard gap" in r.stderr and "guard not hit" not in r.stdout, (r.stdout, r.stderr)


def test_sim_library_is_not_the_product(sim, engine_lib):
    """the package's own handle is the HIP library; the simulated one is only ever swapped in by a test"""
    from summerset_amd import _lib
    assert _lib.load() is engine_lib
    with sim.patched() as s:
        assert _lib.load() is s and s is not engine_lib
    assert _lib.load() is engine_lib


def test_epaxos_kernels_on_the_host(sim, oracle):
    import test_ep_gpu as t
    with sim.patched():
        t.test_epaxos_handlers_match_oracle("cpu", oracle, 300, 32, 0)
        t.test_epaxos_handlers_match_oracle("cpu", oracle, 257, 16, 3)
        t.test_closed_loop_cluster_matches_oracle("cpu", oracle)


def test_raft_kernels_on_the_host(sim, oracle):
    import test_raft_gpu as t
    with sim.patched():
        t.test_raft_steady("cpu", oracle)
        t._run_batched("cpu", oracle, G=300, R=5, W=64, T=25, batches=(1, 5, 16, 3))      # smr_raft_leader_run_ticks
        t._run_batched("cpu", oracle, G=200, R=3, W=32, T=24, batches=(7, 17), higher_p=0.004)
        t._run_batched("cpu", oracle, G=130, R=7, W=16, T=20, batches=(20,), n_new_max=9)
        t._run_batched("cpu", oracle, G=130, R=5, W=64, T=8, batches=(8,), quiet={("append", 2), ("replies", 4), ("append", 7), ("replies", 7)})
        t.test_craft_leader_refuses_batches("cpu")
        t.test_raft_three_replicas_and_stepdown("cpu", oracle)
        t.test_craft_threshold("cpu", oracle)
        t.test_follower_and_elections_match_oracle("cpu", oracle, 300, 64)
        t.test_closed_loop_cluster_matches_oracle("cpu", oracle)
        assert t.run_one_launch_replication("cpu", oracle, G=200, W=64, K=8, T=12) > 300     # smr_raft_cluster_replicate == the 2 n calls == the oracle


def test_raft_one_launch_tick_on_the_host(sim, oracle):
    import test_raft_gpu as t
    with sim.patched():
        assert t.run_one_launch_tick("cpu", oracle, G=200, W=64, K=8, T=12) > 300            # smr_raft_cluster_tick == the 2 + 2 n calls == the oracle
        assert t.run_one_launch_tick("cpu", oracle, G=70, W=64, K=8, T=8, R=7) > 100            # ... and with seven replicas (the R > 5 instantiation)


def test_craft_leader_kernels_on_the_host(sim, oracle):
    """the CRaft leader variant (a15): reply kernel with the fork's rules, heartbeat tick with the reply counters and the
    full-copy fall-back, mode switches, shard assignment + the RS kernels"""
    import test_zz_craft_gpu as t
    with sim.patched():
        t.test_craft_leader_fallback_and_commit_rule("cpu", oracle)
        t.test_craft_leader_step_down_and_other_populations("cpu", oracle)
        t.test_craft_entry_shards_follow_the_assignment("cpu", oracle)
        t.test_final_state_is_the_golden_one("cpu", oracle)


def test_quorum_read_kernels_on_the_host(sim, oracle):
    """MultiPaxos near quorum reads (f.4): highest-slot table, responder, the issuer's read-quorum tally"""
    import test_zz_qread_gpu as t
    with sim.patched():
        t.test_quorum_reads_match_oracle("cpu", oracle)
        t.test_quorum_reads_other_shapes("cpu", oracle)
        t.test_responder_reads_the_multipaxos_engines_log_in_place("cpu", oracle)
        t.test_final_state_is_the_golden_one("cpu", oracle)


def test_kv_state_machine_kernel_on_the_host(sim, oracle):
    """the device KV executor (f.3) against the reference's state-machine tests and a dict, and under a stable leader's reads"""
    import test_zz_kv_gpu as t
    with sim.patched():
        t.test_reference_state_machine_tests("cpu")
        t.test_put_rand_get_rand_per_group("cpu")
        t.test_stable_leader_reads_the_executed_state("cpu", oracle)


def test_epaxos_execution_kernel_on_the_host(sim, oracle):
    import test_zz_ep_exec_gpu as t
    with sim.patched():
        t.test_execution_traces("cpu", oracle)
        t.test_handler_streams_with_execution("cpu", oracle, 300, 32, 0)
        t.test_handler_streams_with_execution("cpu", oracle, 257, 16, 3)
        t.test_closed_loop_cluster_with_execution("cpu", oracle, 0.0)
        t.test_closed_loop_cluster_with_execution("cpu", oracle, 0.15)


def test_multipaxos_kernels_on_the_host(sim, oracle):
    """steady state, ack loss, leader changes on every group (Prepare phase, wave-cooperative jobs, the
    quorum tally with its LDS hand-off), three replicas, RSPaxos threshold, window back-pressure"""
    import test_mp_gpu as t
    with sim.patched() as lib:
        t._run("cpu", oracle, G=70, R=5, S=1, W=32, n_ticks=24, drop_p=0.0, timeout_frac=0.0, hb_every=4, preset=True)
        t._run("cpu", oracle, G=130, R=5, S=4, W=64, n_ticks=30, drop_p=0.1, timeout_frac=0.0, hb_every=3, preset=True)
        for sticks in (0, 1):
            t._run("cpu", oracle, G=200, R=5, S=2, W=64, n_ticks=40, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True,
                   straggler_ticks=sticks)
        t._run("cpu", oracle, G=96, R=5, S=2, W=64, n_ticks=30, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True,
               per_round=True)
        t._run("cpu", oracle, G=64, R=5, S=1, W=64, n_ticks=24, drop_p=0.05, timeout_frac=0.3, hb_every=4, preset=False)
        t._run("cpu", oracle, G=65, R=3, S=2, W=32, n_ticks=30, drop_p=0.2, timeout_frac=0.5, hb_every=2, preset=True)
        t._run("cpu", oracle, G=100, R=5, S=2, W=64, n_ticks=30, drop_p=0.15, timeout_frac=0.0, hb_every=4, preset=True,
               commit_extra=1)
        eng, _ = t._run("cpu", oracle, G=64, R=5, S=3, W=16, n_ticks=40, drop_p=0.0, timeout_frac=0.0, hb_every=8, preset=True)
        # round 5: R2's bulk launch as fast path + rest (SMR_MP_SPLIT_R2; off by default), a short ttl so that the rest launch has work
        os.environ["SMR_MP_SPLIT_R2"] = "1"
        try:
            t._run("cpu", oracle, G=130, R=5, S=2, W=64, n_ticks=32, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, fused=8, straggler_ticks=1)
            t._run("cpu", oracle, G=96, R=3, S=2, W=32, n_ticks=30, drop_p=0.2, timeout_frac=0.5, hb_every=2, preset=True, fused=5, straggler_ticks=1)
        finally:
            del os.environ["SMR_MP_SPLIT_R2"]
        # a window that overflows while the rest of R3 rides in the next R1 launch (quiet stretch, batches of 8)
        e2, _ = t._run("cpu", oracle, G=70, R=5, S=4, W=16, n_ticks=72, drop_p=0.05, timeout_frac=0.0, hb_every=12, preset=True, fused=8,
                       straggler_ticks=4, every=8, no_array_when_quiet=True)
        assert e2.counters(0)["rejects"] > 0
        assert eng.counters(0)["rejects"] > 0
        # bench.py's shape (S = 32, W = 512, no commit list: the tally's closed form; long outboxes after a leader change),
        # with and without the straggler side launch
        for sticks in (0, 4):
            t._run_bench_shape("cpu", oracle, G=130, frac=0.25, span=8, n_ticks=20, straggler_ticks=sticks, every=4)
        # role rotation (smr_mp_set_role_rotation): rows of the bulk launches by role, same results
        t._run("cpu", oracle, G=130, R=5, S=2, W=64, n_ticks=30, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, straggler_ticks=2, rotate=True)
        t._run("cpu", oracle, G=96, R=3, S=2, W=32, n_ticks=30, drop_p=0.2, timeout_frac=0.5, hb_every=2, preset=True, fused=5, straggler_ticks=4, rotate=True)
        t._run("cpu", oracle, G=70, R=7, S=2, W=64, n_ticks=24, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, fused=6, straggler_ticks=8, rotate=True)
        t._run_bench_shape("cpu", oracle, G=130, frac=0.25, span=8, n_ticks=20, straggler_ticks=4, every=4, fused=8, rotate=True)
        # more client batches per tick than R1 prefetches into registers (R1_PF = 32)
        t._run("cpu", oracle, G=64, R=5, S=40, W=256, n_ticks=12, drop_p=0.05, timeout_frac=0.0, hb_every=4, preset=True)
        # batches of ticks with the straggler list on: the list's groups run the whole batch in one launch
        t._run("cpu", oracle, G=200, R=5, S=2, W=64, n_ticks=40, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, fused=7, straggler_ticks=3)
        t._run("cpu", oracle, G=70, R=8, S=1, W=32, n_ticks=30, drop_p=0.2, timeout_frac=1.0, hb_every=2, preset=True, fused=16, straggler_ticks=8)
        t._run_bench_shape("cpu", oracle, G=130, frac=0.25, span=8, n_ticks=20, straggler_ticks=4, every=4, fused=4)
        t.test_batches_and_single_ticks_share_the_list("cpu", oracle)
        t.run_rest_rides_in_next_r1("cpu", oracle, G=130, S=3, W=64, n_ticks=64, drop_p=0.3)   # mp_rest_then_local (round 4)
        # round 6: the next tick's steady-state appends in the tally launch (MpNextLocal)
        assert t.run_next_appends_ride_in_the_tally("cpu", oracle, G=130, S=32, W=512, n_ticks=18, frac=0.25) > 0
        t.run_next_appends_ride_in_the_tally("cpu", oracle, G=70, S=40, W=512, n_ticks=10, frac=0.1)
        t.run_next_appends_ride_in_the_tally("cpu", oracle, G=96, S=6, W=32, n_ticks=40, frac=0.1, H=12, win_reserve=2, expect_rejects=True)
        t.run_next_appends_ride_in_the_tally("cpu", oracle, G=70, S=3, W=64, n_ticks=24, frac=0.4, R=3, max_drop=1, batch=5, H=3)
        t.run_next_appends_ride_in_the_tally("cpu", oracle, G=70, S=3, W=64, n_ticks=24, frac=0.3, rotate=True, batch=16)
        t.run_next_appends_ride_in_the_tally("cpu", oracle, G=70, S=2, W=64, n_ticks=20, frac=0.2, R=7, max_drop=3)
        t.run_next_appends_ride_in_the_tally("cpu", oracle, G=130, S=3, W=64, n_ticks=32, frac=0.2, drop_p=0.3, max_drop=None)
        # populations the device tests do not run (the 8-replica template instances of the tally and the reply kernels)
        t._run("cpu", oracle, G=100, R=7, S=2, W=64, n_ticks=30, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True)
        t._run("cpu", oracle, G=100, R=4, S=3, W=64, n_ticks=30, drop_p=0.1, timeout_frac=1.0, hb_every=3, preset=True)
        t._run("cpu", oracle, G=70, R=8, S=1, W=32, n_ticks=30, drop_p=0.2, timeout_frac=1.0, hb_every=2, preset=True)
        # the straggler launch keeps a group's replicas on the lanes of one wavefront: eight of them
        t._run("cpu", oracle, G=70, R=8, S=1, W=32, n_ticks=30, drop_p=0.2, timeout_frac=1.0, hb_every=2, preset=True, straggler_ticks=3)
        t._run("cpu", oracle, G=100, R=7, S=2, W=64, n_ticks=30, drop_p=0.15, timeout_frac=0.0, hb_every=4, preset=True, commit_extra=2)


def test_wire_ingest_kernels_on_the_host(sim, oracle):
    """peer traffic parsed on the device (f.1): streams against the sequential decoder, and inside the engine's tick"""
    import test_zz_wire_ingest_gpu as t
    with sim.patched():
        t.test_ingest_matches_the_sequential_decoder("cpu")
        t.test_ingest_against_what_the_test_wrote("cpu")                       # no product codec in the loop
        t.test_empty_and_overfull("cpu")
        t.test_accept_replies_over_the_wire("cpu", oracle)
        import test_zzz_wire_ingest_edges_gpu as te                             # the laid-out edges: fast-path boundary, window ends, extremes
        te.test_payload_lengths_around_the_register_fast_path("cpu")
        te.test_every_frame_of_the_straight_line_path("cpu")
        te.test_frames_at_the_window_edges("cpu")
        te.test_frames_around_the_ring_of_two_lines("cpu")
        te.test_extreme_values_and_odd_encodings("cpu")
        te.test_ragged_wavefront_and_unaligned_buffer_end("cpu")
        te.test_frames_laid_out_by_hand_give_records_laid_out_by_hand("cpu")     # expected records written out by the test


def test_wire_ingest_one_pass_segments_on_the_host(sim, oracle):
    """the same traffic parsed in ONE pass into a segment per connection (smr_wire_ingest_mp_conn), and smr_mp_deliver_acks_conn"""
    import test_zz_wire_ingest_conn_gpu as t
    with sim.patched():
        t.test_segments_are_the_sequential_decoders_lists("cpu")
        t.test_a_full_segment_stops_the_connection_in_front_of_the_frame("cpu")
        import test_zzzz_wire_conn_edges_gpu as t2                             # (no device run yet: sorted last there)
        t2.test_the_dense_calls_edge_streams_through_the_one_pass_call("cpu")
        t2.test_frames_laid_out_by_hand_give_segments_laid_out_by_hand("cpu")     # raw output arrays written out by the test
        t.test_edges("cpu")
        t.test_accept_replies_over_the_wire_in_segments("cpu", oracle)


def test_reply_ingest_kernels_on_the_host(sim, oracle):
    """Raft AppendEntriesReply / EPaxos PreAcceptReply frames parsed into the engines' [R][G] arrays (f.1, round 3): frame by
    frame against what the test wrote, and inside the closed-loop clusters against the oracles"""
    import test_zz_reply_ingest_gpu as t
    with sim.patched():
        t.test_raft_replies_frame_by_frame("cpu")
        t.test_ep_pre_accept_replies_frame_by_frame("cpu")
        t.test_reply_ingest_argument_edges("cpu")
        t.test_raft_cluster_replies_over_the_wire("cpu", oracle)
        t.test_ep_cluster_pre_accept_replies_over_the_wire("cpu", oracle)


def test_fused_raft_wire_replies_on_the_host(sim, oracle):
    """smr_raft_leader_handle_wire_replies (the parse as the prologue of the leader's reply handler, one launch) against the two
    calls and the oracle"""
    import test_zz_reply_ingest_gpu as t
    with sim.patched():
        assert t.run_fused_raft_wire_replies("cpu", oracle, G=300, T=5) > 0
        assert t.run_fused_raft_wire_replies("cpu", oracle, G=1100, R=3, me=0, seed=5, T=3) > 0
        assert t.run_fused_raft_wire_replies("cpu", oracle, G=400, R=7, W=32, me=6, seed=6, T=3) > 0
        t.test_fused_raft_wire_replies_refuse_sparse_connections("cpu")


def test_wire_emit_kernels_on_the_host(sim, oracle):
    """reply frames written by kernels (f.1, the send half): byte for byte what the test lays out, back through the ingest
    kernels, and as the senders of the Raft / EPaxos closed loops"""
    import test_zz_wire_emit_gpu as t
    with sim.patched():
        t.test_mp_accept_reply_frames("cpu")
        t.test_raft_reply_frames("cpu")
        for R in (3, 5, 7):
            t.test_ep_pre_accept_reply_frames("cpu", R)
        t.test_rsp_accept_replies_ingest("cpu")
        t.test_raft_cluster_with_emitted_replies("cpu", oracle)
        t.test_ep_cluster_with_emitted_replies("cpu", oracle)


def test_rs_kernels_on_the_host(sim, oracle):
    import test_rs_gpu as t
    with sim.patched():
        for lut in (False, True):
            t.test_golden_vectors("cpu", oracle, lut)
        for L in (1, 2, 3, 5, 16, 31, 48, 97, 1000, 4099, 4113):
            t.test_encode_matches_oracle_ragged("cpu", oracle, L)
        for scheme in ((3, 2), (6, 4), (12, 8), (5, 5), (4, 1), (1, 1)):
            t.test_other_schemes("cpu", oracle, scheme)
        t.test_all_erasure_patterns_rs32("cpu", oracle)
        t.test_error_cases_mirror_reference("cpu")
        t.test_verify_detects_corruption("cpu")
        t.test_padding_bytes_are_never_read("cpu", oracle)
        t.test_subset_copy_and_absorb_other_rspaxos_flow("cpu", oracle)
        t.test_from_data_and_encode_into_shard_stores("cpu", oracle)            # every shard written once, shard-major (round 4)
        for scheme, L in (((3, 2), 1), ((3, 2), 2), ((3, 2), 47), ((3, 2), 4099), ((6, 4), 777), ((1, 1), 33), ((12, 8), 1000), ((3, 0), 100)):
            t.test_from_data_and_encode_one_pass("cpu", oracle, scheme, L)
        t.test_from_data_and_encode_fans_the_shards_out("cpu", oracle)


def test_rspaxos_kernels_on_the_host(sim, oracle):
    import test_zz_rsp_gpu as t
    with sim.patched():
        t.test_traces_on_the_engine("cpu", oracle)
        t.test_closed_loop_cluster_matches_oracle("cpu", oracle, 130, 32, 0, 0.0)
        t.test_closed_loop_cluster_matches_oracle("cpu", oracle, 130, 32, 1, 0.1)
        t.test_closed_loop_cluster_matches_oracle("cpu", oracle, 70, 16, 1, 0.05)
        t.test_random_handler_calls_match_oracle("cpu", oracle, 150, 8, 0, 0)
        t.test_random_handler_calls_match_oracle("cpu", oracle, 150, 16, 2, 1)
        t.test_random_handler_calls_match_oracle("cpu", oracle, 500, 32, 4, 2)


def test_rspaxos_device_steady_loop_on_the_host(sim, oracle):
    """summerset_amd/rsp_cluster.SteadyLoop (what the config-4 bench leg times) == the numpy-staged closed loop on oracles"""
    import test_zz_rsp_steady_gpu as t
    with sim.patched():
        assert t.run_steady("cpu", oracle, 200, 16, 1, 0.1) > 0
        assert t.run_steady("cpu", oracle, 130, 8, 0, 0.2, T=10) > 0
        assert t.run_steady("cpu", oracle, 90, 16, 1, 0.05, T=5, with_cw=True) > 0
        assert t.run_steady("cpu", oracle, 200, 16, 1, 0.1, one_launch=True) > 0       # smr_rsp_cluster_steady_tick
        assert t.run_steady("cpu", oracle, 130, 8, 0, 0.2, T=10, one_launch=True) > 0
        assert t.run_steady("cpu", oracle, 65, 16, 1, 0.3, T=9, one_launch=True) > 0
        t.test_one_launch_cluster_argument_errors("cpu")


def test_spread_rspaxos_exchange_on_the_host(sim, oracle):
    """layout L2 of the RSPaxos engine with every rank in this process (tests/test_spread_rsp.py): against the co-located steady loop"""
    import test_spread_rsp as t
    with sim.patched():
        t.run_spread_vs_colocated("cpu", 2, 130, T=7)
        t.run_spread_vs_colocated("cpu", 3, 100, T=6)
        t.run_spread_vs_colocated("cpu", 8, 170, T=5, loss=0.0)
        t.run_spread_vs_colocated("cpu", 2, 90, T=7, payload=True, oracle=oracle)     # ... with the bytes in payload stores
        t.run_spread_vs_colocated("cpu", 3, 80, T=5, loss=0.0, payload=True, oracle=oracle)


def test_rspaxos_masks_and_rs_bytes_end_to_end(sim, oracle):
    import test_zz_rsp_bytes_gpu as t
    with sim.patched():
        t.test_tokens_are_real_shard_bytes("cpu", oracle)


def test_rspaxos_payload_store_on_the_host(sim, oracle):
    """the shard bytes behind the replicas' masks (csrc/rsp_payload.hip) through leader changes, against the oracle's codewords"""
    import test_zz_rsp_payload_gpu as t
    with sim.patched():
        tot, n_exec, n_cmp = t.run_closed_loop("cpu", oracle, 40, 16, 1, 0.1, 77, T=15)
        assert tot["rebuilt"] > 0 and n_exec > 0
        t.run_closed_loop("cpu", oracle, 24, 8, 0, 0.0, 20, T=12)
        t.test_steady_tick_is_one_put_and_one_shard_per_follower("cpu", oracle)
        t.test_rows_are_shard_major_batches_the_rs_kernels_accept("cpu", oracle)
        t.test_extract_and_ingest_round_trip("cpu", oracle)
        import test_zzzz_rsp_emit_accepts_gpu as te
        te.test_accept_frames_with_their_payload_are_the_host_encoder_s("cpu", oracle)   # Accept frames with their payload == the host encoder's
        t.test_one_call_with_two_senders_takes_each_group_s_shard_from_its_own_sender("cpu", oracle)
        tot, n_exec, n_cmp = t.run_closed_loop("cpu", oracle, 40, 16, 1, 0.1, 77, T=15, staging=True)    # bytes travel as messages only
        assert tot["rebuilt"] > 0 and n_exec > 0
        n_cmp, c = t.run_random_calls("cpu", oracle, 60, 8, 0, 0, steps=60)                               # adversarial calls: never a shard invented
        assert n_cmp > 500 and c["copied"] > 0 and c["aliases"] > 0 and c["moved_out"] > 0, c    # (votes as aliases; some left the reqs row)
        t.test_argument_errors("cpu")


def test_accept_reply_records_on_the_host(sim, oracle):
    """the AcceptReply record <-> ack matrix kernels (smr_mp_collect_acks / smr_mp_deliver_acks) under the emulator"""
    import test_mp_gpu as t
    with sim.patched():
        t.test_accept_replies_as_records("cpu", oracle)


def test_baseline_config_slice_tests_on_the_host(sim, oracle):
    """tests/test_baseline_configs_gpu.py at small shapes: the whole population on the (emulated) engine, slices of it on
    oracles started at the slice's group offset -- incl. `smr_mp_dump_range` and the handlers being no-ops for groups
    whose flag is clear"""
    import test_baseline_configs_gpu as t
    with sim.patched():
        changed, total, _, _ = t.run_multipaxos_slices("cpu", oracle, G=320, S=4, W=64, n_ticks=14, frac=0.3, span=6, width=64, n_slices=3, every=3)
        assert changed > 0 and total > 0
        # the launch shape bench.py times: batches of 8 through smr_mp_run_ticks with the straggler list on
        changed, total, want, cap = t.run_multipaxos_slices("cpu", oracle, G=320, S=4, W=64, n_ticks=20, frac=0.3, span=10, width=64,
                                                            n_slices=3, every=1, straggler_ticks=4, batch=8)
        assert changed > 0 and total > 0 and 0 < want <= cap
        t.run_rspaxos_slices("cpu", oracle, G=256, W=16, T=12, ft=1, loss=0.05, width=64, n_slices=3)
        assert t.run_rspaxos_one_launch("cpu", oracle, G=200, W=16, L=133, T=9, ft=1) > 0   # config 4's timed launches
        assert t.run_rspaxos_payload("cpu", oracle, G=130, W=8, L=133, T=11, ft=1) > 0      # ... with the bytes in the payload store
        assert t.run_craft_payload("cpu", oracle, G=200, W=16, L=67, T=9, width=64, n_slices=3) > 0   # the `craft_payload` leg's launches
        t.run_epaxos_slices("cpu", oracle, G=256, W=16, K=8, T=5, width=64, n_slices=3)
        for pm in (False, True):                                # the one-launch cluster tick against oracle slices, both orders
            t.run_epaxos_cluster_slices("cpu", oracle, G=256, W=16, K=8, T=5, width=64, n_slices=2, phase_major=pm)


def test_fused_tick_kernel_on_the_host(sim, oracle):
    """smr_mp_run_ticks: batches of ticks in one launch of mp_ticks_fused (a block = 64 groups x all replicas, block
    barriers for round boundaries) give what tick-by-tick launches give"""
    import test_mp_gpu as t
    with sim.patched():
        for fused in (1, 3, 16):
            t._run("cpu", oracle, G=200, R=5, S=2, W=64, n_ticks=36, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, fused=fused)
        t._run("cpu", oracle, G=130, R=5, S=4, W=64, n_ticks=30, drop_p=0.1, timeout_frac=0.0, hb_every=3, preset=True, fused=7)
        t._run("cpu", oracle, G=64, R=5, S=1, W=64, n_ticks=24, drop_p=0.05, timeout_frac=0.3, hb_every=4, preset=False, fused=4)
        t._run("cpu", oracle, G=65, R=3, S=2, W=32, n_ticks=30, drop_p=0.2, timeout_frac=0.5, hb_every=2, preset=True, fused=5)
        t._run("cpu", oracle, G=100, R=7, S=2, W=64, n_ticks=30, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, fused=6)
        t._run_bench_shape("cpu", oracle, G=130, frac=0.25, span=8, n_ticks=20, every=4, fused=16)


def test_spread_layout_images_on_the_host(sim):
    """layout L2 of the MultiPaxos cluster engine with every rank in this process (tests/test_spread_mp.py): live masks,
    image pack / unpack kernels, exchange plan -- against the co-located engine, every tick"""
    import test_spread_mp as t
    with sim.patched():
        t.run_spread_vs_colocated("cpu", G=256, R=5, S=2, W=64, world=2, n_ticks=24, drop_p=0.1, timeout_frac=1.0)
        t.run_spread_vs_colocated("cpu", G=192, R=5, S=2, W=64, world=3, n_ticks=18, drop_p=0.1, timeout_frac=1.0)
        t.run_spread_vs_colocated("cpu", G=128, R=3, S=2, W=32, world=2, n_ticks=16, drop_p=0.2, timeout_frac=0.5, hb_every=2)
        # (the runs above: the blocks' rounds in ONE launch, round 5's default; round 4's per-block launches stay selectable)
        t.run_spread_vs_colocated("cpu", G=192, R=5, S=2, W=64, world=3, n_ticks=12, drop_p=0.1, timeout_frac=1.0, rounds=1)


def test_heartbeater_kernels_on_the_host(sim, oracle):
    """the batched Heartbeater (f.4): hear timers, send ticker, reply counters -- the traces and a random call stream"""
    import test_zz_hb_gpu as t
    with sim.patched():
        t.test_traces_on_the_engine("cpu", oracle)
        t.test_random_calls_match_oracle("cpu", oracle, 300, 5, 0)
        t.test_random_calls_match_oracle("cpu", oracle, 130, 3, 2)


def test_lease_manager_kernels_on_the_host(sim, oracle):
    """the batched LeaseManager (f.4): the reference's unit tests on the kernels, and random notices against the oracle"""
    import test_zz_lease_gpu as t
    import lease_scenarios as LS
    with sim.patched():
        for tr in LS.ALL_TRACES:
            t.test_reference_trace_on_the_engine("cpu", tr)
        t.test_mutual_leases_on_the_engine("cpu")
        t.test_random_notices_match_oracle("cpu", oracle, 300, 5, 2, 0)
        t.test_random_notices_match_oracle("cpu", oracle, 130, 8, 7, 2)
        t.test_create_rejects_what_new_and_setup_rejects("cpu")
        t.test_sets_kernel("cpu")


def test_heartbeater_drives_the_multipaxos_engine_on_the_host(sim, oracle):
    import test_zz_hb_gpu as t
    with sim.patched():
        t.test_timeouts_feed_the_multipaxos_engine("cpu", oracle)


def test_string_kv_state_machine_kernel_on_the_host(sim):
    """the device KV over real keys and values (f.3): reference state-machine tests, dict + host executor, full / refused"""
    import test_zz_skv_gpu as t
    with sim.patched():
        t.test_reference_state_machine_tests("cpu")
        t.test_put_rand_get_rand_per_group_and_the_host_state_machine("cpu")
        t.test_full_table_and_heap_are_sticky_not_silent("cpu")


def test_craft_follower_kernels_on_the_host(sim, oracle):
    """the CRaft follower (shard bitmaps absorbed, execution gated on `majority` shards) and the Reconstruct responder"""
    import test_zz_craft_follower_gpu as t
    with sim.patched():
        for trace in t.TRACES:
            t.test_trace_on_the_engine("cpu", trace)
        t.test_crafted_rounds_match_the_oracle("cpu", oracle, 300, 32, 1)


def test_epaxos_explicit_prepare_kernels_on_the_host(sim, oracle):
    """EPaxos recovery of a dead command leader's row: the hand-derived traces on the kernels, and a crash-and-recovery cluster
    run against the oracle cluster, call by call"""
    import test_zz_ep_recovery_gpu as t
    with sim.patched():
        for trace in t.TRACES:
            t.test_trace_on_the_engine("cpu", trace)
        t.test_crash_and_recovery_matches_the_oracle_cluster("cpu", oracle, 200, 0, 0.0)
        t.test_crash_and_recovery_matches_the_oracle_cluster("cpu", oracle, 130, 3, 0.2)
        t.test_recovery_needs_the_flag("cpu")
        import test_zzz_ep_recovery_exec_gpu as tx
        tx.test_crash_and_recovery_with_execution_matches_the_oracle_cluster("cpu", oracle, 200, 1, 0.0)
        tx.test_crash_and_recovery_with_execution_matches_the_oracle_cluster("cpu", oracle, 150, 5, 0.15)


def test_device_resident_epaxos_cluster_tick_on_the_host(sim, oracle):
    import test_zz_ep_cluster_gpu as t
    with sim.patched():
        t.test_device_cluster_tick_matches_the_oracle_cluster("cpu", oracle, 300, 6, 0.15)
        assert t.run_fused_vs_driver("cpu", 200, 6, 0.15, T=6, oracle=oracle) > 0    # smr_ep_cluster_tick: one launch / launch by launch / the driver / the oracles
        t.run_fused_vs_driver("cpu", 130, 16, 0.0, T=5, execute=False, oracle=oracle)
        assert t.run_fused_vs_driver("cpu", 90, 4, 0.15, T=5, R=3, W=16, oracle=oracle) > 0    # populations 3 and 7 (the 8-replica kernel instances)
        assert t.run_fused_vs_driver("cpu", 70, 6, 0.15, T=5, R=7, W=16, oracle=oracle) > 0
        assert t.run_fused_vs_driver("cpu", 200, 6, 0.15, T=6, oracle=oracle, phase_major=True) > 0   # the leaders' steps phase by phase (set_mode bit 1)
        assert t.run_fused_vs_driver("cpu", 70, 6, 0.15, T=5, R=7, W=16, oracle=oracle, phase_major=True) > 0
        t.run_fused_vs_driver("cpu", 130, 16, 0.1, T=5, execute=False, oracle=oracle, phase_major=True)
        t.run_shared_table_vs_private("cpu", G=150, K=6)             # round 5: one per-key table per cluster vs private tables


def test_spread_epaxos_exchange_on_the_host(sim, oracle):
    """layout L2 of the EPaxos cluster with every rank in this process (tests/test_spread_ep.py): one all-to-all per
    exchange, both schedules -- against the co-located closed loop, every tick and the final state"""
    import test_spread_ep as t
    with sim.patched():
        t.run_spread_vs_colocated("cpu", G=130, world=2, n_ticks=6, loss=0.15, oracle=oracle)     # (oracle: also held against five EpOracles)
        t.run_spread_vs_colocated("cpu", G=100, world=3, n_ticks=5, loss=0.15)
        t.run_spread_vs_colocated("cpu", G=21, world=8, n_ticks=4, loss=0.1)
        job = t.run_spread_vs_colocated("cpu", G=120, world=4, n_ticks=6, loss=0.15, K=6, execute=True, oracle=oracle)
        assert job.ranks[0].exchanges_per_tick() == 17
        # execution on with the 5-exchange schedule: the phase-by-phase order of the colocated loop (smr_ep_cluster_set_mode bit 1)
        job = t.run_spread_vs_colocated("cpu", G=120, world=4, n_ticks=6, loss=0.15, K=6, execute=True, ordered=False, ref_phase_major=True, oracle=oracle)
        assert job.ranks[0].exchanges_per_tick() == 5
        assert t.run_spread_vs_colocated("cpu", G=70, world=1, n_ticks=4, loss=0.1).ranks[0].bytes_sent == 0   # one rank: nothing leaves it
        t.run_spread_vs_colocated("cpu", G=60, world=2, n_ticks=4, loss=0.1, R=3, K=4)                         # three replicas


def test_spread_rspaxos_library_tick_on_the_host(sim):
    """round 6: layout L2 of the RSPaxos engine with the tick's segments inside the library (smr_rsp_spread_*, csrc/rsp_spread.hip)
    against the co-located steady loop: commits, every replica's state, the shard bytes every follower received"""
    import test_spread_rsp as t
    with sim.patched():
        assert t.run_spread_vs_colocated("cpu", 2, 130, library_tick=True) > 0
        assert t.run_spread_vs_colocated("cpu", 3, 100, library_tick=True) > 0
        assert t.run_spread_vs_colocated("cpu", 8, 70, T=5, library_tick=True) > 0
        assert t.run_spread_vs_colocated("cpu", 4, 96, W=32, L=4113, loss=0.0, T=4, library_tick=True) > 0
        assert t.run_spread_vs_colocated("cpu", 1, 60, T=5, library_tick=True) > 0       # one rank: ONE C call per tick, its own exchange a device copy


def test_spread_epaxos_library_tick_on_the_host(sim, oracle):
    """round 6: layout L2 of the EPaxos cluster with the tick inside the library (smr_ep_spread_*, csrc/ep_spread.hip) -- both
    schedules, populations 3 and 5, 1 / 2 / 3 / 4 / 8 ranks -- against the co-located loop and the oracle cluster"""
    import test_spread_ep as t
    with sim.patched():
        t.run_library_tick_cases("cpu", oracle)


def test_cxx_epaxos_host_loop_on_the_host(sim, tmp_path):
    """examples/ep_host_loop.cpp (the one-call EPaxos cluster tick from C++) compiled for the host against the emulator build
    of the library: everything proposed commits and executes, the replicas' KV stores agree"""
    import os
    import subprocess
    import test_zzz_example_ep_gpu as t
    lib = sim.build()
    exe = str(tmp_path / "ep_host_loop_sim")
    here = os.path.dirname(os.path.abspath(sim.__file__))
    subprocess.check_call([sim.CXX, "-std=c++17", "-O1", "-w", "-DEP_HOST_LOOP_ON_THE_EMULATOR", "-DhipStreamCreate(s)=hipStreamCreateWithFlags(s,0)",
                           "-I", here, "-I", os.path.join(t.ROOT, "include"), os.path.join(t.ROOT, "examples", "ep_host_loop.cpp"),
                           lib, "-Wl,-rpath," + os.path.dirname(lib), "-o", exe])
    t.check_output(subprocess.check_output([exe, "96", "6"], timeout=300).decode(), 96, 6)


def test_cxx_rspaxos_payload_loop_on_the_host(sim, tmp_path):
    """examples/rsp_payload_loop.cpp (RSPaxos replicas + payload stores from C++: leader change, reconstruction, every executed batch
    read back) compiled for the host against the emulator build of the library"""
    import os
    import subprocess
    import test_zzz_example_rsp_payload_gpu as t
    lib = sim.build()
    exe = str(tmp_path / "rsp_payload_loop_sim")
    here = os.path.dirname(os.path.abspath(sim.__file__))
    subprocess.check_call([sim.CXX, "-std=c++17", "-O1", "-w", "-DRSP_PAYLOAD_LOOP_ON_THE_EMULATOR", "-I", here, "-I", os.path.join(t.ROOT, "include"),
                           os.path.join(t.ROOT, "examples", "rsp_payload_loop.cpp"), lib, "-Wl,-rpath," + os.path.dirname(lib), "-o", exe])
    t.check_output(subprocess.check_output([exe, "70", "100"], timeout=300).decode(), 70)


def test_cxx_raft_wire_loop_on_the_host(sim, tmp_path):
    """examples/raft_wire_loop.cpp (Raft replication with the replies as wire frames written and parsed by kernels) compiled for
    the host against the emulator build of the library: every appended entry commits"""
    import os
    import subprocess
    import test_zzz_example_raft_wire_gpu as t
    lib = sim.build()
    exe = str(tmp_path / "raft_wire_loop_sim")
    here = os.path.dirname(os.path.abspath(sim.__file__))
    subprocess.check_call([sim.CXX, "-std=c++17", "-O1", "-w", "-DRAFT_WIRE_LOOP_ON_THE_EMULATOR", "-I", here, "-I", os.path.join(t.ROOT, "include"),
                           os.path.join(t.ROOT, "examples", "raft_wire_loop.cpp"), lib, "-Wl,-rpath," + os.path.dirname(lib), "-o", exe])
    t.check_output(subprocess.check_output([exe, "96", "6"], timeout=300).decode(), 96, 6)
