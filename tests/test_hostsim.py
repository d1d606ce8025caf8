"""The Raft and EPaxos kernels -- the shipped .hip sources, compiled for the host by tests/hostsim and run
one lane at a time -- against the CPU oracle, through the shipped C-ABI and Python mirror.  This reruns
the scenarios of tests/test_{raft,ep}_gpu.py where no GPU is at hand: it checks the kernels' logic,
not their behaviour on the device (the gpu-marked tests do that)."""
import pytest


@pytest.fixture(scope="module")
def sim():
    import hostsim
    hostsim.build()
    return hostsim


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
        t.test_raft_three_replicas_and_stepdown("cpu", oracle)
        t.test_craft_threshold("cpu", oracle)
        t.test_follower_and_elections_match_oracle("cpu", oracle, 300, 64)
        t.test_closed_loop_cluster_matches_oracle("cpu", oracle)


def test_epaxos_execution_kernel_on_the_host(sim, oracle):
    import test_zz_ep_exec_gpu as t
    with sim.patched():
        t.test_execution_traces("cpu", oracle)
        t.test_handler_streams_with_execution("cpu", oracle, 300, 32, 0)
        t.test_handler_streams_with_execution("cpu", oracle, 257, 16, 3)
        t.test_closed_loop_cluster_with_execution("cpu", oracle, 0.0)
        t.test_closed_loop_cluster_with_execution("cpu", oracle, 0.15)
