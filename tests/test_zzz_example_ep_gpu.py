"""examples/ep_host_loop.cpp: a C++ host loop over the C-ABI's one-call EPaxos cluster tick (`smr_ep_cluster_tick`), built
with hipcc and run on the device -- every instance the five replicas propose must commit inside its tick, every command
must execute at every replica, and the replicas' KV stores must agree.  Sorted last: written when no device was at hand
(tests/test_hostsim.py builds and runs the same file against the kernel-source emulator)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu          # first device run: GPUTEST_r02 (11 XPASS); quarantine removed in round 3
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check_output(out, G, ticks):
    m = re.search(r"(\d+) instances committed \((\d+) on the fast path\), (\d+) commands executed", out)
    assert m, out
    # (a command can run twice: an executing slot that add_edge re-inserts, execution.rs:57-59 -- counted as submitted again)
    assert int(m.group(1)) == 5 * G * ticks and 5 * 5 * G * ticks <= int(m.group(3)) <= 5 * 5 * G * ticks * 1.02 and int(m.group(2)) > 0, out
    assert "replicas' KV stores agree" in out, out


def test_cxx_epaxos_host_loop_commits_and_executes_everything(engine_lib, tmp_path):
    exe = tmp_path / "ep_host_loop"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "ep_host_loop.cpp"), "-L", os.path.join(ROOT, "summerset_amd"),
                           "-lsummerset_hip", "-Wl,-rpath," + os.path.join(ROOT, "summerset_amd"), "-o", str(exe)])
    check_output(subprocess.check_output([str(exe), "2048", "12"], timeout=120).decode(), 2048, 12)
