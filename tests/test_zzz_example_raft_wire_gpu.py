"""examples/raft_wire_loop.cpp: a C++ Raft replication loop over the C-ABI whose AppendEntriesReplies are written as wire frames
by `smr_wire_emit_raft_replies` and parsed by `smr_wire_ingest_raft_replies` without leaving the device (the emit call's slots
as the leader's connections) -- built with hipcc and run on the device: every appended entry must commit, every reply must
have travelled as a frame.  (tests/test_hostsim.py builds and runs the same file against the kernel-source emulator.)"""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check_output(out, G, ticks):
    m = re.search(r"(\d+) entries committed by the leader of (\d+) groups in (\d+) ticks; (\d+) AppendEntriesReply frames written and parsed on the device, (\d+) malformed", out)
    assert m, out
    assert int(m.group(1)) == 2 * G * ticks and int(m.group(2)) == G and int(m.group(4)) == 4 * G * ticks and int(m.group(5)) == 0, out


def test_cxx_raft_wire_loop_commits_everything(engine_lib, tmp_path):
    exe = tmp_path / "raft_wire_loop"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "raft_wire_loop.cpp"), "-L", os.path.join(ROOT, "summerset_amd"),
                           "-lsummerset_hip", "-Wl,-rpath," + os.path.join(ROOT, "summerset_amd"), "-o", str(exe)])
    check_output(subprocess.check_output([str(exe), "2048", "12"], timeout=120).decode(), 2048, 12)
