"""examples/mp_host_loop.cpp: a C++ host loop over the C-ABI (the reference's run() loop shape) built with
hipcc and run on the device -- every batch the synthetic clients hand in must commit."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cxx_host_loop_commits_everything(engine_lib, tmp_path):
    exe = tmp_path / "mp_host_loop"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "mp_host_loop.cpp"), "-L", os.path.join(ROOT, "summerset_amd"),
                           "-lsummerset_hip", "-Wl,-rpath," + os.path.join(ROOT, "summerset_amd"), "-o", str(exe)])
    out = subprocess.check_output([str(exe), "512", "8", "20"], timeout=120).decode()
    m = re.search(r"(\d+) slots committed by replica 0", out)
    assert m and int(m.group(1)) == 512 * 8 * 20, out
    assert "commit_bar 160, exec_bar 160, log_len 160" in out, out
