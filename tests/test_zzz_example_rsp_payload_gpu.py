"""examples/rsp_payload_loop.cpp: RSPaxos with real bytes from C++ over the C-ABI -- five `smr_rsp_*` replica objects and five
`smr_rsp_pstore_*` payload stores through steady appends, a leader change with shard merging, reconstruction reads and
execution at the new leader; every executed batch is read out of the store and must be the bytes the old leader serialized.
Built with hipcc and run on the device (tests/test_hostsim.py builds and runs the same file against the kernel-source emulator)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check_output(out, G):
    m = re.search(r"(\d+) batches read back byte for byte at the old leader, (\d+) at the new leader after reconstruction, (\d+) open", out)
    assert m and int(m.group(1)) == 4 * G and int(m.group(2)) == 4 * G and int(m.group(3)) == 2 * G, out
    m = re.search(r"(\d+) shards copied, (\d+) rebuilt, (\d+) unsatisfied", out)
    assert m and int(m.group(2)) > 0 and int(m.group(3)) == 0, out
    assert out.strip().endswith("ok"), out


def test_cxx_rspaxos_payload_loop_reads_every_batch_back(engine_lib, tmp_path):
    exe = tmp_path / "rsp_payload_loop"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "rsp_payload_loop.cpp"), "-L", os.path.join(ROOT, "summerset_amd"),
                           "-lsummerset_hip", "-Wl,-rpath," + os.path.join(ROOT, "summerset_amd"), "-o", str(exe)])
    check_output(subprocess.check_output([str(exe), "1024", "1000"], timeout=120).decode(), 1024)
    check_output(subprocess.check_output([str(exe), "130", "4113"], timeout=120).decode(), 130)
