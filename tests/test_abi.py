"""CPU-side checks of the C-ABI library: it builds for gfx950 without a GPU, loads,
exports every symbol include/summerset_hip.h declares, and fails LOUDLY without a device."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "summerset_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smr_[a-z0-9_]+)\s*\(", src)))


def test_exports_every_declared_symbol(engine_lib):
    from summerset_amd import _lib
    names = _declared()
    assert len(names) >= 30
    bound = {n for n, _, _ in _lib.SYMBOLS}
    for n in names:
        assert hasattr(engine_lib, n), "library does not export %s" % n
        assert n in bound, "Python binding table misses %s" % n
    assert engine_lib.smr_abi_version() == 1


def test_integration_md_declares_every_symbol(engine_lib):
    """INTEGRATION.md named 145 of the 237 exported symbols.  Its §6 is generated from the header
    (tools/gen_rust_extern.py): the block must be current, and every `smr_*` symbol the built library exports must have its
    `pub fn` there -- and the other way round."""
    import subprocess
    import sys
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_extern.py"), "--check"])
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index("<!-- BEGIN GENERATED: tools/gen_rust_extern.py -->"):doc.index("<!-- END GENERATED: tools/gen_rust_extern.py -->")]
    bound = set(re.findall(r"pub fn (smr_[a-z0-9_]+)\(", block))
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "summerset_amd", "libsummerset_hip.so")], check=True,
                        capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.split() and ln.split()[-1].startswith("smr_") and ln.split()[-2] in "TtWw"}
    assert exported - bound == set(), "exported but not declared in INTEGRATION.md: %s" % sorted(exported - bound)
    assert bound - exported == set(), "declared in INTEGRATION.md but not exported: %s" % sorted(bound - exported)
    assert bound == set(_declared())
    # the structs are #[repr(C)] field for field: spot checks against the ctypes mirror the tests run on
    assert "pub struct SmrMpCfg {" in block and "pub n_groups: u32," in block and "pub struct SmrEpClusterOut {" in block


def test_host_only_entry_points(engine_lib):
    from summerset_amd import rs_matrix, rs_shard_len
    assert rs_matrix(3, 2).tolist() == [[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1], [15, 8, 6]]
    assert rs_shard_len(4099, 3) == 1367 and rs_shard_len(18, 3) == 6 and rs_shard_len(0, 3) == 0


def test_host_matrix_matches_oracle(engine_lib, oracle):
    from summerset_amd import rs_matrix
    for d, p in ((3, 2), (5, 5), (6, 4), (9, 6), (12, 8), (1, 1), (16, 8)):
        assert np.array_equal(rs_matrix(d, p), oracle.rs_matrix(d, p)), (d, p)


def test_argument_errors_do_not_need_a_device(engine_lib):
    import ctypes as C
    from summerset_amd import _lib
    from summerset_amd._lib import MpCfg, SummersetError, check
    h = C.c_void_p()
    for cfg, frag in ((MpCfg(0, 5, 0, 0, 0, 64, 16, 68, 0), "n_groups"),
                      (MpCfg(8, 2, 0, 0, 0, 64, 16, 68, 0), "population"),
                      (MpCfg(8, 5, 0, 0, 0, 48, 16, 68, 0), "power of two"),
                      (MpCfg(8, 5, 3, 0, 0, 64, 16, 68, 0), "fault_tolerance")):
        with pytest.raises(SummersetError) as e:
            check(engine_lib.smr_mp_cluster_create(C.byref(cfg), C.byref(h)))
        assert e.value.code == _lib.SMR_ERR_ARG and frag in e.value.msg
    with pytest.raises(SummersetError) as e:
        check(engine_lib.smr_rs_encode(None, 0, 0, 1, 3, 2, None, 0, 0, None))
    assert "codeword is null" in e.value.msg
    with pytest.raises(SummersetError) as e:
        check(engine_lib.smr_rs_encode(None, 10, 16, 1, 0, 2, None, 0, 0, None))
    assert "num_data_shards is zero" in e.value.msg


def test_argument_errors_of_the_other_protocol_objects(engine_lib):
    import ctypes as C
    from summerset_amd import _lib
    from summerset_amd._lib import EpCfg, RaftCfg, RspCfg, SummersetError, check
    h = C.c_void_p()
    for cfg, frag in ((RaftCfg(0, 5, 0, 0, 0, 64, 1), "n_groups"), (RaftCfg(8, 9, 0, 0, 0, 64, 1), "population"),
                      (RaftCfg(8, 5, 5, 0, 0, 64, 1), "leader_id"), (RaftCfg(8, 5, 0, 0, 0, 40, 1), "power of two"),
                      (RaftCfg(8, 5, 0, 3, 0, 64, 1), "commit_extra")):
        with pytest.raises(SummersetError) as e:
            check(engine_lib.smr_raft_leader_create(C.byref(cfg), C.byref(h)))
        assert e.value.code == _lib.SMR_ERR_ARG and frag in e.value.msg
    for cfg, frag in ((EpCfg(0, 5, 0, 1, 0, 32, 64), "n_groups"), (EpCfg(8, 2, 0, 1, 0, 32, 64), "population"),
                      (EpCfg(8, 5, 5, 1, 0, 32, 64), "replica id"), (EpCfg(8, 5, 0, 1, 0, 24, 64), "power of two"),
                      (EpCfg(8, 5, 0, 1, 0, 32, 0), "n_keys"), (EpCfg(8, 5, 0, 1, 0, 32, 256), "n_keys")):
        with pytest.raises(SummersetError) as e:
            check(engine_lib.smr_ep_replica_create(C.byref(cfg), C.byref(h)))
        assert e.value.code == _lib.SMR_ERR_ARG and frag in e.value.msg
    for cfg, frag in ((EpCfg(8, 5, 0, 1, 2, 32, 64), "execute must be"), (EpCfg(8, 8, 0, 1, 1, 8192, 64), "15-bit")):
        with pytest.raises(SummersetError) as e:
            check(engine_lib.smr_ep_replica_create(C.byref(cfg), C.byref(h)))
        assert e.value.code == _lib.SMR_ERR_ARG and frag in e.value.msg
    for cfg, frag in ((RspCfg(0, 5, 0, 0, 0, 32), "n_groups"), (RspCfg(8, 2, 0, 0, 0, 32), "population"),
                      (RspCfg(8, 5, 5, 0, 0, 32), "replica id"), (RspCfg(8, 5, 0, 0, 0, 24), "power of two"),
                      (RspCfg(8, 5, 0, 3, 0, 32), "fault_tolerance")):       # mod.rs:600-604: at most population - majority
        with pytest.raises(SummersetError) as e:
            check(engine_lib.smr_rsp_replica_create(C.byref(cfg), C.byref(h)))
        assert e.value.code == _lib.SMR_ERR_ARG and frag in e.value.msg
    # null handles are refused, not dereferenced
    for fn, args in ((engine_lib.smr_raft_replica_preset, (None, 0, 0, 1, 0xFF)),
                     (engine_lib.smr_rsp_req_batch, (None, None, None, None)),
                     (engine_lib.smr_rsp_preset_leader, (None, 0)),
                     (engine_lib.smr_ep_exec_dump, (None, None, None, None, None)),
                     (engine_lib.smr_ep_propose, (None, None, None, None, None)),
                     (engine_lib.smr_mp_end_tick, (None,)),
                     (engine_lib.smr_repnothing_stats, (None, None, None, None, None))):
        with pytest.raises(SummersetError):
            check(fn(*args))


def test_no_silent_cpu_fallback(engine_lib):
    """Without a GPU every compute entry point must raise, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("device present")
    import ctypes as C
    from summerset_amd._lib import MpCfg, SummersetError, check
    assert engine_lib.smr_device_count() < 0
    h = C.c_void_p()
    with pytest.raises(SummersetError) as e:
        check(engine_lib.smr_mp_cluster_create(C.byref(MpCfg(8, 5, 0, 0, 0, 64, 16, 68, 0)), C.byref(h)))
    assert e.value.code == -2


def test_package_does_not_import_the_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import summerset_amd; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle leaked'" % ROOT)
    subprocess.check_call([sys.executable, "-c", code])
    for dirpath, _, files in os.walk(os.path.join(ROOT, "summerset_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def test_header_is_plain_c(tmp_path):
    """the boundary is a C ABI: the header must compile as C99 on its own"""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "summerset_hip.h"\nint main(void) { smr_mp_cfg c; c.n_groups = 1; return (int)sizeof(smr_wire_msg) * 0 + (int)c.n_groups - 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                           "-o", str(tmp_path / "abi.o")])


def test_cxx_host_loop_example_builds_and_links(engine_lib, tmp_path):
    """examples/mp_host_loop.cpp (the reference's run() loop over the C-ABI) compiles and links against the
    built library; running it needs a GPU"""
    import subprocess
    out = tmp_path / "mp_host_loop"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "mp_host_loop.cpp"), "-L", os.path.join(ROOT, "summerset_amd"),
                           "-lsummerset_hip", "-Wl,-rpath," + os.path.join(ROOT, "summerset_amd"), "-o", str(out)])
    assert out.exists()


def test_stream_handle_goes_through_one_helper(monkeypatch):
    """Round 1's device failure: `KvStateMachine._stream = QuorumReadGroup._stream` unwrapped a staticmethod into
    a bound method, and the emulator replaced `_stream` on every class, so no CPU test ran the real line.  Now
    every mirror calls `_lib.stream_ptr`, the emulator patches torch only, and this test runs the helper itself."""
    import inspect
    import torch
    from summerset_amd import _lib, epaxos, multipaxos, quorumread, raft, rscoding, rspaxos
    assert _lib.stream_ptr(7) == 7 and _lib.stream_ptr(0) == 0

    class S:
        cuda_stream = 0x1234
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: S())
    assert _lib.stream_ptr(None) == 0x1234
    n_calls = 0
    for mod in (epaxos, multipaxos, quorumread, raft, rscoding, rspaxos):
        src = inspect.getsource(mod)
        assert "current_stream" not in src, mod.__name__          # no private copy of the helper
        n_calls += src.count("stream_ptr(stream)")
        for _, cls in inspect.getmembers(mod, inspect.isclass):
            assert not hasattr(cls, "_stream"), cls
    assert n_calls > 40


def _import_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("smr_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_final_line_is_small_and_parses(capsys, tmp_path, monkeypatch):
    """round 4's 20.7 KB line was not parsed by the driver.  The final stdout line is built from a full record
    (round 4's own, every leg present) and must stay under 4 KB, round-trip through json, carry the contract fields, `roofline`
    and `cpu_baseline`, and per secondary leg exactly the seven compact keys; the full record goes to bench_detail.json."""
    import json
    bench = _import_bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r7m_bench_driver_command.json")))
    assert len(json.dumps(full)) > 15000                          # the record that did not parse
    full["craft_payload"] = dict(full["rspaxos_payload"])          # round 5's extra leg
    full["epaxos_execution"] = {"error": "RuntimeError: " + "x" * 500}
    full["roofline"]["traffic"] = float("nan")                    # never NaN / Infinity on the line
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit_line(full)
    out = capsys.readouterr().out.strip().splitlines()
    text = out[-1]
    assert len(text) < bench.LINE_BUDGET < 6000, len(text)
    line = json.loads(text)
    assert "NaN" not in text and "Infinity" not in text
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "legs_failed"):
        assert k in line, k
    assert line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1
    assert abs(line["value"] - full["value"]) / full["value"] < 1e-6
    assert abs(line["ms_per_step"] - full["ms_per_step"]) / full["ms_per_step"] < 1e-5
    assert len(line["config"]["workload"]) <= 200 and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["traffic"] is None
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) / r["achieved"] < 1e-3
    assert set(r["whole_tick"]) == {"us", "frac_alg", "frac_pmc"}
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 32 and line["cpu_baseline"]["single_core_value"] > 0
    keys = {"value", "unit", "ms_per_tick", "frac", "frac_on_8d_bytes", "traffic_ratio", "cpu_cores"}
    for name in bench.SECONDARY_LEGS:
        if name in full and "error" not in full[name]:
            assert set(line[name]) == keys, (name, line[name])
    assert abs(line["rspaxos"]["frac_on_8d_bytes"] - 0.2027) < 1e-3 and abs(line["rspaxos"]["frac"] - 0.3228) < 1e-3
    assert abs(line["epaxos_cluster"]["ms_per_tick"] - 0.5404) < 1e-3 and line["epaxos_cluster"]["traffic_ratio"] > 10
    assert abs(line["raft_quorum"]["ms_per_tick"] - 0.01133) < 1e-4 and line["raft_quorum"]["cpu_cores"] == 32
    assert abs(line["wire_ingest"]["ms_per_tick"] - 0.2237) < 1e-3
    assert len(line["epaxos_execution"]["error"]) <= 120
    detail = json.load(open(tmp_path / bench.DETAIL_FILE))       # nothing is lost: the full record sits beside the script
    assert detail["timed_regions"] == full["timed_regions"] and detail["l2"]["exchange"] == full["l2"]["exchange"]
