#!/usr/bin/env python
"""bench.py -- committed slots/sec of the batched MultiPaxos hot path on MI355X.

One "step" = one lock-step tick (rounds R1..R4) over G replica
groups x 5 replicas with S new client batches per group: leader append ->
follower accept -> quorum tally + commit/exec bars -> (every H ticks) heartbeat.
Inputs (batch tokens, reply order / loss words, timeout events) are resident in
HBM before the timed region.  One process per GPU; groups shard across ranks
with no data-path collective (weak scaling: G groups PER GPU).

Prints ONE JSON line (rank 0).  `value` counts leader-side Accepting->Committed
transitions (multipaxos/messages.rs:412-433) per second, whole job.
Also reports, in the same line: the HBM roofline of the quorum kernel (R3), the
CPU oracle timed on this host on a bounded sample (`cpu_baseline`), and the
RS(3,2) encode rate of BASELINE config 4 (`rs_encode`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
PMC_FILES = ("t3z_pmc_traffic.json", "t2z_pmc_traffic.json", "t1z_pmc_traffic.json", "r9z_pmc_traffic.json", "r8z_pmc_traffic.json", "r8m_pmc_traffic.json", "r6m_pmc_traffic.json", "r5m_pmc_traffic.json", "round3/r4m_pmc_traffic.json")      # the newest committed record first (tools/final_record.sh TAG)
PMC_FILE = next((os.path.join(ROOT, "profiles", f) for f in PMC_FILES if os.path.exists(os.path.join(ROOT, "profiles", f))),
                os.path.join(ROOT, "profiles", PMC_FILES[0]))
PMC_NOTE = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/pmc_probe.py, separate runs, gfx950 corrections of "
            "MI355X_MICROARCH.md applied; bytes per launch of the SAME grid size as the timed launch: 65536 groups, S=32, default workload)"
            % os.path.basename(PMC_FILE))


# ---- the final stdout line ------------------------------------------------------------------------------------------------------
# The driver reads the LAST stdout line from a bounded tail: round 4's 20.7 KB line did not parse.  The final
# line is therefore a fixed, small set of fields (< 4 KB); everything else -- prose, sweeps, per-region lists, the exchange's byte
# counts -- goes to bench_detail.json (repo root, and gpurun_out/ when that exists), untouched.
LINE_BUDGET = 4000
DETAIL_FILE = "bench_detail.json"


def _num(x, digits=6):
    """a float to `digits` significant digits (None / ints / bools pass through): the line is for reading and the judge's recomputation"""
    if isinstance(x, bool) or x is None or isinstance(x, int):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, x))
    return x


def _leg_roof(leg):
    """the roofline object a leg's headline figure is quoted on (the legs nest it differently)"""
    for path in (("whole_tick_roofline",), ("roofline",), ("one_call_per_tick_phase_by_phase", "roofline"), ("one_call_per_tick", "roofline")):
        o = leg
        for k in path:
            o = o.get(k) if isinstance(o, dict) else None
        if isinstance(o, dict):
            return o
    return {}


def compact_leg(leg):
    """exactly {value, unit, ms_per_tick, frac, frac_on_8d_bytes, traffic_ratio, cpu_cores} of one secondary leg (or {error})"""
    if not isinstance(leg, dict):
        return None
    if "error" in leg:
        return {"error": str(leg["error"])[:120]}
    src = leg
    if isinstance(leg.get("one_call_per_tick_phase_by_phase"), dict):   # the EPaxos cluster leg quotes its one-launch tick
        src = leg["one_call_per_tick_phase_by_phase"]
    roof = _leg_roof(leg)
    ms = src.get("ms_per_tick")
    if ms is None:
        for k, f in (("us_per_tick", 1e-3), ("call_us", 1e-3)):
            if src.get(k) is not None:
                ms = src[k] * f
                break
    if ms is None and roof.get("avg_launch_us") is not None:
        ms = roof["avg_launch_us"] * 1e-3
    alg = roof.get("alg_bytes_per_launch", roof.get("alg_bytes_per_tick"))
    traffic = roof.get("traffic", roof.get("traffic_per_tick"))
    f8d = roof.get("frac_on_survey_8d_bytes", roof.get("frac"))
    cpu = leg.get("cpu_baseline") if isinstance(leg.get("cpu_baseline"), dict) else {}
    return {"value": _num(src.get("value")), "unit": str(src.get("unit", leg.get("unit")))[:48], "ms_per_tick": _num(ms),
            "frac": _num(roof.get("frac"), 4), "frac_on_8d_bytes": _num(f8d, 4),
            "traffic_ratio": _num(traffic / alg, 4) if traffic and alg else None,
            "cpu_cores": cpu.get("cores", leg.get("cores"))}


SECONDARY_LEGS = ("rs_encode", "raft_quorum", "epaxos_fast_quorum", "epaxos_cluster", "rspaxos", "rspaxos_payload", "craft_payload", "repnothing",
                  "wire_ingest", "reply_ingest", "epaxos_execution", "rspaxos_replica", "craft_leader", "quorum_read")


def compact_line(full):
    """the line the driver parses, from the full record (any layout's): contract fields, `roofline`, `cpu_baseline`, `legs_failed`, the
    L2 pass in four numbers and every secondary leg as compact_leg() -- nothing else."""
    cfg = full.get("config") or {}
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "ranks", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    out["value"], out["ms_per_step"] = _num(out["value"], 9), _num(out["ms_per_step"], 7)
    out["config"] = dict({"workload": str(cfg.get("workload", ""))[:200]},
                         **{k: cfg[k] for k in ("groups_per_gpu", "replicas", "slots_per_tick", "window", "layout", "spread_ranks", "value_bytes") if k in cfg})
    r = full.get("roofline")
    if isinstance(r, dict):
        wt = r.get("whole_tick") or {}
        out["roofline"] = {"bound": r.get("bound"), "kernel": str(r.get("kernel"))[:64], "alg_bytes_per_launch": _num(r.get("alg_bytes_per_launch")),
                           "avg_launch_us": _num(r.get("avg_launch_us")), "achieved": _num(r.get("achieved")), "peak": r.get("peak"),
                           "unit": r.get("unit"), "frac": _num(r.get("frac"), 4), "frac_pmc": _num(r.get("frac_pmc"), 4), "traffic": _num(r.get("traffic")),
                           "whole_tick": {"us": _num(wt.get("us")), "frac_alg": _num(wt.get("frac_alg"), 4), "frac_pmc": _num(wt.get("frac_pmc"), 4)}}
    else:
        out["roofline"] = None
    c = full.get("cpu_baseline")
    if isinstance(c, dict) and "error" not in c:
        out["cpu_baseline"] = {"value": _num(c.get("value")), "unit": c.get("unit"), "cores": c.get("cores"),
                               "single_core_value": _num(c.get("single_core_value")), "kind": c.get("kind"), "sample": str(c.get("sample", ""))[:160]}
    else:
        out["cpu_baseline"] = c if c is None else {"error": str(c.get("error"))[:120]}
    l2 = full.get("l2")
    if isinstance(l2, dict):
        out["l2"] = {"error": str(l2["error"])[:120]} if "error" in l2 else {
            "ranks": l2.get("ranks"), "virtual": str(l2.get("ranks_are", "")).startswith("virtual"), "ms_per_tick": _num(l2.get("ms_per_tick")),
            "ms_per_tick_per_virtual_rank": _num(l2.get("ms_per_tick_per_virtual_rank")),
            "steady_ms_per_tick_per_virtual_rank": _num((l2.get("steady_state") or {}).get("ms_per_tick_per_virtual_rank")),
            "value": _num(l2.get("value")), "backend": l2.get("backend"), "via": str((l2.get("exchange") or {}).get("via", ""))[:60]}
    if isinstance(full.get("exchange"), dict):                   # the spread layouts' own lines
        ex = full["exchange"]
        cpt = ex.get("collectives_per_tick")
        out["exchange"] = {"via": str(ex.get("via", ""))[:80], "collectives_per_tick": cpt if isinstance(cpt, int) else str(cpt)[:80],
                           "bytes_sent_per_tick_per_rank": _num(ex.get("bytes_sent_per_tick_per_rank"))}
    for name in SECONDARY_LEGS:
        if name in full:
            out[name] = compact_leg(full[name])
    out["legs_failed"] = full.get("legs_failed", [])
    out["detail"] = DETAIL_FILE
    return out


def emit_line(full):
    """write the full record beside the script, print the compact line LAST on stdout"""
    if globals().get("L2_EXCHANGE_FAILED"):                       # a library exchange that could not be bound at N > 1: a failed leg of ANY layout's line
        full["legs_failed"] = sorted(set(list(full.get("legs_failed") or []) + ["l2_exchange"]))
        full["l2_exchange_error"] = "; ".join(L2_EXCHANGE_FAILED)[:300]
    where = os.environ.get("SMR_BENCH_DETAIL_DIR")                # (tests: keep the repo root clean)
    for d in ((where,) if where else (ROOT, os.path.join(ROOT, "gpurun_out"))):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAIL_FILE), "w") as f:
                    json.dump(full, f, indent=1)
            except OSError as e:
                sys.stderr.write("bench.py: could not write %s: %s\n" % (os.path.join(d, DETAIL_FILE), e))
    text = json.dumps(compact_line(full), allow_nan=False)
    if len(text) > LINE_BUDGET:
        sys.stderr.write("bench.py: the final line is %d bytes (budget %d)\n" % (len(text), LINE_BUDGET))
    sys.stdout.flush()
    print(text)
    sys.stdout.flush()


def pmc_file():
    """the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, separate runs, gfx950 corrections applied: see the
    file's "corrections"); None if absent.  bench.py cannot collect PMC counters itself -- they need rocprofv3."""
    try:
        with open(PMC_FILE) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def pmc_traffic(kernel, pick=None):
    """the kernel's entry of the committed PMC record.  The record is keyed by (kernel, grid size) (tools/pmc_traffic.py: the
    probe runs some kernels at two sizes): pick = "smallest" / "largest" names the grid size this caller timed; None = the
    kernel's most-launched size (the only one for every kernel the probe runs at one size: `grids` == 1)."""
    d = pmc_file()
    e = d["kernels"].get(kernel) if d else None
    if e and pick and e.get("by_grid"):
        g = (min if pick == "smallest" else max)(int(x) for x in e["by_grid"])
        e = dict(e["by_grid"][str(g)], grid=g, grids=len(e["by_grid"]))
    return e


def _leg_traffic(kernel, pick=None):
    """HBM bytes per launch of a secondary leg's kernel from the committed PMC passes (tools/pmc_probe.py drives the same
    leg at the same shape), or None"""
    t = pmc_traffic(kernel, pick)
    return t["hbm_bytes_per_launch"] if t else None


TIMEOUT_HORIZON = 72     # ticks of the default run (60 timed + 12 warm-up): --timeouts is the fraction of groups per THIS many ticks


def n_regions(args):
    return args.repeats if args.repeats > 0 else (9 if args.steps <= 30 else 3)


def timeout_span(args):
    """ticks the groups' timeout ticks are drawn from: the whole run (warm-up + every timed region), at least the horizon"""
    return args.timeout_span if args.timeout_span is not None else max(args.steps * n_regions(args) + args.warmup, TIMEOUT_HORIZON)


def timeout_frac(args):
    """fraction of the groups that get a leader timeout somewhere in the span: --timeouts is per TIMEOUT_HORIZON ticks, so a longer
    run (more timed regions) sees the same changes per tick"""
    return args.timeouts if args.timeout_span is not None else min(1.0, args.timeouts * timeout_span(args) / TIMEOUT_HORIZON)


def timeouts_text(args):
    span = timeout_span(args)
    return "%.2f%% of the groups per %d ticks with a leader timeout (%.1f groups per tick)" % (
        timeout_frac(args) * 100, span, timeout_frac(args) * args.groups / span)


def parse():
    """options; the measurements behind the defaults are in profiles/HISTORY.md and DESIGN.md §4 / §7"""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--groups", type=int, default=65536, help="replica groups per GPU")
    ap.add_argument("--slots", type=int, default=32, help="S: new batches per group per tick")
    ap.add_argument("--hb-every", type=int, default=4)
    ap.add_argument("--window", type=int, default=512)
    ap.add_argument("--pool", type=int, default=4, help="distinct pre-generated tick inputs cycled in HBM")
    ap.add_argument("--drop", type=float, default=0.1)
    ap.add_argument("--timeouts", type=float, default=0.01)
    ap.add_argument("--timeout-span", type=int, default=None,
                    help="draw timeout ticks from [0, N); default max(steps + warmup, %d): a constant rate per tick" % TIMEOUT_HORIZON)
    ap.add_argument("--straggler-ticks", type=int, default=4, help="ticks a group in a leader change stays on the side list (0 = no list)")
    ap.add_argument("--fused", type=int, default=0, help="> 0: ticks per call through the fused tick kernel (no side stream)")
    ap.add_argument("--batch", type=int, default=8, help="> 0: ticks per smr_mp_run_ticks call with the side list on; 0 = one call per tick")
    ap.add_argument("--round-ticks", type=int, default=12, help="--fused only: ticks of the untimed per-round pass behind the timed region")
    ap.add_argument("--layout", choices=("colocated", "spread", "spread-epaxos", "colocated-epaxos", "spread-rspaxos"), default="colocated",
                    help="spread* = SURVEY 8e L2: replica r of block b on rank (b + r) mod N")
    ap.add_argument("--spread-ranks", type=int, default=4, help="spread layouts on ONE GPU: virtual ranks inside the process")
    ap.add_argument("--repeats", type=int, default=0, help="timed regions of --steps ticks (median reported); 0 = 9 if steps <= 30 else 3")
    ap.add_argument("--no-l2", action="store_true", help="skip the `l2` object (the workload a few ticks in the spread layout)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU oracle baseline")
    ap.add_argument("--no-rs", action="store_true", help="skip the RS(3,2) encode leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the other protocols' legs")
    ap.add_argument("--late-legs", action="store_true", help="also run the replica-engine / CRaft / quorum-read legs (child processes)")
    ap.add_argument("--role-rotation", type=int, default=0, help="1: rows of the bulk round launches by role (row 0 = every leader)")
    ap.add_argument("--leg", default=None, help="internal: run one secondary leg in this process and print its JSON")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--launch-check", action="store_true", help="only prove the launch: count the ranks with one all-reduce, no kernel")
    return ap.parse_args()


def _rs_cpu_run(a):
    host, L, seconds = a
    from oracle import oracle as O
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < seconds:
        O.rs_encode_batch(3, 2, host, L, L, 2048)
        done += 2048
    return done * L / 2**30 / (time.perf_counter() - t0), done


def rse_sweep(torch, dev):
    """benches/rse_bench.rs:19-26: RS(3,2) over String values of 4 KiB ... 4 MiB.  Per size: batches of codewords sized to
    ~256 MB per launch, 3 batches in rotation.  `encode` = compute_parity alone on bytes already laid out as a codeword;
    `from_data_and_encode` adds the from_data-equivalent work of the reference's loop body (rse_bench.rs:161-169): the
    serialized bytes into the codeword's data shards (pad + split) -- in one pass with the parity product
    (`smr_rs_from_data_encode`), and, for comparison, as round 2 did it: a device copy into the codeword buffer, then encode."""
    from summerset_amd import RSCodewordBatch
    out = []
    for size in (4096, 16 * 1024, 64 * 1024, 256 * 1024, 1024 * 1024, 4096 * 1024):
        L = size + (3 if size < 65536 else 5)                 # bincode String: varint length (0xFB u16 / 0xFC u32) + bytes
        n = max(8, (256 << 20) // (L * 5 // 3))
        srcs = [torch.randint(0, 256, (n, L), dtype=torch.uint8, device=dev) for _ in range(3)]
        cws = [RSCodewordBatch.from_data(x, 3, 2) for x in srcs]
        for c in cws:
            c.compute_parity()
        us_enc = _time_us(torch, lambda i: cws[i % 3].compute_parity(), 9)

        def both(i):
            c = cws[i % 3]
            c.buf[:, :L].copy_(srcs[i % 3])                   # from_data: the serialized bytes into the shard buffer
            c.compute_parity()
        us_both = _time_us(torch, both, 9)
        # round 3: from_data + compute_parity as ONE pass over the serialized bytes (smr_rs_from_data_encode) -- checked here
        # against the two steps on the first batch, then timed
        ref = cws[0].buf.clone()
        RSCodewordBatch.from_data_and_encode(srcs[0], 3, 2, out=cws[0])
        assert torch.equal(cws[0].buf, ref), "one-pass from_data + encode differs from from_data, compute_parity"
        del ref
        us_one = _time_us(torch, lambda i: RSCodewordBatch.from_data_and_encode(srcs[i % 3], 3, 2, out=cws[i % 3]), 9)
        sl = cws[0].shard_len
        out.append({"value_bytes": size, "codewords_per_launch": n, "encode_GiBps": n * L / 2**30 / (us_enc * 1e-6),
                    "encode_frac": n * 5 * sl / (us_enc * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "from_data_and_encode_GiBps": n * L / 2**30 / (us_one * 1e-6),
                    "from_data_and_encode_frac": n * (L + 5 * sl) / (us_one * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "from_data_then_encode_two_steps_GiBps": n * L / 2**30 / (us_both * 1e-6)})
        del srcs, cws
    return out


def rs_leg(torch, dev, run_cpu, cpu_seconds):
    """BASELINE config 4: 16384 codewords, 4 KiB values -> bincode(String) L = 4099, RS(3,2)."""
    from summerset_amd import RSCodewordBatch
    n, L = 16384, 4099
    # Config 4's tick is 16384 codewords (67 MB in, 45 MB out): re-encoding ONE such batch would sit in the 256 MiB
    # Infinity Cache (MI355X_MICROARCH.md: FETCH_SIZE counts L3 hits), so the timed loop rotates NB distinct batches whose
    # inputs + outputs total NB x 112 MB > 3 x the L3 -- every launch streams from HBM and writes to HBM.
    NB = 8
    batches = [RSCodewordBatch.from_data(torch.randint(0, 256, (n, L), dtype=torch.uint8, device=dev), 3, 2) for _ in range(NB)]
    data = batches[0].get_data()
    out = {}
    for name, lut in (("xtime", False), ("lut", True)):
        for cw in batches:
            cw.compute_parity(lut=lut)
        torch.cuda.synchronize()
        iters = 48
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            batches[i % NB].compute_parity(lut=lut)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        cw = batches[0]
        alg = n * 5 * cw.shard_len                       # 5 * ceil(L/3) bytes per codeword
        out[name] = {"ms_per_launch": ms, "payload_GiBps": n * L / 2**30 / (ms * 1e-3),
                     "achieved_GBps": alg / (ms * 1e-3) / 1e9, "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # one launch over 65536 codewords (448 MB of traffic by itself), and the single hot batch for comparison
    big = RSCodewordBatch.from_data(torch.randint(0, 256, (65536, L), dtype=torch.uint8, device=dev), 3, 2)
    us_big = _time_us(torch, lambda i: big.compute_parity(), 12)
    us_hot = _time_us(torch, lambda i: batches[0].compute_parity(), 48)
    del big
    t_rs = pmc_traffic("smr::rs_matmul_xtime<2, 4>", pick="smallest")     # the probe's 16384-codeword launches (it also runs 65536 for the calibration)
    # a record WITHOUT `by_grid` (rebuilt by hand from a log) says nothing about which launch size its figure is --
    # rescaling it "from a 65536-codeword launch" put 0.26 where 1.04 was true.  No `by_grid`, no traffic figure.
    if t_rs is not None and not (pmc_file() or {}).get("kernels", {}).get("smr::rs_matmul_xtime<2, 4>", {}).get("by_grid"):
        t_rs = None
    same_size = bool(t_rs and t_rs.get("grids", 1) > 1)
    res = {"workload": "RS(3,2) GF(2^8) encode, 16384 codewords x L=4099 B (4 KiB value as bincode String) per launch, "
                       "%d distinct batches in rotation (%.0f MB in + out: beyond the 256 MiB L3)" % (NB, NB * n * 5 * cw.shard_len / 1e6),
           "value": out["xtime"]["payload_GiBps"], "unit": "GiB/s payload",
           "roofline": {"bound": "hbm", "achieved": out["xtime"]["achieved_GBps"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": out["xtime"]["frac"], "kernel": "rs_matmul_xtime<2, 4>",
                        "alg_bytes_per_launch": n * 5 * cw.shard_len, "avg_launch_us": out["xtime"]["ms_per_launch"] * 1e3,
                        "traffic": (t_rs["hbm_bytes_per_launch"] if same_size else t_rs["hbm_bytes_per_launch"] / 65536 * n) if t_rs else None,
                        "traffic_note": "no PMC figure keyed by grid size in the committed record" if not t_rs else
                                        ("PMC bytes of the probe's %d-codeword launches (same grid size as this leg's)" % n) if same_size else
                                        ("PMC bytes of the record's only launch size (65536 codewords) scaled to this launch's %d codewords" % n)},
           "lut_variant_GiBps": out["lut"]["payload_GiBps"],
           "one_launch_65536_codewords": {"avg_launch_us": us_big, "frac": 65536 * 5 * cw.shard_len / (us_big * 1e-6) / 1e9 / HBM_PEAK_GBS},
           "single_hot_batch_L3_assisted": {"avg_launch_us": us_hot, "frac": n * 5 * cw.shard_len / (us_hot * 1e-6) / 1e9 / HBM_PEAK_GBS},
           "rse_bench_sweep": rse_sweep(torch, dev)}
    if run_cpu:
        import multiprocessing as mp
        host = data[:2048].cpu().numpy().reshape(-1).copy()
        v1, n1 = _rs_cpu_run((host, L, cpu_seconds / 6))
        cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 32)
        vn = v1
        if cores > 1:
            with mp.get_context("fork").Pool(cores) as pool:
                vn = sum(r[0] for r in pool.map(_rs_cpu_run, [(host, L, cpu_seconds / 6)] * cores))
        res["cpu_baseline"] = {"value": vn, "unit": "GiB/s payload", "cores": cores, "kind": "port", "single_core_value": v1,
                               "sample": "oracle/rs_oracle.c table encoder (padding copy included) on 2048 codewords of "
                                         "L=4099, repeated: %d single-threaded processes side by side (rates summed), "
                                         "and one process (%d codewords)" % (cores, n1)}
    return res


def _on_all_cores(fn, seconds=3.0, **kw):
    """a one-thread CPU baseline `fn(seconds=..., **kw)` -> its object, and next to it the same thing as one single-threaded
    process per host core side by side ((a) one thread, (b) nproc): groups are independent, so that is how a
    multi-core host runs them; rates summed, `cores` = the processes really used"""
    import multiprocessing as mp
    one = fn(seconds=seconds, **kw)
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 32)
    out = dict(one, single_core_value=one["value"], cores=1)
    if cores > 1:
        with mp.get_context("fork").Pool(cores) as pool:
            res = pool.starmap(_call_kw, [(fn, dict(kw, seconds=seconds))] * cores)
        out["value"], out["cores"] = sum(r["value"] for r in res), cores
        out["sample"] = "%d single-threaded processes side by side (rates summed), each: %s; one process alone: %.4g %s" % (
            cores, one["sample"], one["value"], one["unit"])
    return out


def _call_kw(fn, kw):
    return fn(**kw)


def rspaxos_cpu_baseline(G=1024, L=4113, seconds=3.0):
    """config 4 on the CPU oracles (one core): per tick the RS(3,2) encode of G request batches of L bytes (oracle/rs_oracle.c)
    and the steady tick of five RspOracle replicas in the numpy-staged closed loop (summerset_amd/rsp_cluster.tick)"""
    from oracle import oracle as O
    from summerset_amd import rsp_cluster, workloads
    R = 5
    orcs = [O.RspOracle(G, R, me=r, W=64, fault_tolerance=1) for r in range(R)]
    for o in orcs:
        o.preset_leader(0)
    rng = np.random.default_rng(0x5EED5EED)
    data = rng.integers(0, 256, (G, L), dtype=np.uint8)
    lost = {k: v.astype(bool) for k, v in workloads.config4_loss(rng, G).items()}
    leader = np.zeros(G, np.uint8)
    spent, t, committed = 0.0, 0, 0
    while spent < seconds:
        val = workloads.config4_tokens(G, t).view(np.uint32)
        t0 = time.perf_counter()
        O.rs_encode_batch(3, 2, data, L, L, G)
        log = rsp_cluster.tick(orcs, val, leader, drop=lost, heartbeat=(t % 4 == 3))
        spent += time.perf_counter() - t0
        committed += sum(int(e["committed"].sum()) for e in log if e["kind"] == "commit")
        t += 1
    return {"value": committed / spent, "unit": "slots/s", "cores": 1, "kind": "port",
            "sample": "oracle/rs_oracle.c + oracle/rsp_oracle.c, %d ticks of %d groups x 5 replicas (RS(3,2) of L = %d per group per tick), "
                      "one thread, %.1f s" % (t, G, L, spent)}


def raft_cpu_baseline(S=32, G=4096, seconds=3.0):
    """oracle/raft_oracle.c on the Raft leg's stream (one core): appends + four replies per group per tick"""
    from oracle import oracle as O
    R, W = 5, 512
    o = O.RaftOracle(G, R, W, leader_id=0, term=2)
    rng = np.random.default_rng(0x5EED5EED)
    n_new = np.full(G, S, np.uint32)
    spent, t = 0.0, 0
    while spent < seconds:
        last = 1 + S * (t + 1) - 1
        lag = rng.integers(0, 4, (R, G))
        u = rng.random((R, G))
        flags = (u >= 0.05).astype(np.uint8)
        term = np.full((R, G), 2, np.uint64)
        term[(u >= 0.05) & (u < 0.055)] = 1
        flags[(u >= 0.055) & (u < 0.06)] |= 2
        end_slot = np.maximum(last - lag, 0).astype(np.uint32)
        ct = np.full((R, G), 2, np.uint64)
        cs = np.maximum(end_slot.astype(np.int64) - 1, 1).astype(np.uint32)
        t0 = time.perf_counter()
        o.append(n_new)
        o.handle_replies(term, end_slot, flags, ct, cs)
        spent += time.perf_counter() - t0
        t += 1
    return {"value": o.total_commits() / spent, "unit": "slots/s", "cores": 1, "kind": "port",
            "sample": "oracle/raft_oracle.c, %d ticks of %d groups, one thread, %.1f s" % (t, G, spent)}


def epaxos_cpu_baseline(G=4096, seconds=3.0):
    """oracle/ep_oracle.c on the EPaxos leg's stream (one core): propose + the four PreAcceptReplies"""
    from oracle import oracle as O
    R, W, K = 5, 32, 64
    o = O.EpOracle(G, R, me=0, W=W, n_keys=K)
    rng = np.random.default_rng(0x5EED5EED)
    zipf = 1.0 / np.arange(1, K + 1) ** 0.99
    zipf /= zipf.sum()
    flags = np.ones((R, G), np.uint8)
    flags[0] = 0
    ballot = np.full((R, G), 1, np.uint64)
    spent, t, committed = 0.0, 0, 0
    while spent < seconds:
        key = rng.choice(K, G, p=zipf).astype(np.uint8)
        ex = rng.random((R, G)) < 0.1
        t0 = time.perf_counter()
        m = o.propose(key)
        spent += time.perf_counter() - t0
        seq = np.ascontiguousarray(np.broadcast_to(m["seq"], (R, G)) + ex.astype(np.uint64))
        deps = np.broadcast_to(m["deps"], (R, R, G)).copy()
        d1 = deps[:, 1, :]
        deps[:, 1, :] = np.where(ex, np.where(d1 == 0xFFFFFFFF, 1, d1 + 1), d1)
        t0 = time.perf_counter()
        r = o.handle_pre_accept_replies(m["col"], ballot, seq, np.ascontiguousarray(deps.astype(np.uint32)), flags)
        spent += time.perf_counter() - t0
        committed += int((r["decision"] == 3).sum())
        t += 1
    return {"value": committed / spent, "unit": "fast-path commits/s (propose + reply handling)", "cores": 1, "kind": "port",
            "sample": "oracle/ep_oracle.c, %d ticks of %d groups, one thread, %.1f s" % (t, G, spent)}


def repnothing_leg(n_ops=200000):
    """BASELINE config 1 (CPU only, plumbing): RepNothing, 1 group x 1 replica, Put{key = "k%07d" (i mod 5), value =
    1024 B alnum}, batches of 1, through the C-ABI host path (csrc/rep_nothing.hip); a bounded sample of the
    1 000 000-op configuration."""
    from summerset_amd import RepNothingReplica
    r = RepNothingReplica()
    rng = np.random.default_rng(0x5EED5EED)
    alnum = np.frombuffer(b"0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", np.uint8)
    vals = [alnum[rng.integers(0, 62, 1024)].tobytes() for _ in range(64)]
    keys = [("k%07d" % i).encode() for i in range(5)]
    t0 = time.perf_counter()
    for i in range(n_ops):
        r.handle_req_batch([(1, i, ("put", keys[i % 5], vals[i & 63]))])
    dt = time.perf_counter() - t0
    s = r.stats()
    assert s["executed"] == n_ops and s["keys"] == 5
    return {"workload": "RepNothing, 1 group x 1 replica, %d Puts of 1024 B on 5 keys, batches of 1 (host only)" % n_ops,
            "value": n_ops / dt, "unit": "ops/s", "cores": 1,
            "note": "Python ctypes call per batch included; WAL accounted (%d bytes), not written" % s["wal_offset"]}


def epaxos_cluster_leg(torch, dev, ticks=10):
    """BASELINE config 5 as written: EPaxos, 65 536 groups x 5 replicas, optimized quorums, EVERY replica proposes one
    instance per group per tick on Zipf(0.99) keys of 64 -- five replica objects in the closed loop of
    summerset_amd/ep_cluster.py (PreAccept fan-out, replies, fast / slow decision, Accept rounds, CommitNotices), every
    message a device tensor between the handlers, dependency-graph execution on."""
    from summerset_amd import EPaxosReplicaGroup, ep_cluster
    G, R, W, K = 65536, 5, 32, 64
    EXEC = os.environ.get("SMR_EPC_EXECUTE", "1") != "0"            # (experiments only: the tick without dependency-graph execution)
    reps = [EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K, execute=EXEC) for r in range(R)]
    rng = np.random.default_rng(0x5EED5EED)
    zipf = 1.0 / np.arange(1, K + 1) ** 0.99
    zipf /= zipf.sum()
    keys = [[torch.from_numpy(rng.choice(K, G, p=zipf).astype(np.uint8)).to(dev) for _ in range(R)] for _ in range(ticks + 2)]
    for t in range(2):
        ep_cluster.tick(reps, keys[t])
    torch.cuda.synchronize()
    committed = torch.zeros((), dtype=torch.int64, device=dev)
    slow = torch.zeros((), dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    for t in range(2, ticks + 2):
        for o in ep_cluster.tick(reps, keys[t]):
            committed += o["committed"].sum()
            slow += (o["decision"] == 2).sum()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ex = sum(int(r.exec_dump()["counters"][0]) for r in reps) if EXEC else 0
    line = {"workload": "EPaxos closed loop, %d groups x 5 replicas, every replica proposes 1 instance per group per tick (Zipf(0.99) keys "
                        "of 64), execution on; five replica objects on one GPU, messages stay on the device" % G,
            "value": int(committed.item()) / dt, "unit": "instances committed/s (handler calls of the Python driver included)",
            "ms_per_tick": dt / ticks * 1e3, "slow_path_fraction": int(slow.item()) / max(int(committed.item()), 1),
            "handler_calls_per_tick": R + 2 * R * (R - 1) + 2 * R, "commands_executed": ex}
    # the same loop as ONE C-ABI call per tick (smr_ep_cluster_tick): as ONE launch -- a block is the five replicas of 64 groups, the
    # handlers are steps of that kernel -- and (`per_handler_launches`, round 2's path) as the handler kernels launched back to
    # back by the library.  Own replicas each; the tick's output arrays are the caller's and are reused.
    # ... and (`phase_major`) with the command leaders' steps PHASE BY PHASE instead of leader by leader: another legal delivery
    # order of the same messages, in which all five replicas of a group work in every step of the kernel.  Its reference is the
    # Python loop run in that order on a fresh set of replicas (same decisions and commits as the default order; the
    # executors' attempt order differs, so their counters may).
    del reps
    reps_pm = [EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K, execute=EXEC) for r in range(R)]
    committed_pm = torch.zeros((), dtype=torch.int64, device=dev)
    for t in range(ticks + 2):
        for o in ep_cluster.tick(reps_pm, keys[t], phase_major=True):
            if t >= 2:
                committed_pm += o["committed"].sum()
    ex_pm = sum(int(r.exec_dump()["counters"][0]) for r in reps_pm) if EXEC else 0
    del reps_pm
    for name, per_handler, pm in (("one_call_per_tick", False, False), ("one_call_per_tick_per_handler_launches", True, False),
                                  ("one_call_per_tick_phase_by_phase", False, True)):
        try:
            reps2 = [EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K, execute=EXEC) for r in range(R)]
            fused = ep_cluster.EPaxosCluster(reps2, per_handler_launches=per_handler, phase_major=pm)
            outs = fused.new_outputs(dev)
            committed_all = torch.zeros((R, G), dtype=torch.uint8, device=dev)     # the five leaders' `committed` arrays as rows of ONE tensor:
            for s_ in range(R):                                                     # the tick's commit count is one reduction, not five + a stack
                outs[s_]["committed"] = committed_all[s_]
            for t in range(2):
                fused.tick(keys[t], out=outs)
            torch.cuda.synchronize()
            c2 = torch.zeros((), dtype=torch.int64, device=dev)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * ticks)]
            t0 = time.perf_counter()
            for t in range(2, ticks + 2):
                ev[2 * (t - 2)].record()
                fused.tick(keys[t], out=outs)
                ev[2 * (t - 2) + 1].record()
                c2 += committed_all.sum()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            tick_us = sorted(ev[2 * i].elapsed_time(ev[2 * i + 1]) * 1e3 for i in range(ticks))
            n_inst = int(c2.item())
            leg = {"entry_point": "smr_ep_cluster_tick", "launches_per_tick": 115 if per_handler else 1,
                   "value": n_inst / dt2, "unit": "instances committed/s", "ms_per_tick": dt2 / ticks * 1e3,
                   "tick_us_device_median": tick_us[len(tick_us) // 2], "tick_us_device_min": tick_us[0],
                   "same_commits_as_the_driver_loop": n_inst == int((committed_pm if pm else committed).item()),
                   "commands_executed": sum(int(r.exec_dump()["counters"][0]) for r in reps2) if EXEC else 0}
            if pm:
                leg["order"] = "the leaders' steps phase by phase (smr_ep_cluster_set_mode bit 1); reference = ep_cluster.tick(.., phase_major=True)"
                leg["same_commands_executed_as_the_driver_loop"] = leg["commands_executed"] == ex_pm
                st = fused.batch_stats()                      # round 6: a lane's 4 PreAccepts / 4 CommitNotices as one batched step each
                leg["batch_stats"] = dict(st, lanes_per_phase=(ticks + 2) * R * G)
            if not per_handler:
                # <= 370 B per instance for the tally (replies read, instance read / written); the tick as a whole
                # -- proposals, 4 PreAccepts, the tally, 4 CommitNotices, execution per instance -- has no per-unit figure there,
                # so this is the tally's figure over the WHOLE tick's time: a lower bound on what the kernel moves
                alg = 370.0 * R * G
                us = leg["tick_us_device_median"]
                # (phase by phase, round 6: the batched tick kernel + the launch that takes the lanes it lists one by one -- both
                #  inside the timed call, both in `traffic`)
                kern = ("ep_cluster_tick_pm_kernel<5> + ep_cluster_commit_one_by_one_kernel<5> (the whole tick: 2 launches)" if pm else
                        "ep_cluster_tick_kernel<5> (the whole tick: 1 launch)")
                traffic = (_sum_traffic("smr::ep_cluster_tick_pm_kernel<5>", "smr::ep_cluster_commit_one_by_one_kernel<5>") if pm else
                           _leg_traffic("smr::ep_cluster_tick_kernel<5, false>"))
                if pm:
                    leg["launches_per_tick"] = 2
                leg["roofline"] = {"bound": "hbm", "kernel": kern, "achieved": alg / us / 1e3,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / us / 1e3 / HBM_PEAK_GBS, "alg_bytes_per_launch": alg,
                                   "avg_launch_us": us, "traffic": traffic, "traffic_source": PMC_NOTE,
                                   "note": "alg bytes = SURVEY 8(d)'s <= 370 B per instance x 5 x 65536 instances per tick (the tally's figure; the "
                                           "tick also runs 5 proposals, 20 PreAccepts, 20 CommitNotices and the execution walks per group)"}
            line[name] = leg
            fused.close()
            del reps2, fused, outs
        except Exception as e:                         # noqa: BLE001
            line[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            sys.stderr.write("bench.py: epaxos_cluster %s FAILED: %s: %s\n" % (name, type(e).__name__, e))
    return line


def rspaxos_leg(torch, dev, ticks=48, warmup=8):
    """BASELINE config 4 on the RSPaxos ENGINE (csrc/rsp_engine.hip; rounds 1-2 timed a MultiPaxos-engine stand-in here):
    16 384 groups x 5 replica objects, f = 1, leader 0, one Put of a 4 KiB value per group per tick.  Per tick, all on the
    device (summerset_amd/rsp_cluster.SteadyLoop): from_data + RS(3,2) encode of the tick's 16 384 request batches in ONE
    pass (smr_rs_from_data_encode, L = 4113 = bincode(ReqBatch) of one 4 KiB Put) out of one of NB rotating source / codeword
    buffer pairs (NB x 180 MB: beyond the 256 MiB L3) which ALSO fills every replica's shard store (the fan-out of
    rspaxos/request.rs:127-142: the co-located stand-in for the Accepts' payload); the leader's handle_req_batch; the four followers' handle_msg_accept
    with the ONE shard they hold; the leader's AcceptReply tally at majority + f with the shard-availability gate of the
    commit-bar run behind it (rspaxos/durability.rs:140-160: a replica holding fewer than d shards of a slot cannot execute
    it -- a follower holds one of five -- so that gate is on the timed path at every replica); every 4th tick the
    Heartbeats.  <= 1 AcceptReply of 4 lost per slot (the threshold is 4 of 5 and nothing is retransmitted).
    The tick is captured into ONE HIP graph of NB ticks and replayed (HIP graphs instead of per-launch host calls: a tick is
    ~14 launches of a few microseconds each); the eager loop is timed beside it."""
    from summerset_amd import RSCodewordBatch, workloads
    c4 = workloads.CONFIG4
    G, R, W, L, NB, H = c4["G"], c4["R"], c4["W"], c4["L"], c4["n_buffers"], c4["H"]
    # cluster, loss masks, tokens and the tick itself from summerset_amd/workloads.py -- what
    # tests/test_baseline_configs_gpu.py::test_config3_rspaxos_one_launch_tick_16384_groups holds against the oracle
    reps, loop = workloads.config4_cluster(G, W, c4["ft"], one_launch=os.environ.get("SMR_RSP_CALL_BY_CALL") is None)
    rng = np.random.default_rng(0x5EED5EED)
    srcs = [torch.randint(0, 256, (G, L), dtype=torch.uint8, device=dev) for _ in range(NB)]
    sl = -(-L // 3)                                                      # shard_len = ceil(L / d)
    masks = [{k_: torch.from_numpy(v).to(dev) for k_, v in workloads.config4_loss(rng, G).items()} for k in range(NB)]   # ~30 % of the slots lose ONE of their four replies
    vals = [torch.from_numpy(workloads.config4_tokens(G, j)).to(dev) for j in range(8 * NB)]   # the ticks' batch tokens: inputs, resident before the timed region
    n_tick = [0]                                                         # (round 3 made them with four torch kernels INSIDE every tick: ~25 us of a 0.125 ms tick, profiles/round3/r4w)

    def one_tick(k, hb):
        val = vals[(n_tick[0] // NB * NB + k) % len(vals)]
        n_tick[0] += 1
        return workloads.config4_tick(loop, k, srcs[k], val, masks[k], hb)[0]   # encode straight into the holders' stores (each shard written once), then the tick (one launch)

    def commits():
        return int(reps[0].dump()["counters"][0])

    for t in range(warmup):
        one_tick(t % NB, t % H == H - 1)
    torch.cuda.synchronize()
    line = {"workload": "RSPaxos engine (f = 1), %d groups x 5 replica objects on one GPU, one 4 KiB Put per group per tick: from_data + RS(3,2) "
                        "encode in one pass (L = %d, %d rotating buffer pairs = %.0f MB), leader's handle_req_batch, shard fan-out to the "
                        "followers' stores, 4 x handle_msg_accept (one shard each), AcceptReply tally at majority + 1 with the "
                        "shard-availability gate, Heartbeats every %d ticks; <= 1 of 4 replies lost per slot"
                        % (G, L, NB, NB * (G * L + G * 5 * sl) / 1e6, H),
            "engine": "csrc/rsp_engine.hip (rsp_cluster_tick_kernel: the tick's handlers in one launch, messages through LDS%s) + rs_from_data_xtime<2, 4> writing every shard once into its holder's store"
                      % ("" if loop._cl is not None else " -- here: SMR_RSP_CALL_BY_CALL, one launch per handler"),
            "launches_per_tick": 2 if loop._cl is not None else "~15"}
    # eager: one host call per handler
    c0 = commits()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for t in range(ticks):
        one_tick(t % NB, t % H == H - 1)
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_c = commits() - c0
    line["eager"] = {"value": n_c / dt, "unit": "slots/s", "ms_per_tick": dt / ticks * 1e3, "device_ms_per_tick": e0.elapsed_time(e1) / ticks,
                     "committed_per_tick": n_c / ticks, "rs_payload_GiBps": G * L * ticks / 2**30 / dt}
    # the same NB ticks as one HIP graph
    try:
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for k in range(NB):                                          # (a pass on the capture stream first: allocator warm-up)
                one_tick(k, k == NB - 1)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for k in range(NB):
                one_tick(k, k == NB - 1)
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        c0 = commits()
        reps_ = max(ticks // NB, 4)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps_):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n_c = commits() - c0
        nt = reps_ * NB
        ms = e0.elapsed_time(e1) / nt
        alg = G * (L + 5 * sl) + G * (52 + 33)   # the encode pass (L read, the five shards written ONCE, into their holders' stores) + the tally's 8(d) bytes
        line["graph"] = {"value": n_c / dt, "unit": "slots/s", "ms_per_tick": dt / nt * 1e3, "device_ms_per_tick": ms, "ticks_per_graph": NB,
                         "committed_per_tick": n_c / nt, "rs_payload_GiBps": G * L * nt / 2**30 / dt}
        t_enc, t_tick = _leg_traffic("smr::rs_from_data_xtime<2, 4>"), _leg_traffic("smr::rsp_cluster_tick_kernel")
        alg_8d = G * 5 * sl + G * (52 + 33)        # 5 * ceil(L / 3) per codeword (3 shards in, 2 out) + the tally's bytes
        line["roofline"] = {"bound": "hbm", "kernel": "the whole tick (one HIP graph of %d ticks): rs_from_data_xtime<2, 4> (encode + fan-out) + rsp_cluster_tick_kernel" % NB,
                            "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "frac_on_survey_8d_bytes": alg_8d / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "survey_8d_bytes_per_launch": alg_8d,
                            "alg_bytes_per_launch": alg, "avg_launch_us": ms * 1e3, "traffic": (t_enc + t_tick) if (t_enc and t_tick) else None, "traffic_source": PMC_NOTE,
                            "note": "alg bytes per tick = 16384 x (L read + 5 shard_len written: every shard once, straight into its holder's "
                                    "store -- the leader's codeword is a view of the stores) for from_data + encode + fan-out, 85 B per slot for the "
                                    "tally (SURVEY 8(d)); round 3 wrote the codeword AND the stores (L + 10 shard_len)"}
        best = "graph" if line["graph"]["value"] >= line["eager"]["value"] else "eager"   # (two launches per tick: a graph of four ticks saves little)
        line["value"], line["unit"], line["ms_per_tick"], line["value_is"] = line[best]["value"], "slots/s", line[best]["ms_per_tick"], best
        line["rs_payload_GiBps"] = line[best]["rs_payload_GiBps"]
    except Exception as e:                         # noqa: BLE001
        line["graph"] = {"error": "%s: %s" % (type(e).__name__, e)}
        line["value"], line["unit"], line["ms_per_tick"] = line["eager"]["value"], "slots/s", line["eager"]["ms_per_tick"]
        line["rs_payload_GiBps"] = line["eager"]["rs_payload_GiBps"]
    # the encode pass alone, in rotation: from_data + compute_parity as two steps against the one-pass kernel
    us_two = _time_us(torch, lambda i: RSCodewordBatch.from_data(srcs[i % NB], 3, 2).compute_parity(), 16)
    cws = [RSCodewordBatch(G, L, 3, 2, device=dev, zero=False) for _ in range(NB)]
    us_one = _time_us(torch, lambda i: RSCodewordBatch.from_data_and_encode(srcs[i % NB], 3, 2, out=cws[i % NB]), 32)
    us_st = _time_us(torch, lambda i: loop.encode_stores(srcs[i % NB], slot=i % NB), 32)
    line["from_data_and_encode"] = {"one_pass_us": us_one, "one_pass_payload_TiBps": G * L / 2**40 / (us_one * 1e-6),
                                    "one_pass_frac": G * (L + 5 * sl) / (us_one * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                    "into_stores_us": us_st, "into_stores_frac": G * (L + 5 * sl) / (us_st * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                    "two_step_us": us_two, "two_step_payload_TiBps": G * L / 2**40 / (us_two * 1e-6),
                                    "note": "one pass moves L + 5 shard_len bytes per codeword (read once, d + p shards written); "
                                            "two steps = copy into a zeroed codeword buffer, then smr_rs_encode"}
    return line


def _payload_leg_traffic(name="rspaxos_payload"):
    """HBM bytes per tick of the payload store's kernels from the committed PMC passes over this very leg, or (None, None): the
    sum over every ps_* / craft_* kernel of (bytes per launch x launches per tick), launches per tick = the kernel's launches in
    the profiled run / the run's ticks (recorded in the file by tools/final_record.sh)."""
    for f in ("t3z_pmc_traffic_%s_leg.json" % name, "t2z_pmc_traffic_%s_leg.json" % name, "t1z_pmc_traffic_%s_leg.json" % name, "r9z_pmc_traffic_%s_leg.json" % name, "r8z_pmc_traffic_%s_leg.json" % name, "r8m_pmc_traffic_%s_leg.json" % name, "r7g_pmc_traffic_payload_leg.json" if name == "rspaxos_payload" else None):
        if not f:
            continue
        try:
            with open(os.path.join(ROOT, "profiles", f)) as fh:
                d = json.load(fh)
            k = d["kernels"]
            if f.startswith("r7g"):                                  # (round 4's file: one follow per replica, five plan + five byte launches)
                return (k["smr::ps_put_kernel<3>"]["hbm_bytes_per_launch"] + 5 * k["smr::ps_plan_kernel"]["hbm_bytes_per_launch"]
                        + 5 * k["smr::ps_bytes_kernel"]["hbm_bytes_per_launch"]), "profiles/" + f
            put = next(v for n, v in k.items() if "::ps_put_" in n)       # (ps_put_kernel / round 6: ps_put_deliver_kernel)
            ticks = put["launches"]                                  # one put per tick
            tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for n, v in k.items() if "::ps_" in n or "::craft_" in n)
            return tot / ticks, "profiles/" + f
        except (OSError, KeyError, ValueError, StopIteration):
            continue
    return None, None


def rspaxos_payload_leg(torch, dev, ticks=32, warmup=6):
    """BASELINE config 4 with the shard BYTES in the product's payload store (csrc/rsp_payload.hip, VERDICT r3 missing #3): the
    `rspaxos` leg's engines and one-launch tick, and behind every tick the leader's `smr_rsp_pstore_put` (from_data + RS(3,2)
    encode of the 16 384 batches into the ring rows of its REQS plane) and one `smr_rsp_pstore_follow` per replica (the leader's
    voted shard; every follower's shard out of the leader's store into its REQS and VOTED planes) -- the general, mask-driven
    path that also serves leader changes, where the `rspaxos` leg writes the tick's shards into flat per-tick stores.  The
    leg ends with the checks a host can make without the oracle: nothing unsatisfied, every row's token and mask = the engine's,
    the last row's parity verified by the RS kernels."""
    from summerset_amd import _lib, workloads
    from summerset_amd.rsp_payload import REQS
    c4 = workloads.CONFIG4
    G, R, L, NB, H, W = c4["G"], c4["R"], c4["L"], c4["n_buffers"], c4["H"], workloads.PAYLOAD_W
    # cluster, stores and the tick itself from summerset_amd/workloads.py -- what
    # tests/test_baseline_configs_gpu.py::test_config3_payload_store_16384_groups holds against the oracles
    reps, loop, stores = workloads.config4_payload_cluster(G, W, c4["ft"], L)
    rng = np.random.default_rng(0x5EED5EED)
    srcs = [torch.randint(0, 256, (G, L), dtype=torch.uint8, device=dev) for _ in range(NB)]
    masks = [{k_: torch.from_numpy(v).to(dev) for k_, v in workloads.config4_loss(rng, G).items()} for k in range(NB)]
    vals = [torch.from_numpy(workloads.config4_tokens(G, j)).to(dev) for j in range(warmup + 2 * ticks + 8)]
    ones = torch.ones(G, dtype=torch.int32, device=dev)
    slots = [torch.full((G,), j, dtype=torch.int32, device=dev) for j in range(len(vals))]   # every group appends every tick: slot = tick
    n = [0]

    def one_tick(_i=0, engine=True, bytes_=True):
        j = n[0]
        n[0] += 1
        workloads.config4_payload_tick(reps, loop, stores, slots[j], srcs[j % NB], vals[j], masks[j % NB], j % H == H - 1, ones, engine, bytes_)
    for _ in range(warmup):
        one_tick()
    # ~17 host calls per tick at 10-20 us each: a 12 ms device-side sleep in front lets the host queue the whole region first
    us = _time_us(torch, one_tick, ticks, sleep_cycles=24_000_000)
    sl = -(-L // 3)
    # put: L read, 5 shards written; 4 followers x one shard WRITTEN -- round 6: by the put launch, out of its registers (until then
    # read back out of the leader's row: + 4 shard_len); every vote is an alias
    moved = G * (L + 5 * sl + 4 * sl)
    c = [st.counters() for st in stores]
    ok = all(x["unsatisfied"] == 0 for x in c)
    for q in range(R):                             # every ring cell of both planes holds what the engine says it holds
        d = reps[q].dump()
        for plane, (kt, km) in enumerate((("s_val", "s_mask"), ("s_vval", "s_vmask"))):
            sd = stores[q].dump(plane)
            ok = ok and bool(np.array_equal(sd["avail"], d[km]) and np.array_equal(sd["tok"][d[km] != 0], d[kt][d[km] != 0]))
            if plane == 1:                         # ... and no vote of this run was stored a second time
                ok = ok and bool(np.array_equal(stores[q].voted_alias(), sd["avail"]))
    v = torch.zeros(G, dtype=torch.uint8, device=dev)
    last = (n[0] - 1) & (W - 1)
    _lib.check(_lib.load().smr_rs_verify(stores[0].plane_ptr(REQS) + last * stores[0].row_stride, sl, stores[0].shard_stride,
                                         stores[0].group_stride, G, 3, 2, v.data_ptr(), _lib.stream_ptr(None)))
    ok = ok and bool(v.all().item())
    us_engine = _time_us(torch, lambda i: one_tick(bytes_=False), 12)   # the engines' tick alone (after the checks: the stores stay behind from here)
    us_bytes = max(us - us_engine, 1e-3)
    return {"workload": "config 4's engines + one-launch tick, and the tick's shard bytes through the payload store: put (from_data + RS(3,2) "
                        "encode into the ring, %d groups x L = %d) + the leader's follow + one follow_many for the four followers (window %d, two planes)" % (G, L, W),
            "value": G / (us * 1e-6), "unit": "slots/s", "ms_per_tick": us * 1e-3, "engine_only_ms_per_tick": us_engine * 1e-3,
            "bytes_path_ms_per_tick": us_bytes * 1e-3, "rs_payload_GiBps": G * L / 2**30 / (us * 1e-6),
            "launches_per_tick": {"engine": 1, "bytes": 4}, "shards_delivered_by_the_put_launch": sum(st.delivered() for st in stores),
            "roofline": {"bound": "hbm", "kernel": "ps_put_deliver_kernel<3> (the leader's five shards and each follower's one) + ps_plan_kernel + ps_bytes_plan_many_kernel + ps_bytes_many_kernel (metadata: nothing is left to copy in a steady tick)", "achieved": moved / (us_bytes * 1e-6) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": moved / (us_bytes * 1e-6) / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_launch": moved,
                         "survey_8d_bytes_per_launch": G * (5 * sl + 85),
                         "frac_on_survey_8d_bytes": G * (5 * sl + 85) / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,   # the WHOLE tick on SURVEY 8(d)'s bytes
                         "avg_launch_us": us_bytes, "traffic": _payload_leg_traffic("rspaxos_payload")[0],
                         "traffic_source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over this leg; per tick = every ps_* launch of a tick)"
                                           % _payload_leg_traffic("rspaxos_payload")[1],
                         "note": "per tick (tick with the stores minus the engines' tick alone); bytes the path has to move: L read + 5 shard_len "
                                 "written by put for the leader and one shard per follower written by the same launch (round 6; rounds 5b-6a read "
                                 "it back out of the leader's row: 359 MB; a vote is an alias of the reqs row's shard, not a second copy: rounds "
                                 "4-5a moved 584 MB here)"},
            "counters": {k: sum(x[k] for x in c) for k in c[0]}, "verified": ok}


def craft_payload_leg(torch, dev, ticks=24, warmup=6, G=16384, L=4113, time_us=None):
    """CRaft with its shard BYTES in the product's store (VERDICT r4 missing #3; csrc/rsp_payload.hip smr_craft_pstore_*): config 4's
    shape on the Raft fork -- 16 384 groups x 5 replicas, one 4 KiB batch appended per group per tick, balanced assignment (every
    follower is sent its own shard, craft/request.rs:86-100).  Per tick: the leader's append + `put` (from_data + RS(3,2) encode of
    the 16 384 batches into its log's rows) + its `follow`; per follower the AppendEntries out of the leader's log and
    `handle_msg_append_entries`; ONE `follow_many` for the four followers; the replies' match-index quorum at the leader.  The leg
    ends with the checks a host can make without the oracle: every follower holds exactly its own shard of every entry, the last
    row's parity verifies, nothing unsatisfied."""
    from summerset_amd import _lib, workloads
    R, W, NB = workloads.CRAFT_PAYLOAD["R"], workloads.CRAFT_PAYLOAD["W"], 3
    time_us = time_us or _time_us                             # (tests/test_craft_payload.py runs the leg's loop on the emulator)
    # cluster, stores and the tick itself from summerset_amd/workloads.py -- what
    # tests/test_baseline_configs_gpu.py::test_craft_payload_store_16384_groups holds against the oracles
    reps, stores, bufs = workloads.craft_payload_cluster(G, W, L, device=dev)
    srcs = [torch.randint(0, 256, (G, L), dtype=torch.uint8, device=dev) for _ in range(NB)]
    slots = [torch.full((G,), 1 + j, dtype=torch.int32, device=dev) for j in range(warmup + 2 * ticks + 8)]
    n = [0]

    def one_tick(_i=0, bytes_=True):
        j = n[0]
        n[0] += 1
        workloads.craft_payload_tick(reps, stores, bufs, slots[j], srcs[j % NB], bytes_=bytes_)
    for _ in range(warmup):
        one_tick()
    us = time_us(torch, one_tick, ticks, sleep_cycles=24_000_000)
    sl = -(-L // 3)
    moved = G * (L + 5 * sl + 4 * sl)                         # put: L read, 5 shards written; every follower's shard written by the same launch (round 6)
    c = [st.counters() for st in stores]
    ok = all(x["unsatisfied"] == 0 for x in c) and reps[0].total_commits() >= G * (n[0] - 2)
    last = n[0] % W                                           # the slot of the last tick is n[0]: its ring cell
    for q in range(1, R):
        d = stores[q].dump()
        ok = ok and bool((d["avail"][last] == (1 << q)).all())
    v = torch.zeros(G, dtype=torch.uint8, device=dev)
    _lib.check(_lib.load().smr_rs_verify(stores[0].plane_ptr(0) + last * stores[0].row_stride, sl, stores[0].shard_stride, stores[0].group_stride, G, 3, 2,
                                         v.data_ptr(), _lib.stream_ptr(None)))
    ok = ok and bool(v.all().item())
    us_engine = time_us(torch, lambda i: one_tick(bytes_=False), 12)   # the engines' tick alone (after the checks)
    us_bytes = max(us - us_engine, 1e-3)
    return {"workload": "CRaft, %d groups x 5 replicas, one %d-byte batch per group per tick, balanced assignment; bytes through smr_craft_pstore_* "
                        "(put + the leader's follow + one follow_many for the four followers, window %d); the engines' tick -- the leader's append, its "
                        "four AppendEntries and their handlers, the replies -- is one launch (smr_raft_cluster_tick)" % (G, L, W),
            "value": G / (us * 1e-6), "unit": "slots/s", "ms_per_tick": us * 1e-3, "engine_only_ms_per_tick": us_engine * 1e-3,
            "bytes_path_ms_per_tick": us_bytes * 1e-3, "rs_payload_GiBps": G * L / 2**30 / (us * 1e-6), "launches_per_tick": {"engine": 1 if os.environ.get("SMR_RAFT_CLUSTER_TICK", "1") != "0" else 3, "bytes": 4},
            "shards_delivered_by_the_put_launch": sum(st.delivered() for st in stores),
            "roofline": {"bound": "hbm", "kernel": "ps_put_deliver_kernel<3, true> + ps_plan_kernel + ps_bytes_plan_many_kernel + ps_bytes_many_kernel",
                         "achieved": moved / (us_bytes * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": moved / (us_bytes * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "alg_bytes_per_launch": moved, "avg_launch_us": us_bytes, "traffic": _payload_leg_traffic("craft_payload")[0] if G == 16384 else None,
                         "traffic_source": _payload_leg_traffic("craft_payload")[1],
                         "survey_8d_bytes_per_launch": G * (5 * sl + 85), "frac_on_survey_8d_bytes": G * (5 * sl + 85) / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "note": "per tick (tick with the stores minus the engines' tick alone); bytes the path has to move: L read + 5 shard_len written "
                                 "by put for the leader, one shard per follower written by the same launch (round 6)"},
            "counters": {k: sum(x[k] for x in c) for k in c[0]}, "verified": ok}


def leg_isolated(name, timeout=180, extra=()):
    """a secondary leg in a child process: a device fault there cannot take the headline line with it"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", name] + list(extra), capture_output=True, timeout=timeout)
    if out.returncode != 0:
        raise RuntimeError("child exited %d: %s" % (out.returncode, out.stderr.decode(errors="replace")[-300:]))
    return json.loads(out.stdout.decode().strip().splitlines()[-1])


def _time_us(torch, fn, iters, sleep_cycles=6_000_000):
    """average device time per call of fn between two HIP events.  A ~3 ms device-side sleep goes first so that the host
    has every launch queued before the first one starts: a Python + ctypes call costs 10-20 us of host time, more than
    some of these kernels run, and an empty queue would make the event pair measure the host instead."""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if sleep_cycles:
        torch.cuda._sleep(int(sleep_cycles))
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def raft_leg(torch, dev, S=32, ticks=48, conflict_p=0.005):
    """BASELINE config 3: Raft, 65 536 groups x 5 replicas, leader-side AppendEntriesReply match-index quorum
    (raft/messages.rs:222-388): per tick S appends, then one reply per follower with end_slot = leader_last -
    lag (lag 0..3 seeded), 5 % dropped, 0.5 % stale-term, 0.5 % conflict replies.  Headline of the leg: batches of 16 ticks
    through smr_raft_leader_run_ticks (ONE launch per batch, the group's state in registers from tick to tick, inputs
    resident); beside it one append call + one reply call per tick (rounds 1-2), whose reply kernel alone gives the
    `roofline` object of the kernel SURVEY 8(d) prices."""
    from summerset_amd import RaftLeaderGroup
    G, R, W = 65536, 5, 512
    rng = np.random.default_rng(0x5EED5EED)
    n_new = torch.full((G,), S, dtype=torch.int32, device=dev)
    pool = []
    for t in range(ticks):
        last = 1 + S * (t + 1) - 1
        lag = rng.integers(0, 4, (R, G))
        u = rng.random((R, G))
        flags = (u >= 0.05).astype(np.uint8)
        term = np.full((R, G), 2, np.uint64)
        term[(u >= 0.05) & (u < 0.055)] = 1                                   # stale term: ignored
        conflict = (u >= 0.055) & (u < 0.055 + conflict_p)
        flags[conflict] |= 2
        end_slot = np.maximum(last - lag, 0).astype(np.uint32)
        pool.append(tuple(torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else
                                           (x.view(np.int32) if x.dtype == np.uint32 else x)).to(dev)
                          for x in (term, end_slot, flags, np.full((R, G), 2, np.uint64),
                                    np.maximum(end_slot.astype(np.int64) - 1, 1).astype(np.uint32))))
    alg = G * (280 + 8 * S)                                                   # SURVEY §8d, per (group, tick)
    # (a) one call per handler per tick
    eng = RaftLeaderGroup(G, R, leader_id=0, window=W, term=2)
    pairs = []

    def tick(i):
        eng.handle_req_batch(n_new)
        rt, es, fl, ct, cs = pool[i]
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        eng.handle_msg_append_entries_reply(rt, es, fl, ct, cs)
        eb.record()
        pairs.append((ea, eb))

    us = _time_us(torch, tick, ticks)            # (queue prefilled: device time, not host call time)
    commits = eng.total_commits()
    # the replies kernel alone: its own event pair in every tick of the run above (real replies, not a re-run)
    us_k = sum(a.elapsed_time(b) for a, b in pairs) / len(pairs) * 1e3
    per_call = {"value": commits / (us * 1e-6 * ticks), "unit": "slots/s", "us_per_tick": us, "launches_per_tick": 2}
    del eng
    # (b) batches of 16 ticks, one launch each
    eng = RaftLeaderGroup(G, R, leader_id=0, window=W, term=2)
    B = 16
    batches = [[dict(n_new=n_new, reply_term=pool[t][0], end_slot=pool[t][1], flags=pool[t][2], conflict_term=pool[t][3], conflict_slot=pool[t][4])
                for t in range(b0, min(b0 + B, ticks))] for b0 in range(0, ticks, B)]
    us_b = _time_us(torch, lambda i: eng.run_ticks(batches[i]), len(batches)) * len(batches) / ticks     # per tick
    commits_b = eng.total_commits()
    t_k = _leg_traffic("smr::raft_ticks_kernel<5>")
    return {"workload": "Raft leader, %d groups x 5 replicas, S=%d appends + 4 AppendEntriesReply per group per tick "
                        "(lag 0-3, 5%% dropped, 0.5%% stale term, 0.5%% conflict)" % (G, S),
            "value": commits_b / (us_b * 1e-6 * ticks), "unit": "slots/s", "us_per_tick": us_b, "ticks_per_launch": B,
            "entry_point": "smr_raft_leader_run_ticks", "same_commits_as_per_call": commits_b == commits,
            "whole_tick_roofline": {"bound": "hbm", "kernel": "raft_ticks_kernel<5> (appends + replies of 16 ticks per launch)", "achieved": alg / (us_b * 1e-6) / 1e9,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (us_b * 1e-6) / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_tick": alg,
                                    "us_per_tick": us_b, "traffic_per_tick": (t_k / B) if t_k else None, "traffic_source": PMC_NOTE},
            "one_call_per_handler": per_call,
            "roofline": {"bound": "hbm", "kernel": "raft_replies_kernel", "achieved": alg / (us_k * 1e-6) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (us_k * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "alg_bytes_per_launch": alg, "avg_launch_us": us_k, "traffic": _leg_traffic("smr::raft_replies_kernel<false, 5>"),
                         "traffic_source": PMC_NOTE}}


def epaxos_leg(torch, dev, ticks=16):
    """BASELINE config 5 (one replica's share): EPaxos, 65 536 groups x 5 replicas, optimized quorums (3/3): per tick
    every group's replica 0 proposes one instance on a Zipf(0.99) key out of 64 and receives the 4
    PreAcceptReplies, 10 % of which carry an extra dependency (dependency.rs:175-240, messages.rs:96-270)."""
    from summerset_amd import EPaxosReplicaGroup
    G, R, W, K = 65536, 5, 32, 64
    eng = EPaxosReplicaGroup(G, R, me=0, window=W, n_keys=K)
    rng = np.random.default_rng(0x5EED5EED)
    zipf = 1.0 / np.arange(1, K + 1) ** 0.99
    zipf /= zipf.sum()
    i32 = lambda x: torch.from_numpy(np.ascontiguousarray(x.view(np.int64) if x.dtype == np.uint64 else
                                                          (x.view(np.int32) if x.dtype == np.uint32 else x))).to(dev)
    keys = [i32(rng.choice(K, G, p=zipf).astype(np.uint8)) for _ in range(ticks)]
    extra = [rng.random((R, G)) < 0.1 for _ in range(ticks)]
    flags = np.ones((R, G), np.uint8)
    flags[0] = 0
    flags_d = i32(flags)
    ballot_d = i32(np.full((R, G), 1, np.uint64))
    # outputs allocated once and the C-ABI called directly, so that the event pairs bracket the kernels alone
    import ctypes as C
    from summerset_amd._lib import EpMsg, check
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    m = dict(flags=z(G, torch.uint8), col=z(G, torch.int32), ballot=z(G, torch.int64), seq=z(G, torch.int64),
             deps=z((R, G), torch.int32))
    r = dict(decision=z(G, torch.uint8), seq=z(G, torch.int64), deps=z((R, G), torch.int32))
    msg = EpMsg(m["flags"].data_ptr(), None, m["col"].data_ptr(), m["ballot"].data_ptr(), m["seq"].data_ptr(),
                m["deps"].data_ptr(), None)
    st_ = torch.cuda.current_stream().cuda_stream
    extra_d = [torch.from_numpy(x).to(dev) for x in extra]
    commits_d = torch.zeros((), dtype=torch.int64, device=dev)
    ev = []
    torch.cuda.synchronize()
    torch.cuda._sleep(6_000_000)                  # the host queues the ticks while the device waits: event pairs = device time
    for t in range(ticks):
        e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        e0.record()
        check(eng._L.smr_ep_propose(eng._h, keys[t].data_ptr(), None, C.byref(msg), st_))
        e1.record()
        # the peers' answers: my (seq, deps), 10 % with seq + 1 and one more dependency (stand-in for the
        # four acceptors, built on the device from the PreAccept just produced; not timed)
        ex = extra_d[t]
        seq = (m["seq"].unsqueeze(0).repeat(R, 1) + ex.to(torch.int64)).contiguous()
        deps = m["deps"].unsqueeze(0).repeat(R, 1, 1)
        deps[:, 1, :] = torch.where(ex, torch.clamp(deps[:, 1, :], min=0) + 1, deps[:, 1, :])
        deps = deps.contiguous()
        e2.record()
        check(eng._L.smr_ep_handle_pre_accept_replies(eng._h, m["col"].data_ptr(), ballot_d.data_ptr(), seq.data_ptr(),
                                                      deps.data_ptr(), flags_d.data_ptr(), None, None,
                                                      r["decision"].data_ptr(), r["seq"].data_ptr(),
                                                      r["deps"].data_ptr(), st_))
        e3.record()
        commits_d += (r["decision"] == 3).sum()
        ev.append((e0, e1, e2, e3, seq, deps))
    torch.cuda.synchronize()
    t_prop = sum(e[0].elapsed_time(e[1]) for e in ev)
    t_rep = sum(e[2].elapsed_time(e[3]) for e in ev)
    committed = int(commits_d.item())
    us_rep = t_rep / ticks * 1e3
    alg = G * 370                                                             # <= 370 B per instance
    return {"workload": "EPaxos command leader, %d groups x 5 replicas, 1 proposal per group per tick on Zipf(0.99) keys "
                        "of 64, 4 PreAcceptReplies each, 10%% with an extra dependency" % G,
            "value": committed / (t_rep * 1e-3), "unit": "fast-path commits/s of the reply kernel",
            "fast_path_fraction": committed / (G * ticks), "propose_kernel_us": t_prop / ticks * 1e3,
            "roofline": {"bound": "hbm", "kernel": "ep_pre_accept_replies_kernel", "achieved": alg / (us_rep * 1e-6) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (us_rep * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "alg_bytes_per_launch": alg, "avg_launch_us": us_rep, "traffic": _leg_traffic("smr::ep_pre_accept_replies_kernel<5>"),
                         "traffic_source": PMC_NOTE}}



def epaxos_exec_leg(torch, dev, ticks=16):
    """the EPaxos leg with dependency-graph execution on (smr_ep_cfg.execute): every handler call is followed by
    ep_execute_kernel (execution.rs:25-211).  Opt-in (--late-legs): the kernel has not had a device run yet."""
    from summerset_amd import EPaxosReplicaGroup
    G, R, W, K = 65536, 5, 32, 64
    eng = EPaxosReplicaGroup(G, R, me=0, window=W, n_keys=K, execute=True)
    rng = np.random.default_rng(0x5EED5EED)
    zipf = 1.0 / np.arange(1, K + 1) ** 0.99
    zipf /= zipf.sum()
    keys = [torch.from_numpy(rng.choice(K, G, p=zipf).astype(np.uint8)).to(dev) for _ in range(ticks)]
    flags = np.ones((R, G), np.uint8)
    flags[0] = 0
    flags_d = torch.from_numpy(flags).to(dev)
    ballot_d = torch.ones((R, G), dtype=torch.int64, device=dev)
    t_prop = t_rep = 0.0
    for t in range(ticks):
        e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        e0.record()
        m = eng.handle_req_batch(keys[t])
        e1.record()
        seq = m["seq"].unsqueeze(0).repeat(R, 1).contiguous()
        deps = m["deps"].unsqueeze(0).repeat(R, 1, 1).contiguous()
        e2.record()
        eng.handle_msg_pre_accept_reply(m["col"], ballot_d, seq, deps, flags_d)
        e3.record()
        torch.cuda.synchronize()
        t_prop += e0.elapsed_time(e1)
        t_rep += e2.elapsed_time(e3)
    x = eng.exec_dump()
    c = [int(v) for v in x["counters"]]
    return {"workload": "EPaxos command leader with execution, %d groups, 1 proposal + 4 agreeing PreAcceptReplies per group per tick" % G,
            "value": c[0] / ((t_prop + t_rep) * 1e-3), "unit": "commands executed/s (handler + execution kernels, incl. host call overhead)",
            "propose_call_us": t_prop / ticks * 1e3, "replies_call_us": t_rep / ticks * 1e3,
            "executed": c[0], "re_executed": c[1], "attempts": c[4], "abandoned": c[5]}


def rspaxos_replica_leg(torch, dev, ticks=32):
    """the RSPaxos replica engine (csrc/rsp_engine.hip), leader side of BASELINE config 4: 16 384 groups, per tick one
    batch per group (handle_req_batch) and the 4 followers' AcceptReplies (10 % lost), threshold majority + 1.
    Opt-in (--late-legs): the kernels have not had a device run yet."""
    from summerset_amd import RSPaxosReplicaGroup
    G, R, W = 16384, 5, 64
    eng = RSPaxosReplicaGroup(G, R, me=0, window=W, fault_tolerance=1)
    eng.preset_leader(0)
    rng = np.random.default_rng(0x5EED5EED)
    b0 = (1 << 8) | 1
    ballot = torch.full((R, G), b0, dtype=torch.int64, device=dev)
    t_req = t_rep = 0.0
    for t in range(ticks):
        val = torch.arange(1 + t * G, 1 + (t + 1) * G, dtype=torch.int32, device=dev)
        fl = (rng.random((R, G)) >= 0.1).astype(np.uint8)
        fl[0] = 0
        fl = torch.from_numpy(fl).to(dev)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        a = eng.req_batch(val)
        e1.record()
        eng.accept_replies(a["a_slot"][0].contiguous(), ballot, fl)
        e2.record()
        torch.cuda.synchronize()
        t_req += e0.elapsed_time(e1)
        t_rep += e1.elapsed_time(e2)
    c = [int(v) for v in eng.dump()["counters"]]
    return {"workload": "RSPaxos replica engine, leader of %d groups x 5 replicas, f = 1: one batch + 4 AcceptReplies (10%% lost) per tick" % G,
            "value": c[0] / ((t_req + t_rep) * 1e-3), "unit": "committed slots/s (incl. host call overhead)",
            "req_batch_call_us": t_req / ticks * 1e3, "accept_replies_call_us": t_rep / ticks * 1e3, "commits": c[0], "executed": c[1]}


def craft_leader_leg(torch, dev, ticks=32):
    """the CRaft leader variant (raft_replies_kernel<true>, craft_heartbeat_kernel): 65 536 groups, per tick up to 2 new
    entries per group, the 4 followers' AppendEntriesReplies (5 % lost, one follower of every third group silent), a
    heartbeat tick every 4th tick.  Opt-in (--late-legs): the kernels have not had a device run yet."""
    from summerset_amd import CRaftLeaderGroup
    G, R, W = 65536, 5, 64
    eng = CRaftLeaderGroup(G, R, 0, W, term=1, fault_tolerance=1, repeat_threshold=2)
    rng = np.random.default_rng(0xC4AF7)
    term = torch.ones((R, G), dtype=torch.int64, device=dev)
    log_len = np.ones(G, np.int64)
    t_app = t_rep = t_hb = 0.0
    n_hb = 0
    for t in range(ticks):
        n_new = rng.integers(0, 3, G).astype(np.int32)
        log_len += n_new
        es = np.maximum(log_len[None, :] - 1 - rng.integers(0, 3, (R, G)), 0).astype(np.int32)
        fl = (rng.random((R, G)) >= 0.05).astype(np.uint8)
        fl[0] = 0
        fl[1, ::3] = 0
        d_new, d_es, d_fl = torch.from_numpy(n_new).to(dev), torch.from_numpy(es).to(dev), torch.from_numpy(fl).to(dev)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        eng.handle_req_batch(d_new)
        e[1].record()
        eng.handle_msg_append_entries_reply(term, d_es, d_fl)
        e[2].record()
        if t % 4 == 3:
            eng.bcast_heartbeats(dev)
            n_hb += 1
        e[3].record()
        torch.cuda.synchronize()
        t_app += e[0].elapsed_time(e[1]); t_rep += e[1].elapsed_time(e[2]); t_hb += e[2].elapsed_time(e[3]) if t % 4 == 3 else 0.0
    c = eng.dump_craft()
    commits = eng.total_commits()
    return {"workload": "CRaft leader of %d groups x 5 replicas, f = 1: appends + 4 AppendEntriesReplies per tick, heartbeat tick every 4th" % G,
            "value": commits / ((t_app + t_rep + t_hb) * 1e-3), "unit": "committed entries/s (incl. host call overhead)",
            "append_call_us": t_app / ticks * 1e3, "replies_call_us": t_rep / ticks * 1e3, "heartbeat_call_us": t_hb / max(n_hb, 1) * 1e3,
            "commits": commits, "groups_in_full_copy_mode": int(c["full_copy_mode"].sum())}


def quorum_read_leg(torch, dev, ticks=32):
    """MultiPaxos near quorum reads (csrc/qread.hip), the issuer's side: 65 536 groups, per tick one ReadQuery of 4 Gets per
    group issued and the 4 peers' ReadQueryReplies tallied (10 % lost).  Opt-in (--late-legs): no device run yet."""
    from summerset_amd import QuorumReadGroup
    G, R, K, B = 65536, 5, 64, 4
    eng = QuorumReadGroup(G, R, 0, K, B, 1)
    rng = np.random.default_rng(0x9EAD)
    n = torch.full((G,), B, dtype=torch.uint8, device=dev)
    t_iss = t_rep = 0.0
    answered = 0
    for t in range(ticks):
        st = rng.integers(0, 3, (R, B, G)).astype(np.uint8)
        sl = rng.integers(1, 50, (R, B, G)).astype(np.int32)
        rep = dict(state=torch.from_numpy(st).to(dev), slot=torch.from_numpy(sl).to(dev), val=torch.from_numpy(sl * 7).to(dev))
        own = {k: v[0].contiguous() for k, v in rep.items()}
        fl = (rng.random((R, G)) >= 0.1).astype(np.uint8)
        fl[0] = 0
        d_fl = torch.from_numpy(fl).to(dev)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        eng.issue(0, n, own)
        e1.record()
        outcome, val, done = eng.handle_msg_read_query_reply(0, rep, d_fl)
        e2.record()
        torch.cuda.synchronize()
        t_iss += e0.elapsed_time(e1); t_rep += e1.elapsed_time(e2)
        answered += int(done.sum())
    c = [int(x) for x in eng.dump()["counters"]]
    return {"workload": "quorum reads, issuer of %d groups x 5 replicas: one ReadQuery of %d Gets + 4 ReadQueryReplies (10%% lost) per tick" % (G, B),
            "value": answered * B / ((t_iss + t_rep) * 1e-3), "unit": "reads answered/s (incl. host call overhead)",
            "issue_call_us": t_iss / ticks * 1e3, "replies_call_us": t_rep / ticks * 1e3, "values": c[0], "retries": c[1], "not_found": c[2],
            "conflicts": c[3]}


def wire_ingest_leg(torch, dev, G=65536, S=32, iters=12):
    """The leader's receive side of one tick on the device (csrc/wire_ingest.hip): 4 connections per group, each
    with the tick's S AcceptReply frames (`[u64 BE 9][00 03 FB slot16 FB ballot16 00]`, 17 bytes: slots and the ballot in the
    two-byte varint range) and a Heartbeat in front of every fourth connection's -- parsed into smr_mp_ack / smr_wire_hb records.
    4 distinct byte buffers in rotation (4 x 145 MB + records: beyond the 256 MiB L3).  Algorithmic bytes: the stream once +
    the records once."""
    from summerset_amd import wire
    from summerset_amd.multipaxos import ACK_DTYPE
    n_conn, POOL = G * 4, 4
    hb = np.frombuffer(wire.heartbeat(0x101, 300, 290, 0), np.uint8)
    one = np.frombuffer(wire.accept_reply(300, 0x101), np.uint8)
    assert len(one) == 17
    has_hb = (np.arange(n_conn) % 4 == 0)
    lens = S * 17 + has_hb * len(hb)
    off = np.zeros(n_conn + 1, np.int64)
    off[1:] = np.cumsum(lens)
    bufs = []
    for k in range(POOL):
        slots = (300 + 32 * k + np.arange(S)).astype("<u2")
        body = np.tile(one, (S, 1))
        body[:, 11:13] = slots.view(np.uint8).reshape(S, 2)          # the slot's two little-endian bytes behind 0xFB
        body = body.reshape(-1)
        four = np.concatenate([hb, body, body, body, body])          # connections 4i .. 4i + 3: the first opens with the Heartbeat
        bufs.append(torch.from_numpy(np.tile(four, n_conn // 4)).to(dev))
        assert bufs[-1].numel() == int(off[-1])
    # the sequential decoder agrees on one connection with and one without the heartbeat
    for c in (0, 1):
        pos, blob0, n_ok = int(off[c]), bufs[1].cpu().numpy().tobytes(), 0
        while pos < off[c + 1]:
            n, m = wire.decode(blob0[pos:int(off[c + 1])])
            assert n > 0 and m["kind"] in (wire.ACCEPT_REPLY, wire.HEARTBEAT)
            pos += n; n_ok += 1
        assert n_ok == S + (1 if has_hb[c] else 0)
    ing = wire.MpIngest(n_conn, n_conn * S, n_conn, 16, device=dev)
    d_off = torch.from_numpy(off).to(dev)
    d_grp = torch.from_numpy((np.arange(n_conn) // 4).astype(np.int32)).to(dev)
    d_peer = torch.from_numpy((1 + np.arange(n_conn) % 4).astype(np.uint8)).to(dev)
    for k in range(POOL):
        ing.ingest(bufs[k], d_off, d_grp, d_peer)
    r = ing.results()
    assert r["n_acks"] == n_conn * S and r["n_hbs"] == n_conn // 4 and r["n_others"] == 0 and r["n_malformed"] == 0
    assert (r["consumed"] == lens).all() and (r["acks"]["slot"][:S] == 300 + 32 * 3 + np.arange(S)).all()
    us = _time_us(torch, lambda i: ing.ingest(bufs[i % POOL], d_off, d_grp, d_peer), iters)
    stream_bytes = int(off[-1])
    alg = stream_bytes + r["n_acks"] * ACK_DTYPE.itemsize + r["n_hbs"] * wire.HB_DTYPE.itemsize
    dense = {"what": "smr_wire_ingest_mp: dense lists in the sequential decoder's order across connections (a counting parse in front of the writing one)",
             "value": r["n_acks"] / (us * 1e-6), "call_us": us,
             "roofline": {"bound": "hbm", "kernel": "wire_ingest_mp_kernel<false> + <true> (one smr_wire_ingest_mp call)", "achieved": alg / (us * 1e-6) / 1e9,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_launch": alg,
                          "avg_launch_us": us, "traffic": _sum_traffic("smr::wire_ingest_mp_kernel<false, false>", "smr::wire_ingest_mp_kernel<true, false>") or
                                                         _sum_traffic("smr::wire_ingest_mp_kernel<false>", "smr::wire_ingest_mp_kernel<true>")}}   # (the names before the template's second parameter)
    del ing
    # round 5: the same parse in ONE pass, a segment per connection (smr_wire_ingest_mp_conn) -- what smr_mp_deliver_acks_conn takes
    ingc = wire.MpIngestConn(n_conn, stream_bytes, 1, 1, device=dev)
    for k in range(POOL):
        ingc.ingest(bufs[k], d_off, d_grp, d_peer)
    cnt = ingc.cnt.cpu().numpy()
    assert int(cnt[:, 0].sum()) == n_conn * S and int(cnt[:, 1].sum()) == n_conn // 4 and int(cnt[:, 2].sum()) == 0
    assert (ingc.status.cpu().numpy() == 0).all() and (ingc.consumed.cpu().numpy() == lens).all()
    seg0 = ingc.acks.cpu().numpy().view(wire.ACK12_DTYPE)[int(off[5]) // 13:int(off[5]) // 13 + S]  # connection 5's segment: (slot, ballot) records
    assert (seg0["slot"] == 300 + 32 * 3 + np.arange(S)).all() and (seg0["ballot_lo"] == 0x101).all() and (seg0["ballot_hi"] == 0).all()
    us1 = _time_us(torch, lambda i: ingc.ingest(bufs[i % POOL], d_off, d_grp, d_peer), iters)
    return {"workload": "leader-side receive path of one tick: %d connections (%d groups x 4 peers), %d AcceptReply frames each + a Heartbeat on every "
                        "fourth, %d MB of frames -> %d AcceptReply records; one pass, a segment per connection, 12-byte (slot, ballot) records (the group "
                        "and the peer are the connection's): the roofline's bytes stay the stream + 24 B per record, what the dense list carries" % (n_conn, G, S, stream_bytes // 1000000, r["n_acks"]),
            "value": r["n_acks"] / (us1 * 1e-6), "unit": "AcceptReply frames/s", "call_us": us1, "stream_GBps": stream_bytes / (us1 * 1e-6) / 1e9,
            "roofline": {"bound": "hbm", "kernel": "wire_ingest_mp_kernel<true, true> (one smr_wire_ingest_mp_conn call)", "achieved": alg / (us1 * 1e-6) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (us1 * 1e-6) / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_launch": alg,
                         "avg_launch_us": us1, "traffic": _leg_traffic("smr::wire_ingest_mp_kernel<true, true>"),
                         "traffic_source": PMC_NOTE},
            "dense_lists": dense}


def _sum_traffic(*kernels):
    t = [_leg_traffic(k) for k in kernels]
    return sum(t) if all(t) else None


def reply_ingest_leg(torch, dev, G=65536, R=5, iters=12, junk_every=16):
    """The Raft leader's receive side of one tick on the device (csrc/wire_ingest_replies.hip): one connection per (group, follower),
    each with one AppendEntriesReply frame (a conflict on every eighth, a RequestVoteReply in front of every sixteenth) -> the
    [R][G] arrays `smr_raft_leader_handle_replies` takes.  A launch-bound call: ~17 bytes per connection."""
    from summerset_amd import wire
    n_conn = G * (R - 1)
    ok = np.frombuffer(wire.raft_append_entries_reply(3, 70000), np.uint8)
    conf = np.frombuffer(wire.raft_append_entries_reply(3, 70000, (2, 69000)), np.uint8)
    vote = np.frombuffer(wire.raft_request_vote_reply(3, False), np.uint8)
    parts = [np.concatenate(([vote] if junk_every and c % junk_every == 0 else []) + [conf if c % 8 == 0 else ok]) for c in range(16)]
    lens = np.tile(np.array([len(x) for x in parts], np.int64), n_conn // 16)
    off = np.zeros(n_conn + 1, np.int64)
    off[1:] = np.cumsum(lens)
    buf = torch.from_numpy(np.tile(np.concatenate(parts), n_conn // 16)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_grp = torch.from_numpy((np.arange(n_conn) // (R - 1)).astype(np.int32)).to(dev)
    d_peer = torch.from_numpy((1 + np.arange(n_conn) % (R - 1)).astype(np.uint8)).to(dev)
    ing = wire.ReplyIngest(n_conn, G, R, n_conn, dev)
    o = ing.raft(buf, d_off, d_grp, d_peer)
    r = ing.results()
    assert r["n_replies"] == n_conn and r["n_others"] == (n_conn // junk_every if junk_every else 0) and r["n_malformed"] == 0 and (r["consumed"] == lens).all()
    assert int(o["flags"][1:].sum().item()) == n_conn + 2 * (n_conn // 8) and int(o["end_slot"][1, 5].item()) == 70000
    us = _time_us(torch, lambda i: ing.raft(buf, d_off, d_grp, d_peer), iters)
    # the follower side + the leader side without leaving the device: reply arrays -> frames (smr_wire_emit_raft_replies, one slot per
    # reply) -> the leader's arrays (the slots as connections: conn_off = slot starts, conn_len = the emitted lengths)
    fl = torch.where(torch.arange(n_conn, device=dev) % 8 == 0, 3, 1).to(torch.uint8)
    term = torch.full((n_conn,), 3, dtype=torch.int64, device=dev)
    es = torch.full((n_conn,), 70000, dtype=torch.int32, device=dev)
    ct, cs = torch.full((n_conn,), 2, dtype=torch.int64, device=dev), torch.full((n_conn,), 69000, dtype=torch.int32, device=dev)
    slot_off = torch.arange(n_conn, dtype=torch.int64, device=dev) * wire.EMIT_RAFT_STRIDE

    def loop(i):
        frames, ln = wire.emit_raft_replies(fl, term, es, ct, cs)
        ing.raft(frames.view(-1), slot_off, d_grp, d_peer, conn_len=ln)
    # round 6: the parse as the prologue of the leader's reply handler (smr_raft_leader_handle_wire_replies: ONE launch, no [R][G]
    # arrays, no memsets) beside the two calls it stands for, on two leaders in the same state: frames -> last_commit
    from summerset_amd import RaftLeaderGroup
    lead_a, lead_b = RaftLeaderGroup(G, R, leader_id=0, window=64, term=3), RaftLeaderGroup(G, R, leader_id=0, window=64, term=3)
    ing_b = wire.ReplyIngest(n_conn, G, R, n_conn, dev)
    n32 = torch.full((G,), 32, dtype=torch.int32, device=dev)
    ok2 = np.frombuffer(wire.raft_append_entries_reply(3, 30), np.uint8)
    conf2 = np.frombuffer(wire.raft_append_entries_reply(3, 30, (2, 9)), np.uint8)
    parts2 = [np.concatenate(([vote] if junk_every and c % junk_every == 0 else []) + [conf2 if c % 8 == 0 else ok2]) for c in range(16)]
    off2 = np.zeros(n_conn + 1, np.int64)
    off2[1:] = np.cumsum(np.tile(np.array([len(x) for x in parts2], np.int64), n_conn // 16))
    buf2, d_off2 = torch.from_numpy(np.tile(np.concatenate(parts2), n_conn // 16)).to(dev), torch.from_numpy(off2).to(dev)
    for ld in (lead_a, lead_b):
        ld.handle_req_batch(n32)

    def two_calls(i):
        o2 = ing.raft(buf2, d_off2, d_grp, d_peer)
        lead_a.handle_msg_append_entries_reply(o2["reply_term"], o2["end_slot"], o2["flags"], o2["conflict_term"], o2["conflict_slot"])
    two_calls(0)
    ing_b.raft_into(lead_b, buf2, d_off2)
    ra, rb = ing.results(), ing_b.results()
    da, db = lead_a.dump(), lead_b.dump()
    assert all(ra[k] == rb[k] for k in ("n_replies", "n_others", "n_malformed", "n_deferred")) and all(np.array_equal(da[k], db[k]) for k in da)
    assert int(db["last_commit"].min()) == 30
    us_two = _time_us(torch, two_calls, iters)
    us_fused = _time_us(torch, lambda i: ing_b.raft_into(lead_b, buf2, d_off2), iters)
    loop(0)
    r2 = ing.results()
    assert r2["n_replies"] == n_conn and r2["n_malformed"] == 0 and int(o["flags"][1:].sum().item()) == n_conn + 2 * (n_conn // 8)
    us_loop = _time_us(torch, loop, iters)
    alg = int(off[-1]) + n_conn * (8 + 4 + 1) + (n_conn // 8) * 12
    return {"workload": "Raft leader-side receive path of one tick: %d connections (%d groups x %d followers), one AppendEntriesReply each -> the "
                        "leader's last_commit in one launch" % (n_conn, G, R - 1), "value": n_conn / (us_fused * 1e-6), "unit": "AppendEntriesReply frames/s",
            "call_us": us_fused, "emit_then_ingest_us": us_loop,
            "frames_to_last_commit": {"one_launch_us": us_fused, "two_calls_us": us_two, "entry_point": "smr_raft_leader_handle_wire_replies",
                                      "same_state_and_counts_as_the_two_calls": True},
            # the leg's figure since round 6: the ONE-launch call -- the frames' bytes + the leader's per-group state (280 B per
            # group and tick of replies) over its time; `ingest_alone` keeps the parse-into-arrays call of rounds 3-5
            "roofline": {"bound": "hbm", "kernel": "raft_wire_replies_kernel<false, 5> (parse + reply handler: one smr_raft_leader_handle_wire_replies call)",
                         "achieved": (int(off2[-1]) + 280 * G) / (us_fused * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (int(off2[-1]) + 280 * G) / (us_fused * 1e-6) / 1e9 / HBM_PEAK_GBS,
                         "alg_bytes_per_launch": int(off2[-1]) + 280 * G, "avg_launch_us": us_fused,
                         "traffic": _leg_traffic("smr::raft_wire_replies_kernel<false, 5>"), "traffic_source": PMC_NOTE},
            "ingest_alone": {"kernel": "wire_ingest_replies_kernel<0> (+ two memsets: one smr_wire_ingest_raft_replies call)", "call_us": us,
                             "alg_bytes_per_launch": alg, "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                             "traffic": _leg_traffic("smr::wire_ingest_replies_kernel<0>")}}


def _cpu_run(a):
    """one process, one thread: the CPU oracle on G groups of the bench workload for about `seconds`"""
    slots, window, drop, timeouts, hb_every, G, seconds = a
    from oracle import oracle as O
    from summerset_amd import stream
    R, S, W = 5, slots, window
    cap = W + 4
    m = O.MpOracle(G, R, W, win_reserve=W // 8, cap=cap, record_commits=False)
    m.preset_leader(0)
    st = stream.MultiPaxosStream(G, R, S, cap=cap, n_ticks=64, drop_p=drop, timeout_frac=timeouts,
                                 hb_every=hb_every, rand_rows=S + 4, max_drop=2)
    pool = [st.tick(t) for t in range(8)]
    spent, ticks, t = 0.0, 0, 0
    while spent < seconds:
        inp = dict(pool[t % 8])
        inp.update(st.tick_events(t))
        t0 = time.perf_counter()
        m.tick(**inp)
        spent += time.perf_counter() - t0
        ticks += 1
        t += 1
    commits = sum(m.total_commits(r) for r in range(R))
    return commits / spent, ticks, spent


def cpu_leg(args, seconds):
    """The CPU oracle (literal restatement of the reference handlers) on a bounded sample of the
    same workload (same S / H / loss / timeout rates, 1024 groups per process): one core, then one
    single-threaded process per host core side by side (groups are independent, so this is how a
    multi-core host would run them), rates summed."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 32)                         # ~160 MB of oracle state per process; `cores` = processes really used
    cfg = (args.slots, args.window, args.drop, args.timeouts, args.hb_every, 1024)
    v1, n1, s1 = _cpu_run(cfg + (seconds * 0.4,))
    vn, sn = v1, s1
    if cores > 1:
        with mp.get_context("fork").Pool(cores) as pool:
            res = pool.map(_cpu_run, [cfg + (seconds * 0.6,)] * cores)
        vn, sn = sum(r[0] for r in res), max(r[2] for r in res)
    return {"value": vn, "unit": "slots/s", "cores": cores, "kind": "port", "single_core_value": v1,
            "sample": "oracle/mp_oracle.c, 1024 groups x 5 replicas x S=%d per process: %d single-threaded processes side by "
                      "side for %.1f s (rates summed), and one process for %.1f s (%d ticks); the reference's Rust/tokio "
                      "path cannot be built here (no cargo, no vendored crates)" % (args.slots, cores, sn, s1, n1)}


def spread_run(args, torch, dist, rank, world, dev, steps, warmup, call_by_call=False):
    """the headline workload with the replicas of every group on different ranks (layout L2): `steps` timed ticks behind `warmup`.
    Weak scaling like the co-located line: args.groups groups per GPU.  At world 1 the job's ranks are virtual
    (spread_mp.in_process).  Collective on every rank; returns the line's dict."""
    from summerset_amd import shard, spread_mp, stream
    R, S, W, H = 5, args.slots, args.window, args.hb_every
    cap = W + 4
    virtual = world == 1
    nr = args.spread_ranks if virtual else world
    total = args.groups * (1 if virtual else world)
    kw = dict(win_reserve=W // 8, outbox_cap=cap)
    job = spread_mp.in_process(total, R, W, nr, dev, S, **kw) if virtual else spread_mp.SpreadMultiPaxos(total, R, W, rank, world, dev, S, **kw)
    job.preset_leader(0)
    via = "a device copy between the virtual ranks' buffers" if virtual else "torch.distributed.all_to_all_single"
    comm = None
    if not virtual and dist.get_backend() == "nccl" and os.environ.get("SMR_L2_TORCH") is None:
        # the exchange inside the library (round 4): smr_comm_exchange = grouped ncclSend / ncclRecv pairs on the plans' buffers, the whole
        # tick one C call (smr_mp_spread_tick).  SMR_L2_TORCH=1 keeps the collectives in torch.distributed.
        try:
            from summerset_amd import comm as _comm
            comm = _comm.Comm.from_torch_distributed(dev)
            job.bind_comm(comm)
            via = "smr_comm_exchange (libsummerset_hip.so: RCCL send / recv pairs), the tick one smr_mp_spread_tick call"
        except Exception as e:                                   # noqa: BLE001
            via += " (smr_comm_init_rank failed: %s)" % e
    mine = sorted({b for rk in job.ranks for b in rk.blocks} if virtual else job.blocks)
    n_ticks = warmup + steps
    skw = dict(cap=cap, n_ticks=n_ticks, drop_p=args.drop, timeout_frac=timeout_frac(args), hb_every=H, rand_rows=S + 4, max_drop=2,
               timeout_span=timeout_span(args))                  # the co-located line's rate of leader changes per tick
    sts = {b: stream.MultiPaxosStream(hi - lo, R, S, group_base=lo, **skw) for b, (lo, hi) in ((b, shard.group_range(total, nr, b)) for b in mine)}
    pools = {b: [{k: torch.from_numpy(v).to(dev) for k, v in st.tick(t).items() if k in ("req_cnt", "req_val", "ackctl")} for t in range(args.pool)]
             for b, st in sts.items()}
    evs = {b: [{k: torch.from_numpy(v).to(dev) for k, v in st.tick_events(t).items()} for t in range(n_ticks)] for b, st in sts.items()}
    hb = next(iter(sts.values())).heartbeat

    tick_fn = job.tick_call_by_call if call_by_call else job.tick

    def step(t):
        tick_fn({b: dict(pools[b][t % args.pool], **evs[b][t]) for b in mine}, heartbeat=hb(t))
    commits_of = (lambda: sum(rk.commits() for rk in job.ranks)) if virtual else job.commits
    for t in range(warmup):
        step(t)
    torch.cuda.synchronize()
    c0 = commits_of()
    sent0 = sum(rk.bytes_sent for rk in job.ranks) if virtual else job.bytes_sent
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(warmup, n_ticks):
        step(t)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    commits = commits_of() - c0
    sent = (sum(rk.bytes_sent for rk in job.ranks) if virtual else job.bytes_sent) - sent0
    dropped = sum(rk.dropped_overflow_entries() for rk in job.ranks) if virtual else job.dropped_overflow_entries()
    elapsed, commits = shard.reduce_metric(elapsed, commits, device=dev)
    plans = (job.ranks[0] if virtual else job)._plans
    line = {"metric": "committed_slots_per_sec", "value": commits / elapsed, "unit": "slots/s", "n_gpus": world,
            "ranks": shard.count_ranks(dev), "backend": dist.get_backend() if world > 1 else None, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "MultiPaxos lock-step, %d groups/GPU x 5 replicas, S=%d new slots/group/tick, heartbeat every %d ticks, "
                                   "%.0f%% ack loss (<= 2 lost per slot), %s" % (args.groups, S, H, args.drop * 100, timeouts_text(args)),
                       "groups_per_gpu": args.groups, "replicas": R, "slots_per_tick": S, "window": W, "layout": "spread",
                       "spread_ranks": nr, "ranks_are": "virtual (one process, one GPU: the collective is a device copy)" if virtual else "processes, one per GPU"},
            "exchange": {"via": via, "collectives_per_tick": "3 with a heartbeat round, else 2 (one all_to_all_single each)",
                         "host_calls_per_tick": "one smr_mp_spread_segment call per segment (3, 4 with a heartbeat round) + the collectives"
                                                if not call_by_call else "one per round per block + one per pack / unpack (rounds 1-2)",
                         "bytes_per_exchange_per_rank": {ph: int(sum(p["in_split"])) for ph, p in plans.items()},
                         "bytes_sent_per_tick_per_rank": sent / steps / (nr if virtual else 1), "overflow_entries_dropped": dropped},
            "roofline": None, "cpu_baseline": None,
            "note": "correctness layout of the north star's inter-replica fan-out; the roofline / cpu_baseline objects belong to the co-located line"}
    for rk in (job.ranks if virtual else [job]):
        rk.close()
    if comm is not None:
        comm.close()
    return line


def spread_main(args, torch, dist, rank, local, world, dev):
    """--layout spread: see spread_run"""
    line = spread_run(args, torch, dist, rank, world, dev, args.steps, args.warmup)
    if os.environ.get("SMR_SPREAD_AB"):                            # (experiments: the call-by-call tick of rounds 1-2 beside it)
        old = spread_run(args, torch, dist, rank, world, dev, args.steps, args.warmup, call_by_call=True)
        line["call_by_call"] = {"ms_per_step": old["ms_per_step"], "value": old["value"]}
    if rank == 0:
        emit_line(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


L2_EXCHANGE_FAILED = []                                       # what _bind_library_exchange could not bind (-> the line's legs_failed)


def _bind_library_exchange(job, dist, dev, virtual):
    """at N > 1 over RCCL the spread layouts' exchanges run inside the library (smr_comm_exchange: grouped ncclSend / ncclRecv on the
    plans' buffers); SMR_L2_TORCH=1 keeps torch.distributed.all_to_all_single.  Returns (comm or None, how the bytes travel)."""
    if virtual:
        return None, "a device copy between the virtual ranks' buffers"
    if dist.get_backend() != "nccl" or os.environ.get("SMR_L2_TORCH") is not None:
        return None, "torch.distributed.all_to_all_single"
    try:
        from summerset_amd import comm as _comm
        c = _comm.Comm.from_torch_distributed(dev)
        job.bind_comm(c)
        return c, "smr_comm_exchange (libsummerset_hip.so: RCCL send / recv pairs)"
    except Exception as e:                                       # noqa: BLE001
        # (at N > 1 a library exchange that cannot be had is a FAILED leg of the line, not a `via` string;
        #  the run still completes over torch.distributed so that the job's other ranks are not left in a collective)
        L2_EXCHANGE_FAILED.append("smr_comm_init_rank: %s" % e)
        return None, "torch.distributed.all_to_all_single (smr_comm_init_rank FAILED: %s)" % e


def spread_rspaxos_main(args, torch, dist, rank, local, world, dev):
    """--layout spread-rspaxos: BASELINE config 4 in layout L2 -- RSPaxos, 16 384 groups per GPU x 5 replicas, one 4 KiB Put per
    group per tick, the replicas of a group on different ranks (summerset_amd/spread_rsp.py): per tick one all_to_all_single
    with the Accepts AND the followers' shards of the tick's codewords, one with the AcceptReplies, two more on heartbeat
    ticks.  At world 1 the ranks are virtual (--spread-ranks).  Weak scaling: --groups is ignored, 16 384 groups per GPU."""
    from summerset_amd import shard, spread_rsp
    G, R, W, L, NB, H = 16384, 5, 64, 4113, 3, 4
    virtual = world == 1
    nr = args.spread_ranks if virtual else world
    total = G * nr
    job = spread_rsp.in_process(total, R, W, nr, dev, L) if virtual else spread_rsp.SpreadRSPaxos(total, R, W, rank, world, dev, L)
    comm, via = _bind_library_exchange(job, dist, dev, virtual)
    ranks = job.ranks if virtual else [job]
    # round 6: the tick's phases inside the library (smr_rsp_spread_*: fused encode + scatter, handlers, headers, loss masks -- and, with
    # the communicator bound, the exchanges: ONE C call per tick); SMR_L2_PYTHON_TICK=1 keeps the Python-driven tick of rounds 3-5
    lib_tick = os.environ.get("SMR_L2_PYTHON_TICK") is None
    if lib_tick:
        for rk in ranks:
            rk.use_library_tick()
    blocks = sorted({b for rk in ranks for b in rk.lead})
    srcs = {b: [torch.randint(0, 256, (shard.group_range(total, nr, b)[1] - shard.group_range(total, nr, b)[0], L), dtype=torch.uint8, device=dev)
                for _ in range(NB)] for b in blocks}
    ar = {b: torch.arange(srcs[b][0].shape[0], dtype=torch.int64, device=dev) for b in blocks}
    n_ticks = args.warmup + args.steps

    def step(t):
        job.tick({b: srcs[b][t % NB] for b in blocks}, {b: ((1 + t * total + ar[b]) & 0x3FFFFFFF).to(torch.int32) for b in blocks},
                 heartbeat=t % H == H - 1)
    commits_of = lambda: sum(rk.commits() for rk in ranks)       # noqa: E731
    for t in range(args.warmup):
        step(t)
    torch.cuda.synchronize()
    c0, sent0 = commits_of(), sum(rk.bytes_sent for rk in ranks)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.warmup, n_ticks):
        step(t)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    commits = commits_of() - c0
    sent = sum(rk.bytes_sent for rk in ranks) - sent0
    elapsed, commits = shard.reduce_metric(elapsed, commits, device=dev)
    p = ranks[0]._plans
    line = {"metric": "committed_slots_per_sec", "value": commits / elapsed, "unit": "slots/s", "n_gpus": world, "ranks": shard.count_ranks(dev),
            "backend": dist.get_backend() if world > 1 else None, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 (GF(2^8)) + u64 ballots", "data": "synthetic",
            "config": {"workload": "RSPaxos (f = 1), %d groups/GPU x 5 replicas, one 4 KiB Put per group per tick (L = %d), RS(3,2), heartbeats every %d ticks"
                                   % (G, L, H), "groups_per_gpu": G, "replicas": R, "layout": "spread-rspaxos", "spread_ranks": nr,
                       "ranks_are": "virtual (one process, one GPU: the collective is a device copy)" if virtual else "processes, one per GPU"},
            "exchange": {"via": via, "collectives_per_tick": "2 (Accepts + shards out, AcceptReplies back); 4 on a heartbeat tick",
                         "tick": ("smr_rsp_spread_tick: one C call per tick, exchanges inside" if (lib_tick and comm is not None) else
                                  "smr_rsp_spread_segment: one C call per segment (3; 6 on a heartbeat tick), the host moves the exchange's buffers" if lib_tick
                                  else "summerset_amd/spread_rsp.py: one ctypes call per handler + torch ops (SMR_L2_PYTHON_TICK)"),
                         "bytes_per_exchange_per_rank": {k: int(sum(x["in_split"])) for k, x in p.items()},
                         "bytes_sent_per_tick_per_rank": sent / args.steps / len(ranks),
                         "rs_payload_GiBps": len(blocks) * G * L * args.steps / 2**30 / elapsed * (1 if virtual else world)},
            "roofline": None, "cpu_baseline": None}
    if rank == 0:
        emit_line(line)
    for rk in ranks:
        rk.close_library_tick()
    if comm is not None:
        job.bind_comm(None)
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def spread_epaxos_main(args, torch, dist, rank, local, world, dev):
    """--layout spread-epaxos: BASELINE config 5 as written -- EPaxos, args.groups groups per GPU x 5 replicas, every replica
    proposes one instance per group per tick on Zipf(0.99) keys of 64, the replicas of a group on different ranks
    (summerset_amd/spread_ep.py: five all_to_all_single per tick).  At world 1 the ranks are virtual."""
    from summerset_amd import shard, spread_ep
    R, W, K = 5, 32, 64
    virtual = world == 1
    nr = args.spread_ranks if virtual else world
    total = args.groups * (1 if virtual else world)
    # dependency-graph execution ON with the 5-exchange schedule: the co-located loop's phase-by-phase order (tests/test_zzy_spread_ep_gpu.py)
    kw = dict(window=W, n_keys=K, execute=True, ordered=False)
    job = spread_ep.in_process(total, R, nr, dev, **kw) if virtual else spread_ep.SpreadEPaxos(total, R, rank, world, dev, **kw)
    comm, via = _bind_library_exchange(job, dist, dev, virtual)
    # round 6: the tick itself inside the library (smr_ep_spread_*: schedule, message plan, packing -- and, with the communicator
    # bound, the exchanges: ONE C call per tick; virtual ranks / torch collectives: one call per segment); SMR_L2_PYTHON_TICK=1
    # keeps the Python-driven tick of rounds 3-5
    lib_tick = os.environ.get("SMR_L2_PYTHON_TICK") is None
    if lib_tick:
        for rk in (job.ranks if virtual else [job]):
            rk.use_library_tick()
    homes = [k for rk in job.ranks for k in rk.reps] if virtual else list(job.reps)
    zipf = 1.0 / np.arange(1, K + 1) ** 0.99
    zipf /= zipf.sum()
    n_ticks = args.warmup + args.steps
    keys = []
    for t in range(min(n_ticks, 8)):                                   # keyed by (tick, block, replica): every rank draws its own replicas' keys
        keys.append({(b, r): torch.from_numpy(np.random.default_rng([0x5EED5EED, t, b, r]).choice(
            K, shard.group_range(total, nr, b)[1] - shard.group_range(total, nr, b)[0], p=zipf).astype(np.uint8)).to(dev) for b, r in homes})
    committed = torch.zeros((), dtype=torch.int64, device=dev)
    slow = torch.zeros((), dtype=torch.int64, device=dev)
    def one(t):
        for o in job.tick(keys[t % len(keys)]).values():
            committed.add_(o["committed"].sum())
            slow.add_((o["decision"] == 2).sum())
    for t in range(args.warmup):                                       # (the counting ops too: torch loads their kernels on first use)
        one(t)
    torch.cuda.synchronize()
    committed.zero_(); slow.zero_()
    sent0 = sum(rk.bytes_sent for rk in job.ranks) if virtual else job.bytes_sent
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.warmup, n_ticks):
        one(t)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    sent = (sum(rk.bytes_sent for rk in job.ranks) if virtual else job.bytes_sent) - sent0
    n_slow = int(slow.item())
    elapsed, commits = shard.reduce_metric(elapsed, int(committed.item()), device=dev)
    line = {"metric": "committed_instances_per_sec", "value": commits / elapsed, "unit": "instances/s", "n_gpus": world,
            "ranks": shard.count_ranks(dev), "backend": dist.get_backend() if world > 1 else None, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "EPaxos closed loop, %d groups/GPU x 5 replicas, every replica proposes 1 instance per group per tick "
                                   "(Zipf(0.99) keys of 64), optimized quorums, dependency-graph execution on" % args.groups,
                       "groups_per_gpu": args.groups, "replicas": R, "window": W, "layout": "spread", "spread_ranks": nr,
                       "ranks_are": "virtual (one process, one GPU: the collective is a device copy)" if virtual else "processes, one per GPU"},
            "exchange": {"via": via, "collectives_per_tick": 5, "bytes_sent_per_tick_per_rank": sent / args.steps / (nr if virtual else 1),
                         "tick": ("smr_ep_spread_tick: one C call per tick, exchanges inside" if (lib_tick and comm is not None) else
                                  "smr_ep_spread_segment: one C call per segment (6 per tick), the host moves the exchange's buffers" if lib_tick else
                                  "summerset_amd/spread_ep.py: one ctypes call per handler (SMR_L2_PYTHON_TICK)")},
            "legs_failed": (["l2_exchange"] if L2_EXCHANGE_FAILED else []),
            "slow_path_instances_this_rank": n_slow, "roofline": None, "cpu_baseline": None,
            "note": "correctness layout of config 5's inter-replica fan-out (handler calls of the Python driver included); the roofline / "
                    "cpu_baseline objects belong to the co-located line"}
    if rank == 0:
        emit_line(line)
    for rk in (job.ranks if virtual else [job]):
        rk.close_library_tick()
    if comm is not None:
        job.bind_comm(None)
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def colocated_epaxos_main(args, torch, dist, rank, local, world, dev):
    """--layout colocated-epaxos: BASELINE config 5 in layout L1 -- every rank holds ALL five replicas of its block of groups
    (args.groups per GPU) and runs the closed loop as one C-ABI call per tick (smr_ep_cluster_tick); no data-path collective,
    weak scaling like the headline line.  Dependency-graph execution on."""
    from summerset_amd import EPaxosReplicaGroup, ep_cluster, shard
    G, R, W, K = args.groups, 5, 32, 64
    lo, _ = shard.group_range(G * world, world, rank)
    reps = [EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K, execute=True) for r in range(R)]
    job = ep_cluster.EPaxosCluster(reps, phase_major=True)            # (the leaders' steps phase by phase: DESIGN §4, smr_ep_cluster_set_mode bit 1)
    zipf = 1.0 / np.arange(1, K + 1) ** 0.99
    zipf /= zipf.sum()
    n_ticks = args.warmup + args.steps
    keys = [[torch.from_numpy(np.random.default_rng([0x5EED5EED, t, lo, r]).choice(K, G, p=zipf).astype(np.uint8)).to(dev) for r in range(R)]
            for t in range(min(n_ticks, 8))]                              # keyed by (tick, my block's first group, replica)
    committed = torch.zeros((), dtype=torch.int64, device=dev)
    slow = torch.zeros((), dtype=torch.int64, device=dev)
    outs = job.new_outputs(dev)                                          # the caller's arrays, reused; the leaders' `committed` / `decision`
    com_all, dec_all = torch.zeros((R, G), dtype=torch.uint8, device=dev), torch.zeros((R, G), dtype=torch.uint8, device=dev)   # as rows of one tensor each
    for s_ in range(R):
        outs[s_]["committed"], outs[s_]["decision"] = com_all[s_], dec_all[s_]

    def one(t):
        job.tick(keys[t % len(keys)], out=outs)
        committed.add_(com_all.sum())
        slow.add_((dec_all == 2).sum())
    for t in range(args.warmup):                                         # (the counting ops too: torch loads their kernels on first use)
        one(t)
    torch.cuda.synchronize()
    committed.zero_(); slow.zero_()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.warmup, n_ticks):
        one(t)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    n_slow = int(slow.item())
    executed = sum(int(r.exec_dump()["counters"][0]) for r in reps)
    elapsed, commits = shard.reduce_metric(elapsed, int(committed.item()), device=dev)
    line = {"metric": "committed_instances_per_sec", "value": commits / elapsed, "unit": "instances/s", "n_gpus": world,
            "ranks": shard.count_ranks(dev), "backend": dist.get_backend() if world > 1 else None, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "EPaxos closed loop, %d groups/GPU x 5 replicas, every replica proposes 1 instance per group per tick "
                                   "(Zipf(0.99) keys of 64), optimized quorums, dependency-graph execution on" % G,
                       "groups_per_gpu": G, "replicas": R, "window": W, "layout": "colocated",
                       "launch": "one smr_ep_cluster_tick call = ONE launch per tick, the command leaders' steps phase by phase (smr_ep_cluster_set_mode 2)"},
            "slow_path_instances_this_rank": n_slow, "commands_executed_this_rank": executed, "roofline": None, "cpu_baseline": None,
            "note": "config 5 in layout L1; the roofline / cpu_baseline objects belong to the headline line"}
    if rank == 0:
        emit_line(line)
    job.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def self_spawn(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks exactly as the driver's own
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` would and hand
    rank 0's JSON line (the only thing the ranks print on stdout) through."""
    import subprocess
    from summerset_amd import shard
    cmd = shard.launch_command(n, os.path.abspath(__file__), sys.argv[1:])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS="1")
    sys.stderr.write("bench.py: --gpus %d without a launcher: %s\n" % (n, " ".join(cmd)))
    raise SystemExit(subprocess.call(cmd, env=env))


def launch_check(rank, local, world):
    import torch
    import torch.distributed as dist
    gpu = torch.cuda.is_available()
    if gpu and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d devices are visible" % (world, torch.cuda.device_count()))
    dev = torch.device("cuda", local) if gpu else torch.device("cpu")
    if gpu:
        torch.cuda.set_device(local)
    ranks = 1
    if world > 1:
        dist.init_process_group("nccl", device_id=dev) if gpu else dist.init_process_group("gloo")
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one)
        ranks = int(one.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks": ranks, "backend": ("nccl" if gpu else "gloo") if world > 1 else None}))


def main():
    args = parse()
    from summerset_amd import shard
    plan = shard.resolve_world(args.gpus)          # raises when the launcher's world is not --gpus
    if plan[0] == "spawn":
        self_spawn(plan[1])
    _, rank, local, world = plan
    if args.launch_check:
        return launch_check(rank, local, world)
    import torch
    import torch.distributed as dist
    if args.leg == "l2":                           # child of the headline run at N = 1: the spread layout on virtual ranks
        torch.cuda.set_device(local)
        line = spread_run(args, torch, dist, 0, 1, torch.device("cuda", local), steps=args.steps, warmup=4)
        # the same layout without leader changes: in L2 every round is a launch the whole job waits for, so ONE group's leader
        # change (a chain of ~100 us of wave-cooperative handlers, on the side stream in L1) is the round's length for everybody;
        # the steady figure is what the layout itself costs
        import copy
        a2 = copy.copy(args)
        a2.timeouts = 0.0
        st = spread_run(a2, torch, dist, 0, 1, torch.device("cuda", local), steps=args.steps, warmup=4)
        line["steady_state"] = {"ms_per_tick": st["ms_per_step"], "value": st["value"], "unit": st["unit"],
                                "note": "--timeouts 0: no leader change in flight (layout L1's steady tick is ~0.056 ms for the same groups)"}
        print(json.dumps(line))
        return
    if args.leg:                                   # child of leg_isolated(): one secondary leg, own process
        torch.cuda.set_device(local)
        legs = {"rspaxos": rspaxos_leg, "epaxos_cluster": epaxos_cluster_leg, "epaxos_execution": epaxos_exec_leg, "rspaxos_replica": rspaxos_replica_leg,
                "craft_leader": craft_leader_leg, "quorum_read": quorum_read_leg, "wire_ingest": wire_ingest_leg, "reply_ingest": reply_ingest_leg,
                "rspaxos_payload": rspaxos_payload_leg, "craft_payload": craft_payload_leg}
        print(json.dumps(legs[args.leg](torch, torch.device("cuda", local))))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    if torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d devices are visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    from summerset_amd import MultiPaxosCluster, stream
    if args.layout == "spread":
        return spread_main(args, torch, dist, rank, local, world, dev)
    if args.layout == "spread-epaxos":
        return spread_epaxos_main(args, torch, dist, rank, local, world, dev)
    if args.layout == "colocated-epaxos":
        return colocated_epaxos_main(args, torch, dist, rank, local, world, dev)
    if args.layout == "spread-rspaxos":
        return spread_rspaxos_main(args, torch, dist, rank, local, world, dev)

    G, R, S, W, H = args.groups, 5, args.slots, args.window, args.hb_every
    cap = W + 4
    reps = n_regions(args)
    n_timed = args.warmup + reps * args.steps     # warm-up, then `reps` timed regions of exactly --steps ticks each
    n_ticks = n_timed + args.round_ticks          # the per-round pass goes on where the timed region stopped
    if args.fused:
        args.straggler_ticks = 0                  # the fused tick kernel and the side stream exclude each other
    # cluster, stream and launch mode come from summerset_amd/workloads.py: the SAME helpers the BASELINE-size parity tests
    # call (tests/test_baseline_configs_gpu.py::test_headline_*_bench_launch), so the timed shape is the checked shape
    from summerset_amd import workloads
    eng = workloads.headline_cluster(G, W=W, R=R, straggler_ticks=args.straggler_ticks, role_rotation=bool(args.role_rotation))
    st = workloads.headline_stream(G, n_timed, timeout_frac(args), timeout_span(args), S=S, W=W, R=R, H=H, drop_p=args.drop,
                                   group_base=shard.group_range(G * world, world, rank)[0])   # my block of the job's groups
    # inputs resident in HBM before the clock starts
    pool = []
    for t in range(args.pool):
        x = st.tick(t)
        pool.append({k: torch.from_numpy(x[k]).to(dev) for k in ("req_cnt", "req_val", "ackctl")})
    events = []
    for t in range(n_ticks):
        e = st.tick_events(t)
        events.append({k: torch.from_numpy(v).to(dev) for k, v in e.items() if isinstance(v, np.ndarray)})

    fired = [bool((st.timeout_tick == t).any()) for t in range(n_ticks)]   # host-side fact: did any timer fire

    def tick_args(t):
        p, e = pool[t % args.pool], events[t]
        return dict(timeout_rep=e["timeout_rep"] if fired[t] else None, timeout_src=e["timeout_src"] if fired[t] else None,
                    req_target=e["req_target"], req_cnt=p["req_cnt"], req_val=p["req_val"], ackctl=p["ackctl"],
                    heartbeat=st.heartbeat(t))

    launches = []                                 # fused: (event pair, ticks) of every launch of the timed region

    def run(t0_, t1_, timed=False):
        if args.batch and not args.fused:
            # event pairs around the round kernels of ONE batch of the timed region (the last: the shortest when the
            # steps do not divide): they cost launch-gap time on every tick they cover
            workloads.drive_headline(eng, tick_args, t0_, t1_, batch=args.batch,
                                     before_call=lambda i, n, ch: eng.profile_enable(timed and i == n - 1))
            return
        if not args.fused:
            # event pairs on every third tick (all tick phases come by): the events themselves cost launch-gap time
            workloads.drive_headline(eng, tick_args, t0_, t1_, batch=0,
                                     before_call=(lambda i, n, ch: eng.profile_enable(i % 3 == 0)) if timed else None)
            return
        for b0 in range(t0_, t1_, args.fused):
            batch = [tick_args(t) for t in range(b0, min(b0 + args.fused, t1_))]
            chunks = [batch[i:i + 16] for i in range(0, len(batch), 16)]       # one launch per 16 ticks
            for ch in chunks:
                if timed:                         # HIP events on the launch stream (torch's current stream IS the
                    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # stream the
                    ea.record()                   # kernel is launched on: the mirror passes it to the C-ABI)
                eng.run_ticks(ch)
                if timed:
                    eb.record()
                    launches.append((ea, eb, len(ch)))

    run(0, args.warmup)
    torch.cuda.synchronize()
    regions = []                                  # (seconds, commits) of every timed region: MAX / SUM over ranks
    for i in range(reps):
        a = args.warmup + i * args.steps
        c0 = sum(eng.counters(r)["commits"] for r in range(R))
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(a, a + args.steps, timed=True)        # EXACTLY --steps ticks
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        eng.profile_enable(False)
        c1 = sum(eng.counters(r)["commits"] for r in range(R))
        regions.append(shard.reduce_metric(el, c1 - c0, device=dev))
    elapsed, commits = sorted(regions)[len(regions) // 2]        # the median region (by time)
    rej = sum(eng.counters(r)["rejects"] for r in range(R))
    overflow = int(eng.dump(0)["overflow"].sum()) if G <= 4096 else None
    # untimed: the same workload a few ticks further through the per-round kernels, an event pair around each
    if args.fused and args.round_ticks:
        eng.profile_enable(True)
        for t in range(n_timed, n_ticks):
            eng.tick(**tick_args(t))
        torch.cuda.synchronize()
        eng.profile_enable(False)
    prof = {}
    for i, name in enumerate(("R1_local", "R2_deliver", "R3_replies", "R4_heartbeat", "mp_quorum_tally")):
        ms, n = eng.profile_read(i)
        prof[name] = {"avg_us": (ms / n * 1e3) if n else None, "launches": int(n)}
    alg_tick = G * (52 * S + 33)                  # algorithmic bytes of one tick's quorum decisions
    tick_us = elapsed / args.steps * 1e6
    pmc = pmc_file()
    # HBM bytes per tick from the committed PMC passes (per launch; the heartbeat round runs every H-th tick)
    pmc_tick = None
    if pmc and S == 32 and G == 65536:
        k = pmc["kernels"]
        if args.fused and "smr::mp_ticks_fused<5, 5>" in k:
            pmc_tick = k["smr::mp_ticks_fused<5, 5>"]["hbm_bytes_per_launch"] / pmc.get("ticks_per_fused_launch", 16)
        elif not args.fused:
            tpb = float(pmc.get("ticks_per_batch", 8))           # the list's launches: one per batch
            per = {"smr::mp_round_heartbeat": H, "smr::mp_straggler_batch": tpb, "smr::mp_mark_batch": tpb}
            names = ("smr::mp_round_local", "smr::mp_round_deliver_all", "smr::mp_round_deliver", "smr::mp_round_deliver_rest", "smr::mp_quorum_tally<5>", "smr::mp_round_replies",
                     "smr::mp_round_heartbeat") + (("smr::mp_straggler_batch", "smr::mp_mark_batch") if args.batch else
                                                   ("smr::mp_straggler_tick", "smr::mp_mark_stragglers"))
            pmc_tick = sum(k[n]["hbm_bytes_per_launch"] / per.get(n, 1) for n in names if n in k)
    if args.fused:
        # the dominant kernel of the path is the fused tick kernel itself: every launch of the timed region between
        # its own HIP event pair; algorithmic bytes = the §8(d) figure x the decisions of the launch's ticks
        us = [ea.elapsed_time(eb) * 1e3 for ea, eb, _ in launches]
        nt = sum(n for _, _, n in launches)
        avg_launch_us = sum(us) / max(len(us), 1)
        alg_launch = alg_tick * nt / max(len(us), 1)
        achieved = alg_launch / (avg_launch_us * 1e-6) / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_tick * nt / max(len(us), 1) if pmc_tick else None,
                "traffic_source": PMC_NOTE, "kernel": "mp_ticks_fused<5, 5>", "launches": len(us), "ticks_per_launch": nt / max(len(us), 1),
                "alg_bytes_per_launch": alg_launch, "avg_launch_us": avg_launch_us}
    else:
        qt_us = prof["mp_quorum_tally"]["avg_us"]
        achieved = alg_tick / (qt_us * 1e-6) / 1e9
        t_qt = pmc["kernels"].get("smr::mp_quorum_tally<5>") if pmc and S == 32 else None
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": (t_qt["hbm_bytes_per_launch"] * (G / 65536.0)) if t_qt else None, "traffic_source": PMC_NOTE,
                "kernel": "mp_quorum_tally", "alg_bytes_per_launch": alg_tick, "avg_launch_us": qt_us}
        # ... and on the bytes the kernel really moves (since round 4 the ballot runs are not stored, and `frac` is a
        # figure on 8(d) bytes the kernel does not touch: it is latency-bound, and this is the number that says so)
        roof["frac_pmc"] = (roof["traffic"] / (qt_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if roof["traffic"] and qt_us else None
    # the whole tick against the roofline, both ways: on the §8(d) algorithmic bytes of its decisions and on the HBM
    # bytes the tick really moves (PMC); and the quorum kernel alone, from the per-round pass
    roof["whole_tick"] = {"us": tick_us, "frac_alg": alg_tick / (tick_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                          "hbm_bytes_pmc": pmc_tick,
                          "frac_pmc": (pmc_tick / (tick_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if pmc_tick else None}
    if prof["mp_quorum_tally"]["avg_us"]:
        qt_us = prof["mp_quorum_tally"]["avg_us"]
        t_qt = pmc["kernels"].get("smr::mp_quorum_tally<5>") if pmc and S == 32 else None
        roof["quorum_kernel_alone"] = {"kernel": "mp_quorum_tally", "avg_launch_us": qt_us, "alg_bytes_per_launch": alg_tick,
                                       "frac": alg_tick / (qt_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                       "traffic": (t_qt["hbm_bytes_per_launch"] * (G / 65536.0)) if t_qt else None,
                                       "pass": "untimed per-round pass of %d ticks behind the timed region" % args.round_ticks
                                               if args.fused else ("timed region, last batch" if args.batch else "timed region, every third tick")}
    line = {
        "metric": "committed_slots_per_sec", "value": commits / elapsed, "unit": "slots/s",
        "n_gpus": world, "ranks": shard.count_ranks(dev), "backend": dist.get_backend() if world > 1 else None,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "timed_regions": {"n": reps, "reported": "median region (by time)", "ms_per_step": [e / args.steps * 1e3 for e, _ in regions],
                          "value": [c / e for e, c in regions],
                          "note": "every region is exactly --steps ticks between its own barrier + synchronize pair, the regions run "
                                  "back to back on one cluster (region i starts where i - 1 stopped); leader-timeout rate per tick constant"},
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "MultiPaxos lock-step, %d groups/GPU x 5 replicas, S=%d new slots/group/tick, "
                               "heartbeat every %d ticks, %.0f%% ack loss (<= 2 lost per slot), %s"
                               % (G, S, H, args.drop * 100, timeouts_text(args)),
                   "groups_per_gpu": G, "replicas": R, "slots_per_tick": S, "window": W, "layout": "colocated",
                   "launch": ("fused tick kernel, <= %d ticks per launch" % min(args.fused, 16)) if args.fused else
                             ("batches of <= %d ticks per smr_mp_run_ticks call: five per-round launches per tick for the bulk, one "
                              "side-stream launch per batch for the straggler list" % min(args.batch, 16)) if args.batch else
                             "five per-round launches per tick + one side-stream launch per tick for the straggler list"},
        "roofline": roof,
        "kernels": prof, "kernels_pass": "untimed per-round pass" if args.fused else "timed region",
        "rejected_batches": rej, "overflow_groups": overflow,
        "generic_path_batches": [eng.debug_generic_units(r) for r in range(R)],
        "straggler_list": dict(zip(("capacity", "wanted_by_last_mark_pass"), eng.straggler_stats())) if args.straggler_ticks else None,
        "decisions_per_sec": G * S * args.steps / elapsed,
    }
    # Layout L2 in the line the driver runs: the same workload a few ticks in the spread layout -- the replicas of every group on
    # different ranks, every protocol message through one all_to_all_single per exchange (RCCL over xGMI at N > 1; at N = 1 the
    # job's four ranks are virtual: same kernels, plans and buffers, the collective a device copy).  Collective: every rank runs it.
    l2 = None
    if not args.no_l2:
        try:
            del eng
            torch.cuda.empty_cache()
            if world == 1:                         # virtual ranks: a child process (its quarter-size launches stay out of this process' kernel trace)
                x = leg_isolated("l2", extra=["--groups", str(args.groups), "--slots", str(args.slots), "--window", str(args.window), "--steps",
                                              str(max(4, min(args.steps, 12))), "--spread-ranks", str(args.spread_ranks)])
            else:
                # Real ranks: the first time this pass meets RCCL is the driver's scaling run.  It runs on a watchdog thread: if it
                # has not come back in SMR_BENCH_L2_TIMEOUT seconds (a collective that never completes), the headline line --
                # measured above, nothing of it depends on this pass -- is printed with l2 = {"error": "timeout"} and the
                # process leaves through os._exit instead of the closing barrier.
                import threading
                box = {}

                def work():
                    try:
                        torch.cuda.set_device(dev)
                        box["x"] = spread_run(args, torch, dist, rank, world, dev, steps=max(4, min(args.steps, 12)), warmup=4)
                    except BaseException as e:     # noqa: BLE001
                        box["e"] = e
                th = threading.Thread(target=work, daemon=True)
                th.start()
                th.join(timeout=float(os.environ.get("SMR_BENCH_L2_TIMEOUT", "150")))
                if th.is_alive():
                    raise TimeoutError("the spread pass did not finish in time on rank %d" % rank)
                if "e" in box:
                    raise box["e"]
                x = box["x"]
            l2 = {"layout": "spread (SURVEY 8e L2): replica r of block b on rank (b + r) mod N", "ranks": x["config"]["spread_ranks"],
                  "ranks_are": x["config"]["ranks_are"], "value": x["value"], "unit": "slots/s", "ms_per_tick": x["ms_per_step"],
                  "steps": x["steps"], "warmup": x["warmup"], "exchange": x["exchange"], "backend": x["backend"]}
            if "steady_state" in x:
                l2["steady_state"] = dict(x["steady_state"], ms_per_tick_per_virtual_rank=x["steady_state"]["ms_per_tick"] / max(x["config"]["spread_ranks"], 1))
            if world == 1:                         # all the virtual ranks' kernels ran one after the other on this ONE GPU
                l2["ms_per_tick_per_virtual_rank"] = x["ms_per_step"] / max(x["config"]["spread_ranks"], 1)
                l2["note"] = ("virtual ranks share one GPU and one stream order: ms_per_tick is the SUM of the ranks' work (each holds "
                              "groups / ranks groups' five replicas in pieces); a rank's share = ms_per_tick / ranks, before the wire")
        except Exception as e:                     # noqa: BLE001 -- never at the headline's cost; named in legs_failed
            l2 = {"error": "%s: %s" % (type(e).__name__, e)}
            sys.stderr.write("bench.py: the l2 pass FAILED: %s: %s\n" % (type(e).__name__, e))
    if rank == 0:
        failed = []
        if l2 is not None:
            line["l2"] = l2
            if "error" in l2:
                failed.append("l2")

        def leg(name, fn, *a):                     # a secondary leg must never cost the headline line -- but it must not
            try:                                   # fail silently either: `legs_failed` names it at the top level, stderr says why
                line[name] = fn(*a)
            except Exception as e:                 # noqa: BLE001
                line[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                failed.append(name)
                sys.stderr.write("bench.py: leg %s FAILED: %s: %s\n" % (name, type(e).__name__, e))
        if world > 1:                              # secondary legs and the CPU baseline belong to the N=1 line only: the
            args.no_cpu = args.no_rs = args.no_extra = True   # other ranks sit in the closing barrier meanwhile
        if not args.no_cpu:
            leg("cpu_baseline", cpu_leg, args, args.cpu_seconds)
        if not args.no_rs:
            leg("rs_encode", rs_leg, torch, dev, not args.no_cpu, args.cpu_seconds)
        if not args.no_extra:
            leg("raft_quorum", raft_leg, torch, dev)
            leg("epaxos_fast_quorum", epaxos_leg, torch, dev)
            leg("epaxos_cluster", leg_isolated, "epaxos_cluster")
            leg("rspaxos", leg_isolated, "rspaxos")
            leg("rspaxos_payload", leg_isolated, "rspaxos_payload")
            leg("craft_payload", leg_isolated, "craft_payload")
            leg("repnothing", repnothing_leg)
            leg("wire_ingest", leg_isolated, "wire_ingest")
            leg("reply_ingest", leg_isolated, "reply_ingest")
            if args.late_legs:
                leg("epaxos_execution", leg_isolated, "epaxos_execution")
                leg("rspaxos_replica", leg_isolated, "rspaxos_replica")
                leg("craft_leader", leg_isolated, "craft_leader")
                leg("quorum_read", leg_isolated, "quorum_read")
            if not args.no_cpu:                    # their CPU baselines sit inside the legs' objects
                for name, fn in (("raft_quorum", raft_cpu_baseline), ("epaxos_fast_quorum", epaxos_cpu_baseline), ("rspaxos", rspaxos_cpu_baseline)):
                    if isinstance(line.get(name), dict) and "error" not in line[name]:
                        try:
                            line[name]["cpu_baseline"] = _on_all_cores(fn, seconds=2.0)   # one thread AND one process per core
                        except Exception as e:     # noqa: BLE001
                            line[name]["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
                            failed.append(name + ".cpu_baseline")
        line["legs_failed"] = failed               # [] = every leg that was asked for ran (a missing cpu_baseline / roofline is an error here)
        emit_line(line)
        sys.stdout.flush()
    if world > 1:
        # Every rank learns whether ANY rank's L2 pass failed or hung before it picks the way out: through the job's
        # key-value store, not through the backend that may just have hung.  A rank whose own pass was fine used to walk into
        # the plain barrier and wait there for a peer that had already left.
        bad = _l2_outcome_of_all_ranks(dist, world, l2 is not None and "error" in l2)
        if bad:
            import threading                       # the closing barrier gets a few seconds on a thread, then every rank leaves (the line is out)
            th = threading.Thread(target=lambda: dist.barrier(), daemon=True)
            th.start()
            th.join(timeout=20.0)
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)                            # 0: the headline line is valid and printed; the failure is in `legs_failed` and on stderr
        dist.barrier()
        dist.destroy_process_group()


def _l2_outcome_of_all_ranks(dist, world, mine_failed, wait_s=None):
    """True if any rank's L2 pass failed (or the question itself could not be settled).  Each rank adds itself to `l2_done` and its
    failure to `l2_failed` in the process group's store, then waits until all `world` ranks are in: the longest a rank can be behind is
    the watchdog's own timeout."""
    import time
    try:
        from torch.distributed import distributed_c10d as c10d
        store = c10d._get_default_store()
        store.add("smr_bench_l2_failed", 1 if mine_failed else 0)
        store.add("smr_bench_l2_done", 1)
        limit = time.time() + (float(os.environ.get("SMR_BENCH_L2_TIMEOUT", "150")) + 30.0 if wait_s is None else wait_s)
        while store.add("smr_bench_l2_done", 0) < world:
            if time.time() > limit:
                return True
            time.sleep(0.05)
        return store.add("smr_bench_l2_failed", 0) > 0
    except Exception as e:                         # noqa: BLE001 -- no store, or its host is gone: do not trust a collective
        sys.stderr.write("bench.py: could not settle the ranks' L2 outcome (%s: %s)\n" % (type(e).__name__, e))
        return True if mine_failed else False


if __name__ == "__main__":
    main()
