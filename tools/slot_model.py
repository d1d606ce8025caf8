#!/usr/bin/env python3
"""Register-file arithmetic behind DESIGN 9.1: how many wavefronts of a bulk kernel fit on a CU that also holds a block of
the side launch (mp_straggler_batch), and what share of a launch's wavefronts the chip can hold at once.  Static: VGPR
counts come from tools/kernel_resources.py's table; the measured times beside it are the record's (profiles/r9z_*).

    python tools/slot_model.py profiles/r9z_kernel_resources.txt [n_side_blocks] > profiles/r9l_slot_model.txt

Model: 256 CUs x 4 SIMDs x 512 VGPRs, allocation granule 8, at most 8 wavefronts per SIMD.  A side block is 5 wavefronts
placed 2 / 1 / 1 / 1 on its CU's SIMDs, one block per listed group, at most one per CU while the list is shorter than the
CUs.  The driver's command lists 70-110 groups per batch of 8 ticks: 9.1 groups per tick meet a leader timeout and stay
listed for 4 more ticks (`straggler_list` in bench_detail.json: 67 wanted by the run's last mark pass); 192 = the launch's
grid, every block with a group.  A bulk launch of the headline shape (65 536 groups x 5 replicas) is 5120 wavefronts (the tally: 4096)."""
import re
import sys

CUS, SIMDS, FILE, GRAN, MAXW = 256, 4, 512, 8, 8


def alloc(v):
    return (v + GRAN - 1) // GRAN * GRAN


def per_cu(v, side_v=0, side=(0, 0, 0, 0)):
    return sum(min(MAXW - k, (FILE - k * alloc(side_v)) // alloc(v)) for k in side)


def main():
    table = open(sys.argv[1]).read()
    n_side = int(sys.argv[2]) if len(sys.argv) > 2 else 90
    vg = {}
    for line in table.splitlines():
        m = re.match(r"\S+\s+(?:void )?(\S+(?:<[^>]*>)?)\s+(\d+)\s+\d+\s+\d+\s+(\d+)\s+(\d+)\s+\d+\s*$", line)
        if m:
            vg[m.group(1)] = int(m.group(2))
    side_v = vg["mp_straggler_batch"]
    # kernel, wavefronts per launch, measured us alone / beside the side launch (profiles/r9z_kernel_stats_default_bench.txt, r9z_bench_steady.json)
    rows = [("mp_round_local", 5120, 12.2, 16.0), ("mp_round_deliver_all", 5120, 15.4, 26.1), ("mp_quorum_tally<5>", 4096, 17.0, 23.6),
            ("mp_round_deliver", 5120, None, None)]
    print("# side block: 5 wavefronts x %d VGPRs (%d allocated), 2 / 1 / 1 / 1 on a CU; %d of %d CUs hold one" % (side_v, alloc(side_v), n_side, CUS))
    print("# %-22s %5s %9s %9s %10s %10s %8s %14s" % ("kernel", "VGPR", "free CU", "side CU", "chip alone", "chip beside", "launch", "model / measured"))
    for name, waves, t0, t1 in rows:
        v = vg[name]
        a, b = per_cu(v), per_cu(v, side_v, (2, 1, 1, 1))
        cap0, cap1 = CUS * a, (CUS - n_side) * a + n_side * b
        need0, need1 = max(1.0, waves / cap0), max(1.0, waves / cap1)
        meas = "" if t0 is None else "%.2fx / %.2fx" % (need1 / need0, t1 / t0)
        print("  %-22s %5d %9d %9d %10d %10d %8d %14s" % (name, v, a, b, cap0, cap1, waves, meas))
    print("# model = (wavefronts of the launch / wavefronts the chip holds beside the side blocks) over the same alone: the launch's length in")
    print("# passes of the chip; measured = the kernel's average beside the side launch (driver's command) over its steady-state average.")
    print("# What a cap would buy (wavefronts per side CU at n VGPRs):", ", ".join("%d: %d" % (n, per_cu(n, side_v, (2, 1, 1, 1))) for n in (64, 72, 80, 88, 96, 128)))
    print("# ... and a side kernel of n VGPRs (R2 wavefronts per side CU):", ", ".join("%d: %d" % (n, per_cu(vg["mp_round_deliver_all"], n, (2, 1, 1, 1))) for n in (216, 208, 168, 128)))
    print("# ... and a side block of 4 / 3 / 2 wavefronts (R2 wavefronts per side CU):", ", ".join(
        "%s: %d" % (s, per_cu(vg["mp_round_deliver_all"], side_v, s)) for s in ((1, 1, 1, 1), (1, 1, 1, 0), (1, 1, 0, 0))))


if __name__ == "__main__":
    main()
