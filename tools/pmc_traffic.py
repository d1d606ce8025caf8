"""HBM bytes per launch of every kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE: they do not fit one pass,
MI355X_MICROARCH.md "rocprofv3 PMC slots"), collected with --kernel-trace only and --output-format csv.
Corrections exactly as that guide prescribes: counter unit = KB; on gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced read -> doubled; WRITE_SIZE as reported; both checked against the RS encode whose bytes are known exactly.

Keyed by (kernel, grid size) since round 4: the probe runs some kernels at two sizes (the RS encode
at 65 536 codewords for the calibration and at 16 384 for config 4), and one average over both described neither.
`kernels[name]` is the entry of the grid size with the most launches (ties: the larger grid) and carries `grids` = how many
sizes the kernel ran at; `kernels[name]["by_grid"][grid]` holds every size.  bench.py picks the size it timed
(`pmc_traffic(kernel, pick=...)`).  The calibration uses the LARGEST rs_matmul_xtime<2, 4> launches only.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> "<what ran>" > profiles/NAME_pmc_traffic.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    """(kernel name, grid size) -> the counter's value of every launch"""
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter:
                    grid = row.get("Grid_Size") or row.get("Grid_Size_X") or "0"
                    acc[(row["Kernel_Name"], int(float(grid)))].append(float(row["Counter_Value"]))
    return acc


def entry(f, w):
    rd = 2.0 * 1000.0 * (sum(f) / len(f)) if f else 0.0
    wr = 1000.0 * (sum(w) / len(w)) if w else 0.0
    return {"launches": max(len(f), len(w)), "fetch_size_kb_avg": sum(f) / len(f) if f else None,
            "write_size_kb_avg": sum(w) / len(w) if w else None, "hbm_read_bytes_per_launch": rd,
            "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}


def summarise(fetch, write):
    names = sorted({k[0] for k in fetch} | {k[0] for k in write})
    kernels = {}
    for name in names:
        if "smr::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "")
        grids = sorted({k[1] for k in list(fetch) + list(write) if k[0] == name})
        by = {str(g): entry(fetch.get((name, g), []), write.get((name, g), [])) for g in grids}
        main_g = max(grids, key=lambda g: (by[str(g)]["launches"], g))
        e = dict(by[str(main_g)])
        e["grid"], e["grids"], e["by_grid"] = main_g, len(grids), by
        kernels[short] = e
    return kernels


def calibration(kernels):
    cal = kernels.get("smr::rs_matmul_xtime<2, 4>")
    if not cal:
        return None
    big = cal["by_grid"][str(max(int(g) for g in cal["by_grid"]))]          # the 65 536-codeword launches only
    n, L, sl = 65536, 4099, 1367
    rd, wr = big["hbm_read_bytes_per_launch"], big["hbm_write_bytes_per_launch"]
    return {"text": "rs_matmul_xtime<2,4> on 65536 x 4099 B (%d launches of the largest grid): reads %.1f MB vs %.1f MB payload, writes %.1f MB vs "
                    "%.1f MB expected" % (big["launches"], rd / 1e6, n * L / 1e6, wr / 1e6, n * 2 * sl / 1e6),
            "read_ratio": rd / (n * L), "write_ratio": wr / (n * 2 * sl)}


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    kernels = summarise(fetch, write)
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace only): " + sys.argv[3],
               "corrections": "counter unit = KB (1000 B); FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 reports half of a wide "
                              "coalesced read); WRITE_SIZE as reported", "keyed_by": "(kernel, grid size): `by_grid`; the top-level figures of a "
                              "kernel are those of its most-launched grid size", "calibration": calibration(kernels),
               "ticks_per_fused_launch": 16, "ticks_per_batch": 8, "kernels": kernels}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
