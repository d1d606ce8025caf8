"""HBM bytes per launch of every kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE: they do not fit one pass,
MI355X_MICROARCH.md "rocprofv3 PMC slots"), collected with --kernel-trace only and --output-format csv.
Corrections exactly as that guide prescribes: counter unit = KB; on gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced read -> doubled; WRITE_SIZE as reported; both checked against the RS encode whose bytes are known exactly.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> "<what ran>" > profiles/NAME_pmc_traffic.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter:
                    acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        if not name.startswith("smr::") and "smr::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "")
        f, w = fetch.get(name, []), write.get(name, [])
        rd = 2.0 * 1000.0 * (sum(f) / len(f)) if f else 0.0
        wr = 1000.0 * (sum(w) / len(w)) if w else 0.0
        kernels[short] = {"launches": max(len(f), len(w)), "fetch_size_kb_avg": sum(f) / len(f) if f else None,
                          "write_size_kb_avg": sum(w) / len(w) if w else None, "hbm_read_bytes_per_launch": rd,
                          "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    cal = kernels.get("smr::rs_matmul_xtime<2, 4>")
    note = None
    if cal:
        n, L, sl = 65536, 4099, 1367
        note = ("rs_matmul_xtime<2,4> on 65536 x 4099 B: reads %.1f MB vs %.1f MB payload, writes %.1f MB vs %.1f MB expected"
                % (cal["hbm_read_bytes_per_launch"] / 1e6, n * L / 1e6, cal["hbm_write_bytes_per_launch"] / 1e6, n * 2 * sl / 1e6))
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace only): " + sys.argv[3],
               "corrections": "counter unit = KB (1000 B); FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 reports half of a wide "
                              "coalesced read); WRITE_SIZE as reported", "calibration": note, "ticks_per_fused_launch": 16, "ticks_per_batch": 8, "kernels": kernels}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
