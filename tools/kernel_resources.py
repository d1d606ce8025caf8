#!/usr/bin/env python3
"""Static resource table of every kernel of libsummerset_hip.so, from the gfx950 compiler itself
(`-Rpass-analysis=kernel-resource-usage`, device pass only): VGPRs, AGPRs, SGPRs, scratch (spill) bytes per lane,
LDS bytes per block, and the occupancy (waves per SIMD) the register / LDS budget allows.  No GPU needed.

    python tools/kernel_resources.py > profiles/round1/r1h_kernel_resources.txt

It is what can be said about a kernel that has not run on the device yet: whether it spills, and how many
wavefronts can be in flight to hide its HBM latency.  Not a measurement."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from summerset_amd import build as B  # noqa: E402

FIELDS = ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.split("\n")[:len(names)] if out.returncode == 0 else names


def main():
    rows = []
    for src in B.SOURCES:
        path = os.path.join(B.CSRC, src)
        cmd = [B.HIPCC] + B.FLAGS + ["--offload-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", os.devnull]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur = None
        for line in err.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"src": src, "name": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = int(m.group(2))
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["name"] = re.sub(r"\(.*", "", n).replace("smr::", "").replace("(anonymous namespace)::", "")
    print("# gfx950, flags: %s" % " ".join(B.FLAGS))
    print("%-16s %-44s %5s %5s %5s %8s %8s %5s" % ("source", "kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS", "occ"))
    for r in rows:
        print("%-16s %-44s %5d %5d %5d %8d %8d %5d" % ((r["src"], r["name"][:44]) + tuple(r.get(f, -1) for f in FIELDS)))
    spills = [r["name"] for r in rows if r.get(FIELDS[3], 0) > 0]
    print("# kernels with scratch: %s" % (", ".join(spills) if spills else "none"))


if __name__ == "__main__":
    main()
