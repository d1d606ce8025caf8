"""Builds a compile-time variant of the engine next to the shipped library, for device A/B runs
(tools/experiments/README.md):  python tools/build_variant.py ackbits -DSMR_ACK_BITS -DSMR_SKIP_REG_OUTBOX
-> summerset_amd/variants/libsummerset_hip_ackbits.so; use it with SUMMERSET_HIP_LIB=<that path> (the override
summerset_amd/_lib.py keeps for exactly this)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from summerset_amd import build as B   # noqa: E402


def main():
    tag, flags = sys.argv[1], sys.argv[2:]
    out_dir = os.path.join(B.HERE, "variants", tag)
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    only = os.environ.get("VARIANT_ONLY", "").split()      # e.g. VARIANT_ONLY=mp_engine.hip: the other objects are the shipped build's
    for s in B.SOURCES:
        if only and s not in only:
            objs.append(os.path.join(B.CSRC, s.replace(".hip", ".o")))
            continue
        o = os.path.join(out_dir, s.replace(".hip", ".o"))
        subprocess.check_call([B.HIPCC] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, s), "-o", o])
        objs.append(o)
    lib = os.path.join(B.HERE, "variants", "libsummerset_hip_%s.so" % tag)
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
    print(lib)


if __name__ == "__main__":
    main()
