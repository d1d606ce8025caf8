"""Builds libsummerset_hip.so (gfx950) in-tree with hipcc.

The built library lives next to the sources (summerset_amd/libsummerset_hip.so);
it is git-ignored but travels with gpurun snapshots.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsummerset_hip.so")
SOURCES = ["core.hip", "rs_kernels.hip", "mp_engine.hip", "raft_engine.hip", "ep_engine.hip", "ep_spread.hip", "rsp_engine.hip", "rsp_spread.hip", "rsp_payload.hip", "rep_nothing.hip", "wire.hip", "wire_ingest.hip", "wire_ingest_replies.hip", "wire_emit.hip", "qread.hip", "kv_exec.hip", "heartbeater.hip", "skv_exec.hip", "leaseman.hip", "comm.hip"]
HEADERS = ["smr_common.h", "mp_types.h", "mp_device.h", "rsp_peek.h", "raft_peek.h", "wire_rd.h", os.path.join("..", "..", "include", "summerset_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("SMR_EXTRA_HIPCC_FLAGS", "").split()       # e.g. -DSMR_JOB_STAMPS for tools/dbg_stamps.py


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for s in srcs:
        o = os.path.join(CSRC, os.path.basename(s).replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        # librccl: the L2 exchange behind the C-ABI (csrc/comm.hip).  Its soname (librccl.so.1) is the one PyTorch's own copy
        # carries, so a process that imported torch first runs both on ONE RCCL.
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
