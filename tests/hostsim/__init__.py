"""Kernel-source simulation for the CPU test suite (TEST INFRASTRUCTURE ONLY).

`build()` compiles the engine from the very .hip sources that ship, for the host, against
tests/hostsim/hip/hip_runtime.h (a SIMT emulation: lanes as fibers, wavefront meetings, lock-step memory
order; hipsim_rt.cpp), into tests/hostsim/_build/; `patched()` points the package's ctypes handle at that
library for the duration of a test, so the Python mirror and the C-ABI entry points under test are the
shipped ones and only the "device" is simulated.  See the header of the shim for what this does and does
not show."""
import contextlib
import ctypes as C
import os
import re
import struct
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_CSRC = os.path.join(_ROOT, "summerset_amd", "csrc")
_OUT = os.path.join(_HERE, "_build")
SOURCES = ["core.hip", "rs_kernels.hip", "mp_engine.hip", "raft_engine.hip", "ep_engine.hip", "ep_spread.hip", "rsp_engine.hip", "rsp_spread.hip", "rsp_payload.hip", "rep_nothing.hip", "wire.hip", "wire_ingest.hip", "wire_ingest_replies.hip", "wire_emit.hip", "qread.hip", "kv_exec.hip", "heartbeater.hip", "skv_exec.hip", "leaseman.hip", "comm.hip"]
LIB = os.path.join(_OUT, "libsummerset_sim.so")
# clang: the RS kernels use ext_vector_type, which g++ does not have (this is the host compiler hipcc itself drives)
CXX = os.environ.get("HOSTSIM_CXX", "/opt/rocm/lib/llvm/bin/clang++")


# kernels: -fsanitize=thread only for the call it puts in front of every load / store (hipsim_rt.cpp provides the
# hooks; no sanitizer runtime is linked); no block placement; SMR_ARENA_GUARD: unowned gaps between an arena's arrays
_BASE = [CXX, "-O1", "-gline-tables-only", "-fno-optimize-sibling-calls", "-fno-omit-frame-pointer", "-std=c++17", "-fPIC",
         "-w", "-I", _HERE]
# STRAG_BATCH_BLOCKS: the side launch's grid -- 192 blocks of 512 lanes on the device, where a listed group has a block to itself; here a
# launch costs its lanes whether they have work or not, and 12 blocks x 6 lanes walk a test's list of 100-200 groups in several passes
# (the multi-pass loop the device's grid hardly ever enters)
_KERN = ["-DSMR_ARENA_GUARD=4096", "-DSTRAG_BATCH_BLOCKS=12", "-mllvm", "-disable-block-placement", "-fsanitize=thread", "-mllvm",
         "-tsan-instrument-func-entry-exit=0", "-mllvm", "-tsan-instrument-atomics=0", "-mllvm", "-tsan-instrument-memintrinsics=0"]


def build(defines=()):
    """`defines`: extra -D flags for the kernel sources (a kernel experiment, tools/experiments/README.md): own library"""
    global LIB
    if defines:
        saved = LIB
        LIB = os.path.join(_OUT, "libsummerset_sim_%s.so" % "_".join(d.replace("=", "-") for d in defines))
        try:
            return _build(tuple("-D" + d for d in defines))
        finally:
            LIB = saved
    return _build(())


def _build(extra):
    os.makedirs(_OUT, exist_ok=True)
    # one builder at a time: pytest-xdist workers all reach here when a source is newer than the library, and a worker that
    # linked while another was still compiling an object has handed out a half-written file before (FileNotFoundError, round 4)
    import fcntl
    with open(os.path.join(_OUT, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(extra)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(extra):
    srcs = [os.path.join(_CSRC, s) for s in SOURCES]
    rt = os.path.join(_HERE, "hipsim_rt.cpp")
    # the shipped csrc/comm.hip is compiled like every other source; what stands in is RCCL itself (rccl/rccl.h + rccl_sim.cpp:
    # sends and receives between the PROCESSES of one host through POSIX shared memory), so smr_comm_exchange's N > 1 path runs here
    comm = os.path.join(_HERE, "rccl_sim.cpp")
    deps = srcs + [rt, comm, os.path.join(_HERE, "rccl", "rccl.h"), os.path.join(_HERE, "hip", "hip_runtime.h"), os.path.join(_ROOT, "include", "summerset_hip.h")]
    deps += [os.path.join(_CSRC, h) for h in os.listdir(_CSRC) if h.endswith(".h")]
    deps.append(os.path.abspath(__file__))                       # (the flags above are part of the build)
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    base, kern = _BASE, _KERN
    objs = []
    hdrs = [d for d in deps if d.endswith(".h")] + [os.path.abspath(__file__)]

    def stale(obj, src):                                         # per object: a change to one source does not rebuild the other sixteen
        return not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in [src] + hdrs)
    for s in srcs:
        o = os.path.join(_OUT, os.path.basename(LIB) + "." + os.path.basename(s) + ".o")
        if stale(o, s):
            subprocess.run(base + kern + list(extra) + ["-c", "-x", "c++", s, "-o", o], check=True)
        objs.append(o)
    o = os.path.join(_OUT, "hipsim_rt.o")
    if stale(o, rt):
        subprocess.run(base + ["-c", rt, "-o", o], check=True)
    oc = os.path.join(_OUT, "rccl_sim.o")
    if stale(oc, comm):
        subprocess.run(base + ["-c", comm, "-o", oc], check=True)
    objs.append(oc)
    subprocess.run([CXX, "-shared", "-o", LIB + ".tmp"] + objs + [o, "-lrt"], check=True)
    _rank_sites(LIB + ".tmp", LIB + ".sites")
    os.replace(LIB + ".tmp", LIB)
    return LIB


def build_selftest():
    """tests/hostsim/selftest.cpp with the library's flags (its own copy of the runtime, its own sites file)"""
    os.makedirs(_OUT, exist_ok=True)
    exe = os.path.join(_OUT, "selftest")
    rt, src = os.path.join(_HERE, "hipsim_rt.cpp"), os.path.join(_HERE, "selftest.cpp")
    deps = [rt, src, os.path.join(_HERE, "hip", "hip_runtime.h"), os.path.abspath(__file__)]
    if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps):
        return exe
    subprocess.run(_BASE + _KERN + ["-c", "-x", "c++", src, "-o", exe + ".k.o"], check=True)
    subprocess.run(_BASE + ["-c", rt, "-o", exe + ".rt.o"], check=True)
    subprocess.run([CXX, "-pie", "-rdynamic", "-o", exe + ".tmp", exe + ".k.o", exe + ".rt.o", "-ldl"], check=True)
    _rank_sites(exe + ".tmp", exe + ".sites")
    os.replace(exe + ".tmp", exe)
    return exe


_XLANE = re.compile(r"call\w*\s+\S+\s+<(__tsan_(?:unaligned_)?(?:read|write)\d+|_Z\d+__(?:shfl|shfl_xor|ballot|all|any)\w*)(?:@plt)?>")


def _rank_sites(lib, out):
    """Source position of every parking site (the return address of a memory hook or of a cross-lane
    operation): the (line, column) of each inlined frame, outermost first.  Sorted, that is program order
    inside a function whatever the code layout; the scheduler lets the lane at the lowest rank go first.
    File: u64 count, then (u64 offset in the library, u64 rank) pairs sorted by offset."""
    tools = os.path.dirname(CXX)
    dis = subprocess.run([os.path.join(tools, "llvm-objdump"), "-d", "--no-show-raw-insn", lib], check=True,
                         capture_output=True, text=True).stdout.splitlines()
    sites, pending = [], False
    for ln in dis:
        head = ln.split(":", 1)
        is_insn = len(head) == 2 and head[0].strip() and all(c in "0123456789abcdef" for c in head[0].strip())
        if pending and is_insn:
            sites.append(int(head[0].strip(), 16))              # the instruction after the call = the return address
            pending = False
        if is_insn and _XLANE.search(ln):
            pending = True
    sym = subprocess.run([os.path.join(tools, "llvm-symbolizer"), "-e", lib, "--inlines", "--functions=none"],
                         input="\n".join(hex(a - 1) for a in sites) + "\n", check=True, capture_output=True, text=True).stdout
    keys, cur = [], []
    for ln in sym.splitlines():
        if not ln.strip():                                       # blank line = end of one address's frames (innermost first)
            keys.append(tuple(reversed(cur)))
            cur = []
            continue
        parts = ln.rsplit(":", 2)
        try:
            cur.append((int(parts[1]), int(parts[2])))
        except (IndexError, ValueError):
            cur.append((0, 0))
    assert len(keys) == len(sites), (len(keys), len(sites))
    rank = {k: i for i, k in enumerate(sorted(set(keys)))}
    with open(out, "wb") as f:
        f.write(struct.pack("<Q", len(sites)))
        for a, k in sorted(zip(sites, keys)):
            f.write(struct.pack("<QQ", a, rank[k]))


def load(defines=()):
    from summerset_amd import _lib
    lib = C.CDLL(build(defines))
    for name, res, args in _lib.SYMBOLS:
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


class _NullStream:
    """what `torch.cuda.current_stream()` hands back under the emulator: the null stream"""
    cuda_stream = 0


@contextlib.contextmanager
def patched(defines=()):
    """the package talks to the simulated library.  Nothing of the Python mirror is replaced: the
    mirror's own `_lib.stream_ptr(None)` runs and asks torch for the current stream, and only
    torch's answer (there is no device here) is the null stream."""
    import torch
    from summerset_amd import _lib
    sim = load(defines)
    spots = [(_lib, "_lib", sim), (torch.cuda, "current_stream", lambda device=None: _NullStream())]
    saved = [(o, n, o.__dict__[n]) for o, n, _ in spots]
    for o, n, v in spots:
        setattr(o, n, v)
    try:
        yield sim
    finally:
        for o, n, v in saved:
            setattr(o, n, v)
