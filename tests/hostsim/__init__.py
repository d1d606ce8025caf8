"""Kernel-source simulation for the CPU test suite (TEST INFRASTRUCTURE ONLY).

`build()` compiles the lane-independent engines (Raft, EPaxos) from the very .hip sources that ship,
for the host, against tests/hostsim/hip/hip_runtime.h, into tests/hostsim/_build/; `patched()` points
the package's ctypes handle at that library for the duration of a test, so the Python mirror and the
C-ABI entry points under test are the shipped ones and only the "device" is simulated.  See the header
of the shim for what this does and does not show."""
import contextlib
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_CSRC = os.path.join(_ROOT, "summerset_amd", "csrc")
_OUT = os.path.join(_HERE, "_build")
SOURCES = ["core.hip", "raft_engine.hip", "ep_engine.hip"]
LIB = os.path.join(_OUT, "libsummerset_sim.so")


def build():
    os.makedirs(_OUT, exist_ok=True)
    srcs = [os.path.join(_CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(_HERE, "hip", "hip_runtime.h"), os.path.join(_CSRC, "smr_common.h"),
                   os.path.join(_ROOT, "include", "summerset_hip.h")]
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-w", "-I", _HERE, "-o", LIB]
    for s in srcs:
        cmd += ["-x", "c++", s]
    subprocess.run(cmd, check=True)
    return LIB


def load():
    from summerset_amd import _lib
    lib = C.CDLL(build())
    for name, res, args in _lib.SYMBOLS:
        fn = getattr(lib, name, None)       # the simulated library holds only the lane-independent engines
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    return lib


@contextlib.contextmanager
def patched():
    """the package talks to the simulated library; streams are the null stream"""
    from summerset_amd import _lib, epaxos, raft
    sim = load()
    saved = (_lib._lib, epaxos.EPaxosReplicaGroup._stream, raft.RaftLeaderGroup._stream)
    _lib._lib = sim
    epaxos.EPaxosReplicaGroup._stream = staticmethod(lambda stream: 0)
    raft.RaftLeaderGroup._stream = staticmethod(lambda stream: 0)
    try:
        yield sim
    finally:
        _lib._lib, epaxos.EPaxosReplicaGroup._stream, raft.RaftLeaderGroup._stream = saved
