"""the RSPaxos lock-step cluster driver lives in the package (summerset_amd/rsp_cluster.py); tests use it with the
CPU oracle's objects or the engine's"""
from summerset_amd.rsp_cluster import *          # noqa: F401,F403
from summerset_amd.rsp_cluster import NumpyEngine, tick, NULL, NO_REP   # noqa: F401
