"""summerset_amd -- MI355X-native batched multi-group consensus engine.

Host-side mirror of the one hot path of josehu07/summerset this project
accelerates: the MultiPaxos / Raft / RSPaxos / EPaxos quorum-check +
log-advance + commit loop over thousands of independent replica groups, and the
RS(3,2) GF(2^8) erasure encode.  All compute happens in hand-written HIP
kernels inside libsummerset_hip.so, reached through the C-ABI of
include/summerset_hip.h; PyTorch only provides device buffers and streams.
"""
from ._lib import SummersetError, SMR_CTL_IDENTITY, SMR_NO_REPLICA  # noqa: F401
from .rscoding import RSCodewordBatch, rs_matrix, rs_shard_len  # noqa: F401
from .multipaxos import MultiPaxosCluster  # noqa: F401
from .quorumread import KvStateMachine, QuorumReadGroup, StringKvStateMachine  # noqa: F401
from .raft import CRaftLeaderGroup, RaftLeaderGroup  # noqa: F401
from .epaxos import EPaxosReplicaGroup  # noqa: F401
from .rspaxos import RSPaxosReplicaGroup  # noqa: F401
from .rsp_payload import CRaftPayloadStore, RSPaxosPayloadStore, RSPaxosReplicaWithPayload  # noqa: F401
from .repnothing import RepNothingReplica  # noqa: F401
from .heartbeater import Heartbeater  # noqa: F401
from .leaseman import LeaseManager  # noqa: F401
from . import shard, stream  # noqa: F401
