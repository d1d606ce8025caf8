"""The L2 exchange through the library: RCCL send / receive pairs behind the C-ABI (`smr_comm_*`, csrc/comm.hip).

Stand-in for `TransportHub::send_msg` / `bcast_msg` (src/server/transport.rs:208-275).  A `Comm` is one rank's end of the
job's communicator; `exchange` is an all-to-all with split sizes on device buffers, enqueued on a HIP stream -- the call
`torch.distributed.all_to_all_single` made for the spread layouts before round 4, now an entry point a Rust
`GenericReplica` host can bind as well.  Bootstrap: one rank makes the 128-byte id, the host's control
channel ships it (`from_torch_distributed`: a broadcast over whatever process group is up -- gloo or nccl)."""
import ctypes as C

from . import _lib
from ._lib import check, stream_ptr

ID_BYTES = 128
SUM, MAX = 0, 1


class Comm:
    def __init__(self, unique_id, rank, world):
        """blocks until every one of the `world` ranks has called it; the rank's device must be current (torch.cuda.set_device)"""
        if len(unique_id) != ID_BYTES:
            raise ValueError("the communicator id is %d bytes" % ID_BYTES)
        self._L = _lib.load()
        self.rank, self.world = int(rank), int(world)
        buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(bytes(unique_id))
        h = C.c_void_p()
        check(self._L.smr_comm_init_rank(buf, ID_BYTES, self.rank, self.world, C.byref(h)))
        self._h = h

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * ID_BYTES)()
        check(_lib.load().smr_comm_unique_id(buf, ID_BYTES))
        return bytes(buf)

    @classmethod
    def from_torch_distributed(cls, device):
        """rank 0 of the default process group makes the id and broadcasts it; every rank joins.  `device`: where the
        broadcast's tensor lives (the rank's cuda device under nccl, "cpu" under gloo)."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        t = torch.zeros(ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            t = torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8).clone()
        t = t.to(device)
        dist.broadcast(t, src=0)
        return cls(bytes(t.cpu().numpy().tobytes()), rank, world)

    def close(self):
        if getattr(self, "_h", None):
            self._L.smr_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def exchange(self, sbuf, in_split, rbuf, out_split, stream=None, self_via_rccl=False):
        """segment k of `sbuf` (in_split[k] bytes, back to back in rank order) to rank k; out_split[k] bytes from rank k into
        segment k of `rbuf` -- `all_to_all_single(rbuf, sbuf, out_split, in_split)` on uint8 device tensors"""
        if len(in_split) != self.world or len(out_split) != self.world:
            raise ValueError("split sizes are per rank")
        a = (C.c_uint64 * self.world)(*[int(x) for x in in_split])
        b = (C.c_uint64 * self.world)(*[int(x) for x in out_split])
        check(self._L.smr_comm_exchange(self._h, sbuf.data_ptr() if sbuf is not None else None, a,
                                        rbuf.data_ptr() if rbuf is not None else None, b, 1 if self_via_rccl else 0, stream_ptr(stream)))

    def all_reduce(self, t, op=SUM, stream=None):
        """in place over the ranks; t: device tensor of 64-bit words reduced as UNSIGNED (ncclUint64): uint64, or int64 holding
        non-negative values only -- MAX over negative int64 values is refused here rather than answered wrongly"""
        import torch
        if t.dtype not in (torch.int64, torch.uint64):
            raise TypeError("all_reduce takes 64-bit integer tensors")
        if t.dtype == torch.int64 and t.numel() and bool((t < 0).any()):
            raise ValueError("all_reduce reduces unsigned words: negative int64 values are not supported")
        check(self._L.smr_comm_all_reduce_u64(self._h, t.data_ptr(), t.numel(), int(op), stream_ptr(stream)))
        return t

    def info(self):
        arr = (C.c_uint64 * 5)()
        check(self._L.smr_comm_info(self._h, C.byref(arr)))
        return dict(rank=int(arr[0]), world=int(arr[1]), exchanges=int(arr[2]), bytes_sent=int(arr[3]), bytes_received=int(arr[4]))
