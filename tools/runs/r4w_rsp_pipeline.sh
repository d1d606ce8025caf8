#!/bin/bash
# r4w: does the RSPaxos tick overlap with the next batch's encode?  eager, two streams, device time per tick
export PYTHONPATH=$PWD
mkdir -p gpurun_out
python - <<'P' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4w.log
import torch, numpy as np, time
from summerset_amd import RSPaxosReplicaGroup, rsp_cluster
dev = torch.device("cuda:0")
G, R, W, L, NB, H = 16384, 5, 64, 4113, 4, 4
from summerset_amd import RSCodewordBatch
reps = [RSPaxosReplicaGroup(G, R, me=r, window=W, fault_tolerance=1) for r in range(R)]
for r in reps: r.preset_leader(0)
loop = rsp_cluster.SteadyLoop(reps, leader=0, one_launch=True)
srcs = [torch.randint(0, 256, (G, L), dtype=torch.uint8, device=dev) for _ in range(NB)]
cws = [RSCodewordBatch(G, L, 3, 2, device=dev, zero=False) for _ in range(NB)]
ar = torch.arange(G, dtype=torch.int64, device=dev); base = torch.ones((), dtype=torch.int64, device=dev)
vals = [((base + ar + k * G) & 0x3FFFFFFF).to(torch.int32) for k in range(64)]
def serial(n):
    for k in range(n):
        loop.encode(srcs[k % NB], out=cws[k % NB], slot=0)
        loop.tick(vals[k % 64], heartbeat=k % H == H - 1)
s_enc = torch.cuda.Stream()
def piped(n):
    cur = torch.cuda.current_stream()
    loop.encode(srcs[0], out=cws[0], slot=0)
    for k in range(n):
        if k + 1 < n:
            s_enc.wait_stream(cur)
            with torch.cuda.stream(s_enc):
                loop.encode(srcs[(k + 1) % NB], out=cws[(k + 1) % NB], slot=(k + 1) & 1)
        loop.tick(vals[k % 64], heartbeat=k % H == H - 1)
        if k + 1 < n:
            cur.wait_stream(s_enc)
for name, fn in (("serial", serial), ("two streams", piped), ("serial", serial), ("two streams", piped)):
    fn(8); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); fn(48); e1.record(); torch.cuda.synchronize()
    print(name, "device ms/tick %.4f" % (e0.elapsed_time(e1) / 48), "host ms/tick %.4f" % ((time.perf_counter() - t0) / 48 * 1e3))
# the two kernels alone
for name, fn in (("encode+fanout alone", lambda k: loop.encode(srcs[k % NB], out=cws[k % NB], slot=0)), ("tick alone", lambda k: loop.tick(vals[k % 64], heartbeat=k % H == H - 1))):
    for k in range(4): fn(k)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(32): fn(k)
    e1.record(); torch.cuda.synchronize()
    print(name, "device us %.1f" % (e0.elapsed_time(e1) / 32 * 1e3))
P
