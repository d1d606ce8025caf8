#!/bin/bash
# r3z: what pass 2's stores cost: the shipped build, a build whose 64 lanes write side by side (wrong addresses), a build that stores nothing  (the -DSMR_WI_DIAG builds it names were removed from wire_ingest.hip after this run)
mkdir -p gpurun_out
R=$PWD; export PYTHONPATH=$R
cat > gpurun_out/r3z_wi_time.py <<'P'
import torch, numpy as np, bench
from summerset_amd import wire
dev = torch.device('cuda:0')
G, S = 65536, 32
n_conn = G * 4
hb = np.frombuffer(wire.heartbeat(0x101, 300, 290, 0), np.uint8); one = np.frombuffer(wire.accept_reply(300, 0x101), np.uint8)
has_hb = (np.arange(n_conn) % 4 == 0); lens = S * 17 + has_hb * len(hb)
off = np.zeros(n_conn + 1, np.int64); off[1:] = np.cumsum(lens)
body = np.tile(one, (S, 1)).reshape(-1); four = np.concatenate([hb, body, body, body, body])
bufs = [torch.from_numpy(np.tile(four, n_conn // 4)).to(dev) for _ in range(4)]
ing = wire.MpIngest(n_conn, n_conn * S, n_conn, 16, device=dev)
d_off = torch.from_numpy(off).to(dev); d_grp = torch.from_numpy((np.arange(n_conn) // 4).astype(np.int32)).to(dev); d_peer = torch.from_numpy((1 + np.arange(n_conn) % 4).astype(np.uint8)).to(dev)
for k in range(4): ing.ingest(bufs[k], d_off, d_grp, d_peer)
print("call_us", bench._time_us(torch, lambda i: ing.ingest(bufs[i % 4], d_off, d_grp, d_peer), 12))
P
for v in "" wi_diag1 wi_diag2; do
  [ -n "$v" ] && export SUMMERSET_HIP_LIB=$R/summerset_amd/variants/libsummerset_hip_$v.so
  echo "== ${v:-shipped}"
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_z && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_z -o z -- python $R/gpurun_out/r3z_wi_time.py 2>&1 | grep call_us )
  python tools/rocpd_summary.py /tmp/prof_z --only wire_ingest 2>&1 | cut -c1-130 | tail -2
done 2>&1 | tee gpurun_out/r3z.log
