# needs a stamps build (tools/build_file_variant.sh stamps mp_engine.hip -DSMR_JOB_STAMPS); the bench's launch shape: batches of 8,
# straggler list on -- ONE group times out, so the stamps of the cooperative jobs are that group's (the last job of each kind wins)
import sys; sys.path.insert(0, '.')
import numpy as np, torch, ctypes as C
from summerset_amd import stream, workloads
from summerset_amd._lib import check
G, R, S, W, H = 65536, 5, 32, 512, 4
cap = W + 4
eng = workloads.headline_cluster(G, W=W, R=R, straggler_ticks=4)
st = stream.MultiPaxosStream(G, R, S, cap=cap, n_ticks=32, drop_p=0.1, timeout_frac=0.0, hb_every=H, rand_rows=S + 4, max_drop=2)
st.timeout_tick[:] = -1
st.timeout_tick[1000] = 18     # one group times out at tick 18 (two ticks after a heartbeat), the third tick of the third batch
dev = torch.device('cuda')
pool = [{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in st.tick(t).items()} for t in range(2)]
def stamps():
    out = np.zeros(64, np.uint64); check(eng._L.smr_mp_debug_stamps(eng._h, out.ctypes.data_as(C.c_void_p)))
    return out.astype(np.int64)
def tick_in(t):
    x = dict(pool[t % 2]); x['heartbeat'] = st.heartbeat(t)
    ev = st.tick_events(t); x.update({k: torch.from_numpy(v).to(dev) for k, v in ev.items()})
    if not (ev['timeout_rep'] != 0xFF).any():
        x['timeout_rep'] = x['timeout_src'] = None
    return x
for b in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.run_ticks([tick_in(t) for t in range(8 * b, 8 * b + 8)]); e1.record(); torch.cuda.synchronize()
    o = stamps()
    f = lambda a, c: round((int(o[c]) - int(o[a])) / 100, 1)
    print('batch', b, '| last R1job load/bal/store', f(8, 9), f(9, 10), f(10, 11), '| last R2job load/gen/store', f(16, 17), f(17, 18), f(18, 19),
          '| last R3job load/prep/acc/store', f(24, 25), f(25, 26), f(26, 27), f(27, 28), '| msg_prepare end_run/check_leader/pad+last/loop', f(40, 41), f(41, 42), f(42, 43), f(43, 44))
    print('         longest job so far R1-timeout / R2 / R3 us', [round(int(o[48 + k]) / 100, 1) for k in range(3)], 'jobs', [int(o[52 + k]) for k in range(3)],
          'mean us', [round(int(o[56 + k]) / 100 / max(int(o[52 + k]), 1), 1) for k in range(3)])
