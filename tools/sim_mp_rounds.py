"""Per-round byte counts of the MultiPaxos cluster engine on the emulator (tests/hostsim), default build vs the prepared
compile-time experiments together (tools/experiments/README.md): every load / store outside a lane's stack is counted, no
caches -- an upper bound of a kernel's HBM traffic, good for comparing builds.  Usage: python tools/sim_mp_rounds.py"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostsim, numpy as np
import test_mp_gpu as t
from summerset_amd import MultiPaxosCluster, stream
variants = [(), ("SMR_ACK_BITS","SMR_SKIP_REG_OUTBOX","SMR_BAL_RUN","SMR_BAL_LAZY","SMR_STATUS_LAZY")]
for defs in variants:
    with hostsim.patched(defines=defs) as lib:
        G,R,S,W=128,5,32,512
        cap=W+4
        eng=MultiPaxosCluster(G,R,W,win_reserve=W//8,outbox_cap=cap)
        eng.preset_leader(0)
        st=stream.MultiPaxosStream(G,R,S,cap=cap,n_ticks=12,drop_p=0.1,timeout_frac=0.0,hb_every=4,rand_rows=S+4,max_drop=2)
        out=(C.c_ulonglong*4)()
        acc={}
        def meas(name, fn):
            lib.hipsim_traffic(out,1); fn(); lib.hipsim_traffic(out,0)
            a=acc.setdefault(name,[0,0,0]); a[0]+=out[0]; a[1]+=out[1]; a[2]+=1
        for tick in range(12):
            inp=t._to_dev(st.tick(tick),"cpu")
            if tick<4:
                eng.tick(**inp); continue
            hb = (tick % 4) == 3
            meas("local", lambda: eng.round_local(inp.get("timeout_rep"), inp.get("timeout_src"), inp.get("req_target"), inp.get("req_cnt"), inp.get("req_val")))
            meas("deliver", lambda: eng.round_deliver())
            meas("replies+tally", lambda: eng.round_replies(inp.get("ackctl"), hb))
            if hb: meas("heartbeat", lambda: eng.round_heartbeat())
            eng.end_tick()
        print(defs or "default")
        for k,(l,s_,n) in acc.items():
            print("  %-14s per launch: loaded %8.1f B/group-slot stored %8.1f   (%d launches)"%(k, l/n/(G*S), s_/n/(G*S), n))
