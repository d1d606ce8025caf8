# needs a stamps build:  SMR_EXTRA_HIPCC_FLAGS=-DSMR_JOB_STAMPS python summerset_amd/build.py --force
import sys; sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
from summerset_amd import MultiPaxosCluster, stream
from summerset_amd._lib import check
G,R,S,W,H=65536,5,32,512,4
cap=W+4
eng=MultiPaxosCluster(G,R,W,win_reserve=W//8,outbox_cap=cap); eng.preset_leader(0)
st=stream.MultiPaxosStream(G,R,S,cap=cap,n_ticks=14,drop_p=0.1,timeout_frac=0.0,hb_every=H,rand_rows=S+4,max_drop=2)
st.timeout_tick[:] = -1
st.timeout_tick[1000] = 6      # one group times out at tick 6 (two ticks after a heartbeat)
dev=torch.device('cuda')
pool=[{k:(torch.from_numpy(v).to(dev) if isinstance(v,np.ndarray) else v) for k,v in st.tick(t).items()} for t in range(2)]
def stamps():
    out=np.zeros(64,np.uint64); check(eng._L.smr_mp_debug_stamps(eng._h, out.ctypes.data_as(C.c_void_p)))
    return out.astype(np.int64)
eng.profile_enable(True)
prev=[0.0]*4
for t in range(12):
    x=dict(pool[t%2]); x['heartbeat']=st.heartbeat(t)
    ev=st.tick_events(t); x.update({k:torch.from_numpy(v).to(dev) for k,v in ev.items()})
    eng.tick(**x); torch.cuda.synchronize()
    o=stamps()
    cur=[eng.profile_read(i)[0] for i in range(4)]
    print('   rounds us:',[round((c-p)*1e3) for c,p in zip(cur,prev)]); prev=cur
    f=lambda a,b: round((int(o[b])-int(o[a]))/100,1)
    print('tick',t,'R1job load/bal/store',f(8,9),f(9,10),f(10,11),'| R2job load/gen/store',f(16,17),f(17,18),f(18,19),'| R3job load/prep/acc/store',f(24,25),f(25,26),f(26,27),f(27,28))
