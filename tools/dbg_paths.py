import sys; sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
from summerset_amd import MultiPaxosCluster, stream
from summerset_amd._lib import check
G,R,S,W,H=65536,5,32,512,4
cap=W+4
eng=MultiPaxosCluster(G,R,W,win_reserve=W//8,outbox_cap=cap); eng.preset_leader(0)
st=stream.MultiPaxosStream(G,R,S,cap=cap,n_ticks=40,drop_p=0.1,timeout_frac=0.01,hb_every=H,rand_rows=S+4,max_drop=2,timeout_span=4)
dev=torch.device('cuda')
pool=[{k:(torch.from_numpy(v).to(dev) if isinstance(v,np.ndarray) else v) for k,v in st.tick(t).items()} for t in range(4)]
def stamps():
    out=np.zeros(64,np.uint64); check(eng._L.smr_mp_debug_stamps(eng._h, out.ctypes.data_as(C.c_void_p)))
    return out.astype(np.int64)
prev=stamps()
for t in range(24):
    x=dict(pool[t%4]); x['heartbeat']=st.heartbeat(t)
    ev=st.tick_events(t); x.update({k:torch.from_numpy(v).to(dev) for k,v in ev.items()})
    eng.tick(**x); torch.cuda.synchronize()
    o=stamps(); d=o-prev; prev=o
    if t in (0,1,5,6,7,8,15,16,22,23):
        print('tick',t,'R1fast/row',d[32:37],'R1gen/row',d[40:45],'R2fast/row',d[48:53],'R2any/row',d[56:61])
