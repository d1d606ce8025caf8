import sys; sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C
from summerset_amd import MultiPaxosCluster, stream
from summerset_amd._lib import check
G,R,S,W,H=65536,5,32,512,4
cap=W+4
eng=MultiPaxosCluster(G,R,W,win_reserve=W//8,outbox_cap=cap); eng.preset_leader(0)
st=stream.MultiPaxosStream(G,R,S,cap=cap,n_ticks=8,drop_p=0.1,timeout_frac=0.0,hb_every=H,rand_rows=S+4,max_drop=2)
dev=torch.device('cuda')
pool=[{k:(torch.from_numpy(v).to(dev) if isinstance(v,np.ndarray) else v) for k,v in st.tick(t).items()} for t in range(2)]
for t in range(7):
    x=dict(pool[t%2]); x['heartbeat']=st.heartbeat(t); eng.tick(**x)
    torch.cuda.synchronize()
    out=np.zeros(64,np.uint64); check(eng._L.smr_mp_debug_stamps(eng._h, out.ctypes.data_as(C.c_void_p)))
    o=out.astype(np.int64).reshape(8,8)
    base=o[0,0]
    print('tick',t,'hb',st.heartbeat(t),' wave0 stamps (us from start):',[round((int(v)-int(base))/100,1) if v else None for v in o[0,:7]], '| w3:',[round((int(v)-int(base))/100,1) if v else None for v in o[3,:5]])
