import sys; sys.path.insert(0,'.')
import numpy as np, torch
from summerset_amd import MultiPaxosCluster, stream
G,R,S,W,H=256,5,32,512,4
cap=W+4
eng=MultiPaxosCluster(G,R,W,win_reserve=W//8,outbox_cap=cap); eng.preset_leader(0)
st=stream.MultiPaxosStream(G,R,S,cap=cap,n_ticks=40,drop_p=0.1,timeout_frac=1.0,hb_every=H,rand_rows=S+4,max_drop=2)
st.timeout_tick[:] = 5
dev=torch.device('cuda')
for t in range(10):
    inp=st.tick(t)
    eng.tick(**{k:(torch.from_numpy(v).to(dev) if isinstance(v,np.ndarray) else v) for k,v in inp.items()})
    print(t, [eng.debug_generic_units(r) for r in range(R)], eng.counters(1), eng.counters(0)['commits'])
