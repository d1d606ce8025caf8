"""RS encode timing probe: input / output alignment, launch size, and a plain-copy ceiling."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from summerset_amd import _lib
from summerset_amd._lib import check
dev = torch.device("cuda")
L_ = _lib.load()
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
st = torch.cuda.current_stream().cuda_stream
for n, L in ((16384, 4099), (16384, 4128), (65536, 4099)):
    slen = (L + 2) // 3
    for name, cws, pss in (("tight rows, parity packed", L, slen), ("tight rows, parity shards 16-aligned", L, (slen + 15) // 16 * 16),
                           ("rows 16-aligned, parity shards 16-aligned", (L + 15) // 16 * 16, (slen + 15) // 16 * 16)):
        data = torch.randint(0, 256, (n, cws), dtype=torch.uint8, device=dev)
        par = torch.zeros((n, 2 * pss), dtype=torch.uint8, device=dev)
        fn = lambda: check(L_.smr_rs_encode(data.data_ptr(), L, cws, n, 3, 2, par.data_ptr(), 2 * pss, pss, st))
        us = timeit(fn)
        alg = n * 5 * slen
        print("n=%d L=%d %s: %.1f us = %.0f GB/s (%.1f%% of 8 TB/s)" % (n, L, name, us, alg / us / 1e3, alg / us / 1e3 / 80), flush=True)
