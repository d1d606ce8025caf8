"""Large-shape parity run (bench.py's shape): HIP engine vs CPU oracle, compared every few ticks.
usage: python tests/parity_big.py [groups] [timeout_frac] [timeout_span] [ticks] [straggler_ticks]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
import torch

from oracle import oracle as O
from tests.test_mp_gpu import _run_bench_shape

a = sys.argv[1:]
G = int(a[0]) if len(a) > 0 else 4096
frac = float(a[1]) if len(a) > 1 else 0.05
span = int(a[2]) if len(a) > 2 else 4
nt = int(a[3]) if len(a) > 3 else 72
strag = int(a[4]) if len(a) > 4 else 0
O.build()
eng, orc = _run_bench_shape(torch.device("cuda"), O, G, frac, span, nt, strag, log=lambda m: print(m, flush=True))
print("commits", [eng.counters(r)["commits"] for r in range(5)])
