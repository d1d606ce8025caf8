#!/bin/bash
# round 5, second call: the whole -m gpu suite on the tree with the CRaft payload store and the library exchange for every L2 layout
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rxX -p no:cacheprovider --durations=8 2>&1 | tail -40 > gpurun_out/r8b_gputests.log
tail -4 gpurun_out/r8b_gputests.log
