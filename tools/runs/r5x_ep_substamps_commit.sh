#!/bin/bash
# where one PreAccept handler's ~14 us go (EPC_STAMPS build with sub-stamps inside ep_acceptor_lane_in)
export SUMMERSET_HIP_LIB=$PWD/summerset_amd/variants/libsummerset_hip_epc_stamps.so
timeout 300 python tools/dbg_epc_stamps.py 1 pm 2>&1 | grep -v "amdgpu.ids" | cut -c1-400
