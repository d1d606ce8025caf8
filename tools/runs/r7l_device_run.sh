# r7l: the C++ example at config 4's size -- what the recovery path of the payload store costs (the new leader's follows behind the ReconstructReplies)
mkdir -p gpurun_out
T=r7l
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Iinclude examples/rsp_payload_loop.cpp -Lsummerset_amd -lsummerset_hip -Wl,-rpath,$PWD/summerset_amd -o /tmp/rsp_payload_loop
timeout 120 /tmp/rsp_payload_loop 16384 4113 > gpurun_out/${T}_rsp_payload_loop_16384x4113.log 2>&1; cat gpurun_out/${T}_rsp_payload_loop_16384x4113.log
timeout 60 /tmp/rsp_payload_loop 1024 1000 | tail -4
