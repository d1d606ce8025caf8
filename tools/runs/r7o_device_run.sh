# r7o: layout L2 of RSPaxos with the bytes in payload stores, every rank in one process, on the device
mkdir -p gpurun_out
T=r7o
timeout 200 python -m pytest tests/test_spread_rsp.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 > gpurun_out/${T}_spread_rsp_tests.log; tail -2 gpurun_out/${T}_spread_rsp_tests.log
