# r7i: the payload store's device tests again (the random-call test is new) and the C++ example
mkdir -p gpurun_out
T=r7i
timeout 300 python -m pytest tests/test_zz_rsp_payload_gpu.py tests/test_zzz_example_rsp_payload_gpu.py -m gpu -x -q -p no:cacheprovider --durations=4 2>&1 | tail -10 > gpurun_out/${T}_payload_tests.log; tail -4 gpurun_out/${T}_payload_tests.log
