#!/bin/bash
mkdir -p gpurun_out/final_r4
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/final_r4/full_gpu_tests.log 2>&1; tail -3 gpurun_out/final_r4/full_gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
LDN_LIB_PATH=$PWD/laudnet_amd/libldn_hip_debug.so timeout 900 python -m pytest tests/test_hip_plan.py tests/test_hip_ops.py -x -q -m gpu -k "plan or index or mask or pool or layer or hint" 2>&1 | tail -2
