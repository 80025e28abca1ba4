#!/bin/bash
O=gpurun_out/r4f; mkdir -p $O
KEEP=0.6066 LDN_LIB_PATH=tools/ablate/libldn_trace.so timeout 300 python tools/trace_chain.py 2>&1 | grep -v amdgpu.ids | tee $O/trace_chain.log
