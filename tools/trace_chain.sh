#!/bin/bash
# builds the LDN_TRACE library (tuning only, git-ignored) and prints the phase split of the chained stage-3 launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/tools/ablate
[ -f $R/tools/ablate/libldn_trace.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DLDN_TRACE -o $R/tools/ablate/libldn_trace.so $R/laudnet_amd/csrc/*.hip || exit 1
LDN_LIB_PATH=$R/tools/ablate/libldn_trace.so python $R/tools/trace_chain.py
