#!/bin/bash
# tuning only: tools/trace_chain.py over every tools/ablate/libldn_tr_*.so (built with tools/build_ablate.sh ... -DLDN_TRACE -DLDN_MASK_HASH=607 ...)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for f in $R/tools/ablate/libldn_tr_*.so; do
  echo "##### $(basename $f)"
  LDN_LIB_PATH=$f timeout 300 python $R/tools/trace_chain.py 2>&1 | grep -v amdgpu.ids
done
