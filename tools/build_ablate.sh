#!/bin/bash
# tuning only: libldn_hip variants with one ablation macro set for ONE translation unit (results are wrong, timings are the point)
# usage: tools/build_ablate.sh <name> <source.hip> <-DMACRO=V ...>   ->  tools/ablate/libldn_<name>.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; src=$2; shift 2
mkdir -p $R/tools/ablate
objs=""
for f in ldn_conv_image ldn_index ldn_regnet ldn_tail ldn_dense ldn_stem ldn_attn ldn_grouped ldn_small ldn_rows3; do
  if [ "$f.hip" = "$src" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c -o $R/tools/ablate/$f.$name.o $R/laudnet_amd/csrc/$f.hip
    objs="$objs $R/tools/ablate/$f.$name.o"
  else
    objs="$objs $R/laudnet_amd/_obj/$f.rel.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ablate/libldn_$name.so $objs
