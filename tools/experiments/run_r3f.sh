R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in base dabl1 dabl2 dabl4 dabl8 dabl5; do
  lib=$R/tools/ablate/libldn_$v.so; [ $v = base ] && lib=$R/laudnet_amd/libldn_hip.so
  echo "== $v"
  LDN_LIB_PATH=$lib python tools/exp_dense.py 50176 2048 1024 2>&1 | grep -v amdgpu.ids | tail -n 2
  LDN_LIB_PATH=$lib python tools/exp_dense.py 12544 2048 2048 2>&1 | grep -v amdgpu.ids | tail -n 1
done
