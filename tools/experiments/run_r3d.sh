R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
LDN_DENSE16=4 timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "dense16" 2>&1 | tail -n 3
for v in 0 4; do
 for shape in "50176 512 1024" "50176 2048 1024" "200704 256 512" "12544 2048 2048" "401408 512 256"; do
  LDN_DENSE16=$v python tools/exp_dense.py $shape 2>&1 | grep -v amdgpu.ids | tail -n 2
 done
done
