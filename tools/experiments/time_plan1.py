"""One configuration of the list build for a kernel trace: python time_plan1.py S H C mode(one|pm|decide|two)"""
import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from laudnet_amd import ops
S, H, C, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
B = 256
torch.manual_seed(0)
patch = (torch.rand(B, S, S, device="cuda") > 0.5).float()
if mode == "two":
    os.environ["LDN_INDEX_PLAN"] = "0"
if mode == "decide":
    x = torch.relu(torch.randn(B, H, H, C, device="cuda"))
    w = torch.randn(2, C, device="cuda") * 0.1; b = torch.zeros(2, device="cuda")
    _, _, work = ops.spatial_masker(x, w, b, 1, S, return_work=True)
    fn = lambda: ops.mask_plan(work.view(B, S, S, C), w, b, H, H, 1, patch_major=True)
elif mode == "pm":
    fn = lambda: ops.mask_to_index(patch, H, H, 1, patch_major=True)
else:
    fn = lambda: ops.mask_to_index(patch, H, H, 1)
for _ in range(30):
    fn()
torch.cuda.synchronize()
