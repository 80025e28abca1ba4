"""Graph replay of the spatial model vs eager (fused spatial masker on)."""
import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import laudnet_amd
from laudnet_amd import ops
from laudnet_amd.laud_resnet import GraphedForward
from fill import fill_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
big = len(sys.argv) > 2
ops.set_math_mode("bf16x3")
kw = dict(dyn_mode=["spatial"] * 4, mask_spatial_granularity=[4, 4, 2, 1])
m = (laudnet_amd.uni_resnet101(**kw) if big else laudnet_amd.uni_resnet50(width_mult=0.5, input_size=224, num_classes=10, **kw)).eval()
sd = fill_state_dict(m.state_dict(), 3)
for k in sd:
    if k.endswith("masker_spatial.conv.bias"):
        sd[k] = torch.zeros_like(sd[k])
m.load_state_dict(sd)
m = m.to("cuda:0")
x = torch.randn(B, 3, 224, 224, device="cuda:0")
with torch.no_grad():
    y0 = m(x, 1.0)[0].clone()
torch.cuda.synchronize()
print("eager ok", float(y0.abs().max()), flush=True)
g = GraphedForward(m, x, 1.0)
print("captured", flush=True)
for i in range(3):
    y1 = g(x)[0]
    torch.cuda.synchronize()
    print("replay", i, bool(torch.equal(y1, y0)), float((y1 - y0).abs().max()), flush=True)
