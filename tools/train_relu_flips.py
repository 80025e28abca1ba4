# tuning / evidence only: backs the gradient criterion of tests/test_hip_training.py (_close) -- counts the ReLU decisions of the bf16x3 forward that
# differ from an fp64 evaluation of the same block (pre-activations within rounding of zero) and shows the gradient difference is confined to them.
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
from helpers import block_input, load_golden, make_block
from fill import seeded_randn
from laudnet_amd import ops
from laudnet_amd.laud_resnet import Bottleneck
import laudnet_amd.training as T
from oracle import torch_ref as TR
fx = load_golden("blocks_s1.pt")["channel_g2_s1"]
ops.set_math_mode("bf16x3")
hip = make_block(Bottleneck, fx).cuda(); ref = make_block(TR.BottleneckRef, fx).cuda().double()
x0 = block_input(fx).cuda(); m0 = fx["channel_mask"].float().cuda()
xr = x0.double().clone().requires_grad_(True); mr = m0.double().clone().requires_grad_(True)
ref.forced_channel_mask = mr
# record oracle pre-activations
pre = {}
for n in ("bn1", "bn2"):
    getattr(ref, n).register_forward_hook(lambda mod, i, o, n=n: pre.__setitem__(n, o.detach()))
out_r = ref((xr, None, None, None, None, None, torch.tensor(0.0, device="cuda")), 1.0)[0]
g = seeded_randn(tuple(out_r.shape), 77).cuda()
out_r.backward(g.double())
xh = x0.clone().requires_grad_(True); mh = m0.clone().requires_grad_(True)
out_h = T.sparse_block_train(hip, xh, mh); out_h.backward(g)
d = (xh.grad.double() - xr.grad).abs()
print("dx: max", d.max().item(), "elements > 1e-3:", int((d > 1e-3).sum()), "of", d.numel())
for n, z in pre.items():
    print(n, "min |pre-activation|", z.abs().min().item(), "count < 1e-4:", int((z.abs() < 1e-4).sum()))
print("final pre-act min:", (out_r.detach()[out_r > 0]).min().item())
bad = (d > 1e-3).nonzero()
print(bad[:10])
