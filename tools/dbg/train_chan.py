import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
from helpers import block_input, load_golden, make_block
from laudnet_amd import ops
from laudnet_amd.laud_resnet import Bottleneck
import laudnet_amd.training as T
fx = load_golden("blocks_s1.pt")["channel_g2_s1"]
print(fx["kw"], fx["x_shape"])
res = {}
orig = ops.conv_rows
for mode in ("fp32", "bf16x3"):
    ops.set_math_mode(mode)
    log = []
    def spy(*a, **kw):
        r = orig(*a, **kw)
        torch.cuda.synchronize()
        log.append((a[4].detach().clone(), {k: (v if not torch.is_tensor(v) else tuple(v.shape)) for k, v in kw.items()}, tuple(a[1].shape)))
        return r
    ops.conv_rows = spy
    blk = make_block(Bottleneck, fx).cuda()
    x = block_input(fx).cuda().requires_grad_(True)
    m = fx["channel_mask"].float().cuda().requires_grad_(True)
    for p in blk.parameters(): p.requires_grad_(True)
    out = T.sparse_block_train(blk, x, m)
    out.backward(torch.ones_like(out))
    res[mode] = log
    ops.conv_rows = orig
for i, (a, b) in enumerate(zip(res["fp32"], res["bf16x3"])):
    print(i, a[2], {k: v for k, v in a[1].items() if k in ("taps", "relu", "m_cap")}, "maxdiff", (a[0] - b[0]).abs().max().item(), "scale", a[0].abs().max().item())
