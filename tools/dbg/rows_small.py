import sys, torch
sys.path.insert(0, ".")
from laudnet_amd import ops
torch.manual_seed(0)
for mode in ("fp32", "bf16x3"):
    ops.set_math_mode(mode)
    for (rows, cin, cout) in ((588, 64, 16), (588, 16, 64), (600, 64, 16), (588, 64, 32), (588, 32, 16)):
        a = torch.randn(rows, cin, device="cuda")
        w = torch.randn(cout, 1, cin, device="cuda")
        z = torch.zeros(cout, device="cuda")
        want = (a.double() @ w.reshape(cout, cin).double().t()).float()
        for init in ("empty", "zeros", "nan"):
            out = {"empty": torch.empty, "zeros": torch.zeros}.get(init, lambda *s, **k: torch.full(s, float("nan"), **k))(rows, cout, device="cuda")
            ops.conv_rows(a, w, None, z, out, taps=1, m_cap=rows, relu=0)
            torch.cuda.synchronize()
            print(mode, rows, cin, cout, init, "maxdiff", (out - want).abs().max().item())
