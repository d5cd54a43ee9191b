"""Times the NCHW 1x1-conv GEMM kernels against MIOpen for the stem's stride-1 1x1 shapes (bf16, 32 samples)."""
import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acr_wsss_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3

shapes = [(64, 256, 112, 4), (64, 64, 112, 1), (256, 64, 112, 2), (256, 128, 112, 1), (128, 512, 56, 4), (512, 128, 56, 3),
          (512, 256, 56, 1), (256, 1024, 28, 9), (1024, 256, 28, 8)]
N = 32
tot = [0, 0, 0, 0]
for cin, cout, hw, cnt in shapes:
    x = torch.randn(N, cin, hw, hw, device="cuda").bfloat16().requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 1, device="cuda") * cin ** -0.5).bfloat16().requires_grad_(True)
    dy = torch.randn(N, cout, hw, hw, device="cuda").bfloat16()
    def fwd_m(): return F.conv2d(x, w)
    def fwd_h(): return ops.conv1x1(x, w)
    ym, yh = fwd_m(), fwd_h()
    def bwd_m(): torch.autograd.grad(ym, (x, w), dy, retain_graph=True)
    def bwd_h(): torch.autograd.grad(yh, (x, w), dy, retain_graph=True)
    a, b, c, d = t(fwd_m), t(fwd_h), t(bwd_m), t(bwd_h)
    mb = (x.numel() + ym.numel()) * 2 / 1e6
    print(f"{cin:5d}->{cout:5d} @{hw:3d}^2 x{cnt}: fwd miopen {a:7.1f} hip {b:7.1f} us ({mb / b * 1e3 / 1e3:5.2f} TB/s)   bwd miopen {c:7.1f} hip {d:7.1f} us", flush=True)
    tot[0] += a * cnt; tot[1] += b * cnt; tot[2] += c * cnt; tot[3] += d * cnt
print("per step: fwd miopen %.2f ms hip %.2f ms; bwd miopen %.2f ms hip %.2f ms" % tuple(v / 1e3 for v in tot))
