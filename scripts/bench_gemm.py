"""A/B: hand-written bf16 MFMA GEMM (acr_linear_bf16) vs torch F.linear (hipBLASLt) on the block's shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acr_wsss_amd import ops
dev = "cuda:0"
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
M = 32 * 785
for (N, K, name) in ((2304, 768, "qkv fwd"), (768, 768, "proj fwd / dx"), (768, 2304, "qkv dx"), (3072, 768, "fc1 fwd"), (768, 3072, "fc2 fwd")):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).to(dev).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16()
    b = torch.randn(N, generator=g).to(dev).bfloat16()
    y = ops.linear_bf16(x, w, b)
    ref = torch.nn.functional.linear(x.float(), w.float(), b.float())
    err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
    t_mine = t(lambda: ops.linear_bf16(x, w, b))
    t_torch = t(lambda: torch.nn.functional.linear(x, w, b))
    fl = 2.0 * M * N * K
    print("%-14s M=%d N=%4d K=%4d  mine %7.1f us (%6.1f TF)  hipBLASLt %7.1f us (%6.1f TF)  relerr %.2e" % (
        name, M, N, K, t_mine, fl / t_mine / 1e6, t_torch, fl / t_torch / 1e6, err))

print("--- weight gradients dW = dY^T X (M = %d) ---" % M)
for (N, K, name) in ((2304, 768, "qkv dW"), (768, 768, "proj dW"), (3072, 768, "fc1 dW"), (768, 3072, "fc2 dW")):
    g = torch.Generator().manual_seed(1)
    dy = torch.randn(M, N, generator=g).to(dev).bfloat16()
    x = torch.randn(M, K, generator=g).to(dev).bfloat16()
    dw = ops.wgrad_bf16(dy, x)
    ref = dy.float().t() @ x.float()
    err = (dw.float() - ref).abs().max().item() / ref.abs().max().item()
    t_mine = t(lambda: ops.wgrad_bf16(dy, x))
    t_torch = t(lambda: torch.mm(dy.t(), x))
    fl = 2.0 * M * N * K
    print("%-14s N=%4d K=%4d  mine %7.1f us (%6.1f TF)  hipBLASLt %7.1f us (%6.1f TF)  relerr %.2e" % (
        name, N, K, t_mine, fl / t_mine / 1e6, t_torch, fl / t_torch / 1e6, err))
