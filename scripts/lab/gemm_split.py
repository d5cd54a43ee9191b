"""GPU lab: acr_gemm_f32 with every product as six bf16-MFMA terms of a three-way operand split (math = ACR_MATH_BF16X3) vs the
exact-fp32 MFMA kernels: time at the bench's block-GEMM shapes (NT with epilogues, TN weight gradient) and error vs float64."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import ops, _lib
dev = torch.device("cuda:0")
def t(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it
torch.manual_seed(0)
# accuracy on a problem float64 can check quickly
M, N, K = 4096, 768, 3072
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; dy = torch.randn(M, N, device=dev)
ref = x.double() @ w.double().t(); refw = dy.double().t() @ x.double()
for split in (0, 1):
    y = torch.empty(M, N, device=dev); dw = torch.empty(N, K, device=dev)
    ops.gemm_f32_raw("nt", x, w, y, math=split); ops.gemm_f32_raw("tn", dy, x, dw, math=split)
    e1 = (y.double() - ref).abs(); e2 = (dw.double() - refw).abs()
    print("split %d: NT max err / max|y| %.3e rms %.3e   TN max %.3e rms %.3e" % (split, float(e1.max() / ref.abs().max()),
          float(e1.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), float(e2.max() / refw.abs().max()), float(e2.pow(2).mean().sqrt() / refw.pow(2).mean().sqrt())), flush=True)
M = 25120
for name, mode, N, K, act in (("qkv", "nt", 2304, 768, 0), ("proj", "nt", 768, 768, 0), ("fc1+gelu", "nt", 3072, 768, 1), ("fc2", "nt", 768, 3072, 0),
                              ("fc2 dx*gelu'", "nt", 3072, 768, 2), ("fc1 dW", "tn", 3072, 768, 0), ("fc2 dW", "tn", 768, 3072, 0), ("qkv dW", "tn", 2304, 768, 0)):
    res = {}
    for split in (0, 1):
        if mode == "nt":
            x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
            aux = torch.randn(M, N, device=dev) if act != 1 else None
            y = torch.empty(M, N, device=dev); y2 = torch.empty(M, N, device=dev) if act == 1 else None
            fn = lambda: ops.gemm_f32_raw("nt", x, w, y, bias=None if act == 2 else b, aux=aux, act=act, c2=y2, math=split)
        else:
            dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev); dw = torch.empty(N, K, device=dev); cs = torch.empty(N, device=dev)
            fn = lambda: ops.gemm_f32_raw("tn", dy, x, dw, colsum=cs, math=split)
        res[split] = t(fn)
    fl = 2.0 * M * N * K
    print("%-14s %s N %4d K %4d: exact %.3f ms (%.0f TF)   split %.3f ms (%.0f TF-equivalent)   x%.2f" % (name, mode, N, K, res[0], fl / res[0] / 1e9, res[1], fl / res[1] / 1e9, res[0] / res[1]), flush=True)
