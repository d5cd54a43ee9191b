"""fc1 + GELU forward and fc2 input gradient through GELU': fused epilogues vs GEMM + separate elementwise kernels."""
import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acr_wsss_amd import ops, _lib as L
lib = L.load(); dev = "cuda:0"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
M, D, Hd = 32 * 785, 768, 3072
x = torch.randn(M, D, device=dev).bfloat16(); w1 = (torch.randn(Hd, D, device=dev) * D ** -0.5).bfloat16(); b1 = torch.randn(Hd, device=dev).bfloat16()
h = torch.empty(M, Hd, device=dev, dtype=torch.bfloat16); a = torch.empty_like(h)
def fused_fwd(): L.check(lib.acr_linear_gelu_bf16(L.ptr(x), D, L.ptr(w1), D, L.ptr(b1), L.ptr(h), L.ptr(a), Hd, M, Hd, D, L.stream_ptr()), "f")
def plain_fwd(): return F.gelu(ops.linear_bf16(x, w1, b1))
def gemm_only(): return ops.linear_bf16(x, w1, b1)
dy = torch.randn(M, D, device=dev).bfloat16(); w2t = (torch.randn(Hd, D, device=dev) * D ** -0.5).bfloat16(); dh = torch.empty_like(h)
fused_fwd()
def fused_bwd(): L.check(lib.acr_linear_dgelu_bf16(L.ptr(dy), D, L.ptr(w2t), D, L.ptr(h), Hd, L.ptr(dh), Hd, M, Hd, D, L.stream_ptr()), "b")
def plain_bwd(): return torch.ops.aten.gelu_backward(ops.linear_bf16(dy, w2t), h)
print("fc1 fwd : GEMM only %.1f us | GEMM + gelu kernel %.1f us | fused epilogue %.1f us" % (t(gemm_only), t(plain_fwd), t(fused_fwd)))
print("fc2 dX  : GEMM + gelu_backward kernel %.1f us | fused epilogue %.1f us" % (t(plain_bwd), t(fused_bwd)))
