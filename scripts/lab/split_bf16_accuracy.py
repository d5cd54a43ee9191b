"""GPU lab: how accurate would an fp32 GEMM assembled from bf16 MFMA terms be?  a = a0 + a1 + a2 (three bf16 pieces, 24
mantissa bits), 6 cross terms a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0 (the dropped ones are <= 2^-24 relative), every product
exact in fp32, fp32 accumulation -- against this library's exact-fp32 MFMA GEMM, both vs float64.  (Products are evaluated
here by fp32 matmuls of the bf16-valued pieces: the pieces' products are exact either way, so the split's truncation error is
what shows.)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, N, K = 8192, 3072, 768
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5
ref = x.double() @ w.double().t()
y = torch.empty(M, N, device=dev)
ops.gemm_f32_raw("nt", x, w, y)
def split(t):
    p0 = t.bfloat16().float(); r = t - p0
    p1 = r.bfloat16().float(); r = r - p1
    return p0, p1, r.bfloat16().float()
a, b = split(x), split(w)
terms = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
acc6 = torch.zeros(M, N, device=dev)
for i, j in sorted(terms, key=lambda ij: -(ij[0] + ij[1])):          # small terms first
    acc6 += a[i] @ b[j].t()
acc3 = a[0] @ b[0].t() + a[0] @ b[1].t() + a[1] @ b[0].t()
scale = ref.abs().max()
for name, got in (("exact-fp32 MFMA (acr_gemm_f32)", y), ("torch fp32 (hipBLASLt)", x @ w.t()), ("bf16 split, 6 terms", acc6), ("bf16 split, 3 terms", acc3),
                  ("plain bf16 inputs", a[0] @ b[0].t())):
    e = (got.double() - ref).abs()
    print("%-32s max |err| / max|y| %.3e   rms err / rms y %.3e" % (name, float(e.max() / scale), float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())))
