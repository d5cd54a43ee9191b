"""Runs one split-product (or exact) fp32 GEMM shape a few times, for rocprofv3 --pmc passes.
usage: one_split_gemm.py mode N K [math] [reps]      mode: nt | tn   (M = 32 * 785 tokens)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import ops
mode, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
math = int(sys.argv[4]) if len(sys.argv) > 4 else 1
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
M = 32 * 785
dev = torch.device("cuda:0")
if mode == "nt":
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev); y = torch.empty(M, N, device=dev)
    fn = lambda: ops.gemm_f32_raw("nt", x, w, y, bias=b, math=math)
else:
    dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev); dw = torch.empty(N, K, device=dev); cs = torch.empty(N, device=dev)
    fn = lambda: ops.gemm_f32_raw("tn", dy, x, dw, colsum=cs, math=math)
for _ in range(reps):
    fn()
torch.cuda.synchronize()
