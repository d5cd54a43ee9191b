"""Runs acr_linear_bf16 on one shape a few times (for rocprofv3 --pmc passes). usage: one_gemm.py N K [reps]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acr_wsss_amd import ops
N, K = int(sys.argv[1]), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
M = 32 * 785
x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
b = torch.randn(N, device="cuda").bfloat16()
for _ in range(reps): y = ops.linear_bf16(x, w, b)
torch.cuda.synchronize()
