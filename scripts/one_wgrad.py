"""Runs acr_wgrad_bf16 on one shape a few times (for rocprofv3 --pmc passes). usage: one_wgrad.py N K [reps]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acr_wsss_amd import ops
N, K = int(sys.argv[1]), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
M = 32 * 785
dy = torch.randn(M, N, device="cuda").bfloat16(); x = torch.randn(M, K, device="cuda").bfloat16()
for _ in range(reps): dw = ops.wgrad_bf16(dy, x)
torch.cuda.synchronize()
