"""Repeated correctness check of acr_wgrad_bf16 against fp32 matmul."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acr_wsss_amd import ops
dev = "cuda:0"
for M in (32 * 785, 4096 + 32):
    for (N, K) in ((768, 768), (2304, 768), (768, 3072), (3072, 768)):
        g = torch.Generator().manual_seed(0)
        dy = torch.randn(M, N, generator=g).to(dev).bfloat16()
        x = torch.randn(M, K, generator=g).to(dev).bfloat16()
        ref = dy.float().t() @ x.float()
        bad = 0
        for it in range(20):
            dw = ops.wgrad_bf16(dy, x)
            err = (dw.float() - ref).abs().max().item() / ref.abs().max().item()
            bad += err > 1e-2
        print("M=%d N=%d K=%d: %d / 20 bad (last err %.2e)" % (M, N, K, bad, err), flush=True)
