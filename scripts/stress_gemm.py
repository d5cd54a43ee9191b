"""Repeated correctness check of acr_linear_bf16 (ACR_GEMM_VARIANT selects the kernel) against fp32 matmul."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acr_wsss_amd import ops
dev = "cuda:0"
M = 32 * 785
for (N, K) in ((768, 768), (2304, 768), (768, 3072), (3072, 768), (768, 2304)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).to(dev).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16()
    b = torch.randn(N, generator=g).to(dev).bfloat16()
    ref = torch.nn.functional.linear(x.float(), w.float(), b.float())
    bad = 0
    for it in range(30):
        y = ops.linear_bf16(x, w, b)
        d = (y.float() - ref).abs()
        err = d.max().item() / ref.abs().max().item()
        if err > 1e-2:
            bad += 1
            if bad <= 3:
                idx = (d > 0.05 * ref.abs().max()).nonzero()
                if idx.shape[0] == 0: continue
                rows = idx[:, 0].unique(); cols = idx[:, 1].unique()
                print("  N=%d K=%d it=%d err=%.3e nbad=%d rows[%d..%d] (%d distinct) cols[%d..%d] (%d distinct)" % (
                    N, K, it, err, idx.shape[0], rows.min().item(), rows.max().item(), rows.numel(), cols.min().item(), cols.max().item(), cols.numel()))
    print("N=%d K=%d: %d / 30 bad" % (N, K, bad), flush=True)
