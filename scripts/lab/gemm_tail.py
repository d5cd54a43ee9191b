"""GPU lab: does the partly filled last round of 128x128 tiles cost the fp32 GEMM?  NT, N = 768, K = 3072, M swept so that the
tile count crosses multiples of the 512 resident workgroups."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import ops
dev = torch.device("cuda:0")
def t(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it
for N, K in ((768, 3072), (3072, 768)):
    w = torch.randn(N, K, device=dev) * K ** -0.5
    for rows in (128, 160, 170, 171, 180, 197, 214, 256):
        M = rows * 128 * (6 if N == 768 else 1) // (6 if N == 768 else 1)
        x = torch.randn(M, K, device=dev)
        y = torch.empty(M, N, device=dev)
        ms = t(lambda: ops.gemm_f32_raw("nt", x, w, y))
        tiles = rows * (N // 128)
        print("N %4d K %4d M %6d: %5d tiles = %.2f rounds of 512  %.3f ms  %.1f TF  (%.4f ms per round-equivalent)" % (
            N, K, M, tiles, tiles / 512, ms, 2.0 * M * N * K / ms / 1e9, ms / (tiles / 512)))
