"""GPU lab: which pipe bounds the bf16x3-split GEMM?  Time of the fc2-shaped NT product (25120 x 768 x 3072) with the library as
built, with the split arithmetic removed (loads + MFMAs only) and with the MFMAs removed (loads + split only); lab builds
SPLIT_LAB=1 / 2 give wrong results on purpose.  usage: ACR_LIB_PATH=... gemm_split_parts.py"""
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
M = 25120
for N, K in ((768, 3072), (3072, 768)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; y = torch.empty(M, N, device=dev)
    ms = t(lambda: ops.gemm_f32_raw("nt", x, w, y, math=1))
    print("%s  N %d K %d: %.3f ms" % (os.path.basename(os.environ.get("ACR_LIB_PATH", "libacr_hip.so")), N, K, ms), flush=True)
