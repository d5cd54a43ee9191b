"""GPU lab: time of the image products (acr_gemm_x3 NT / TN) at the block shapes; with ACR_LAB_LIB a lab build of the library."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import _lib
if os.environ.get("ACR_LAB_LIB"):
    _lib.LIB_PATH = os.environ["ACR_LAB_LIB"]
from acr_wsss_amd import ops
dev = torch.device("cuda:0")
def t(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it
M = 25120
for name, N, K in (("qkv", 2304, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; dy = torch.randn(M, N, device=dev)
    xi, wi, dyi = ops.x3_image(x), ops.x3_image(w), ops.x3_image(dy)
    y = torch.empty(M, N, device=dev); dw = torch.empty(N, K, device=dev)
    tn = t(lambda: ops.gemm_x3("nt", xi, wi, y, K))
    tt = t(lambda: ops.gemm_x3("tn", dyi, xi, dw, M))
    fl = 2.0 * M * N * K
    print("%-4s N %4d K %4d: NT %.3f ms (%.0f TF-eq)   TN %.3f ms (%.0f TF-eq)" % (name, N, K, tn, fl / tn / 1e9, tt, fl / tt / 1e9), flush=True)
