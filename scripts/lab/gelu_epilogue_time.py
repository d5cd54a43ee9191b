"""GPU lab: the fc1-shaped image product with the plain epilogue (act 0) and with the GELU image epilogue (C-ABI act 3 = kernel <5>: c = GELU'(h) in fp32,
c2 = image of GELU(h)); with ACR_LAB_LIB a lab build of the library.  usage: gelu_epilogue_time.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import _lib
if os.environ.get("ACR_LAB_LIB"):
    _lib.LIB_PATH = os.environ["ACR_LAB_LIB"]
from acr_wsss_amd import ops
dev = torch.device("cuda:0")
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
M, N, K = 25120, 3072, 768
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev) * 0.1
xi, wi = ops.x3_image(x), ops.x3_image(w)
y = torch.empty(M, N, device=dev); gi = ops.x3_image_empty(M, N, dev)
t0 = t(lambda: ops.gemm_x3("nt", xi, wi, y, K, bias=b))
t5 = t(lambda: ops.gemm_x3("nt", xi, wi, y, K, bias=b, act=3, c2=gi))
print("fc1 25120 x 3072 x 768: plain epilogue %.1f us, GELU image epilogue %.1f us" % (t0, t5))
