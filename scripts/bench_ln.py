import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acr_wsss_amd import ops
import torch.nn.functional as F
dev = "cuda:0"
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
M, C = 25120, 768
x = torch.randn(M, C, device=dev).bfloat16().requires_grad_(True)
ln = torch.nn.LayerNorm(C, eps=1e-6).to(dev).bfloat16()
dy = torch.randn(M, C, device=dev).bfloat16()
def run(hip):
    y = ops.layer_norm(x, ln, hip)
    y.backward(dy)
    x.grad = None; ln.weight.grad = None; ln.bias.grad = None
def fwd(hip):
    with torch.no_grad():
        ops.layer_norm(x, ln, hip)
print("fwd   hip %.1f us   torch %.1f us" % (t(lambda: fwd(True)), t(lambda: fwd(False))))
print("f+b   hip %.1f us   torch %.1f us" % (t(lambda: run(True)), t(lambda: run(False))))
