"""GPU lab: the stem's SAME max-pool at the step's shape (32 x 64 x 224 x 224 fp32), forward and backward, HIP events; bytes = input +
output + argmax bytes.  usage: pool_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(32, 64, 224, 224, device=dev).requires_grad_(True)
dy = torch.randn(32, 64, 112, 112, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
f = b = 0.0
for it in range(8):
    x.grad = None
    ev[0].record()
    y = ops.maxpool3x3s2_same(x, 0, 0, 1, 1)
    ev[1].record()
    y.backward(dy)
    ev[2].record()
    torch.cuda.synchronize()
    if it >= 2:
        f += ev[0].elapsed_time(ev[1]) / 6; b += ev[1].elapsed_time(ev[2]) / 6
nb = x.numel() * 4 + y.numel() * 5
print("maxpool 32x64x224x224: fwd %.1f us (%.2f TB/s)  bwd incl. autograd %.1f us (%.2f TB/s)" % (f * 1e3, nb / f / 1e9, b * 1e3, nb / b / 1e9))
