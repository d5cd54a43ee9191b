"""GPU lab: fp32 GroupNorm forward / backward at the step's shapes (32 views of 448^2), HIP events, algorithmic bytes / time.
usage: gn_time.py   (ACR_LAB_LIB=path of a lab build of the library for A/B)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import _lib
if os.environ.get("ACR_LAB_LIB"):
    _lib.LIB_PATH = os.environ["ACR_LAB_LIB"]
from acr_wsss_amd import ops
dev = torch.device("cuda:0")
N = 32
# (C, S, act, count per step): norm1/norm2 (relu) and norm3 (+ residual) of the three stages, the stem norm
shapes = [(64, 224, "relu", 1), (64, 112, "relu", 5), (256, 112, "add_relu", 3), (256, 112, "none", 1), (128, 112, "relu", 1), (128, 56, "relu", 7),
          (512, 56, "add_relu", 4), (512, 56, "none", 1), (256, 56, "relu", 1), (256, 28, "relu", 17), (1024, 28, "add_relu", 9), (1024, 28, "none", 1)]
tf = tb = 0.0
for C, S, act, cnt in shapes:
    x = torch.randn(N, C, S, S, device=dev).requires_grad_(True)
    r = torch.randn(N, C, S, S, device=dev).requires_grad_(True) if act == "add_relu" else None
    w, b = torch.ones(C, device=dev).requires_grad_(True), torch.zeros(C, device=dev).requires_grad_(True)
    dy = torch.randn(N, C, S, S, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    f = bw = 0.0
    for it in range(6):
        x.grad = None
        ev[0].record()
        y = ops.groupnorm_act(x, w, b, act, r)
        ev[1].record()
        y.backward(dy)
        ev[2].record()
        torch.cuda.synchronize()
        if it:
            f += ev[0].elapsed_time(ev[1]) / 5; bw += ev[1].elapsed_time(ev[2]) / 5
    nb = x.numel() * 4
    fb = nb * (3 if r is None else 4)            # 2 reads (+1) + 1 write
    bb = nb * (5 if r is None else 8)            # 4 reads (+2) + 1 write (+1)
    print("C %4d %3dx%-3d %-8s x%2d  fwd %7.1f us %5.2f TB/s   bwd %7.1f us %5.2f TB/s" % (C, S, S, act, cnt, f * 1e3, fb / f / 1e9, bw * 1e3, bb / bw / 1e9), flush=True)
    tf += cnt * f; tb += cnt * bw
    del x, r, y, dy
print("per step: fwd %.2f ms  bwd %.2f ms" % (tf, tb))
