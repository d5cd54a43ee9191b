"""GPU lab: only the 1x1 weight gradients of the stem at the step's shapes, 6 launches each -- run under rocprofv3 --kernel-trace and
summarise with kstats_grid.py (main kernel vs slab sum per shape).  usage: conv_wgrad_trace.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import _lib as L
if os.environ.get("ACR_LAB_LIB"):
    L.LIB_PATH = os.environ["ACR_LAB_LIB"]
from acr_wsss_amd import ops
dev = torch.device("cuda:0")
lib = L.load()
N = 32
one = [(64, 64, 112, 1), (64, 256, 112, 4), (256, 64, 112, 2), (256, 128, 112, 1), (128, 512, 56, 4), (256, 512, 56, 1), (512, 128, 56, 3),
       (512, 256, 56, 1), (256, 1024, 28, 9), (512, 1024, 28, 1), (1024, 256, 28, 8), (1024, 768, 28, 1)]
for ci, co, S, cnt in one:
    hw = S * S
    x = torch.randn(N, ci, S, S, device=dev)
    dy = torch.randn(N, co, S, S, device=dev)
    dw = torch.empty(co, ci, device=dev)
    wsd = torch.empty(lib.acr_conv1x1_wgrad_f32_ws_floats(N, co, ci, hw), device=dev)
    for _ in range(6):
        L.check(lib.acr_conv1x1_wgrad_f32(1, L.ptr(dy), L.ptr(x), N, co, ci, hw, L.ptr(wsd), L.ptr(dw), L.stream_ptr()), "wg")
    torch.cuda.synchronize()
    print(ci, co, S, "ws floats", wsd.numel(), flush=True)
