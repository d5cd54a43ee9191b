"""GPU lab: the stem's convolutions under split products at the step's shapes (32 views of 448^2): forward, input gradient and weight
gradient per distinct shape, HIP events around the C-ABI launches only (no autograd / allocator in the timed region).
usage: conv_stem_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import _lib as L
if os.environ.get("ACR_LAB_LIB"):
    L.LIB_PATH = os.environ["ACR_LAB_LIB"]
from acr_wsss_amd import ops
dev = torch.device("cuda:0")
lib = L.load()
N = 32

def t(fn, n=8):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

# (cin, cout, S, count): stride-1 1x1 convolutions of the three stages (+ the subsampled shortcuts of stages 1, 2 and the patch projection)
one = [(64, 64, 112, 1), (64, 256, 112, 4), (256, 64, 112, 2), (256, 128, 112, 1), (128, 512, 56, 4), (256, 512, 56, 1), (512, 128, 56, 3),
       (512, 256, 56, 1), (256, 1024, 28, 9), (512, 1024, 28, 1), (1024, 256, 28, 8), (1024, 768, 28, 1)]
tot = [0.0, 0.0, 0.0]
print("1x1:  cin cout   S  cnt    fwd us  TF-eq    dX us  TF-eq    dW us  TF-eq")
for ci, co, S, cnt in one:
    hw = S * S
    x = torch.randn(N, ci, S, S, device=dev)
    w = torch.randn(co, ci, device=dev) * ci ** -0.5
    dy = torch.randn(N, co, S, S, device=dev)
    y = torch.empty(N, co, S, S, device=dev)
    dx = torch.empty_like(x)
    dw = torch.empty(co, ci, device=dev)
    wsd = torch.empty(lib.acr_conv1x1_wgrad_f32_ws_floats(N, co, ci, hw), device=dev)
    f = t(lambda: ops._conv1x1_f32_launch(1, w, 0, x, None, y, N, co, ci, hw))
    b = t(lambda: ops._conv1x1_f32_launch(1, w, 1, dy, None, dx, N, ci, co, hw))
    g = t(lambda: L.check(lib.acr_conv1x1_wgrad_f32(1, L.ptr(dy), L.ptr(x), N, co, ci, hw, L.ptr(wsd), L.ptr(dw), L.stream_ptr()), "wg"))
    fl = 2.0 * N * hw * ci * co
    print("     %4d %4d %3d  x%2d  %8.1f %6.1f %8.1f %6.1f %8.1f %6.1f" % (ci, co, S, cnt, f, fl / f / 1e6, b, fl / b / 1e6, g, fl / g / 1e6), flush=True)
    tot[0] += cnt * f; tot[1] += cnt * b; tot[2] += cnt * g
print("1x1 per step: fwd %.2f ms  dX %.2f ms  dW %.2f ms" % tuple(v / 1e3 for v in tot))
three = [(64, 112, 3), (128, 56, 3), (256, 28, 8)]
tot = [0.0, 0.0, 0.0]
print("3x3:    c   S  cnt    fwd us  TF-eq    dX us  TF-eq    dW us  TF-eq")
for c, S, cnt in three:
    x = torch.empty(N, c, S, S, device=dev); x.normal_()
    dy = torch.empty(N, c, S, S, device=dev); dy.normal_()
    w = torch.randn(c, c, 3, 3, device=dev) * (9 * c) ** -0.5
    wp = w.permute(0, 2, 3, 1).reshape(c, 9 * c).contiguous()
    wd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(c, 9 * c).contiguous()
    y = torch.empty(N, c, S, S, device=dev)
    ws = torch.empty(lib.acr_conv3x3_wgrad_ws_floats(N, c, c, S, S), device=dev)
    dwp = torch.empty(c, 3, 3, c, device=dev)
    ops.CONV3X3_WIMG = False
    f0 = t(lambda: ops._conv3x3_launch(wp, x, y, N, c, c, S, S))
    ops.CONV3X3_WIMG = True
    print("     (both operands split in registers: fwd %.1f us)" % f0)
    f = t(lambda: ops._conv3x3_launch(wp, x, y, N, c, c, S, S))
    b = t(lambda: ops._conv3x3_launch(wd, dy, y, N, c, c, S, S))
    g = t(lambda: L.check(lib.acr_conv3x3_wgrad_f32(1, L.ptr(dy), L.ptr(x), N, c, c, S, S, L.ptr(ws), L.ptr(dwp), L.stream_ptr()), "wg3"))
    fl = 2.0 * 9 * N * S * S * c * c
    print("     %4d %3d  x%2d  %8.1f %6.1f %8.1f %6.1f %8.1f %6.1f" % (c, S, cnt, f, fl / f / 1e6, b, fl / b / 1e6, g, fl / g / 1e6), flush=True)
    tot[0] += cnt * f; tot[1] += cnt * b; tot[2] += cnt * g
print("3x3 per step: fwd %.2f ms  dX %.2f ms  dW %.2f ms" % tuple(v / 1e3 for v in tot))
