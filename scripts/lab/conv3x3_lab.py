"""GPU lab: the stem's 3x3 convolutions with split products (csrc/conv3x3.hip) at the BASELINE step's three shapes -- forward,
input gradient, weight gradient per launch (HIP events), beside the split-product 1x1 GEMM of the same M, N, K (cin = 9 C: what
the implicit GEMM would cost with no taps) and the library's fp32 convolution.  usage: conv3x3_lab.py [reps]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import _lib as L
if os.environ.get("ACR_LAB_LIB"):
    L.LIB_PATH = os.environ["ACR_LAB_LIB"]
from acr_wsss_amd import ops
import torch.nn.functional as F
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
only = os.environ.get("C3_ONLY", "")


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


lib = L.load()
for (N, C, H) in ((32, 64, 112), (32, 128, 56), (32, 256, 28)):
    x = torch.empty(N, C, H, H, device=dev); x.normal_()
    dy = torch.empty_like(x); dy.normal_()
    w = torch.randn(C, C, 3, 3, device=dev) * (9 * C) ** -0.5
    wp = w.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous()
    y = torch.empty_like(x)
    ws = torch.empty(lib.acr_conv3x3_wgrad_ws_floats(N, C, C, H, H), device=dev)
    dwp = torch.empty(C, 9 * C, device=dev)
    gf = 2.0 * N * H * H * C * C * 9 * 1e-9
    out = []
    if only in ("", "fwd"):
        t = timed(lambda: L.check(lib.acr_conv3x3_f32(1, L.ptr(wp), L.ptr(x), L.ptr(y), N, C, C, H, H, None, L.stream_ptr()), "c3"))
        out.append("fwd %.1f us (%.0f TF-eq)" % (t, gf / t * 1e3))
    if only in ("", "wgrad"):
        t = timed(lambda: L.check(lib.acr_conv3x3_wgrad_f32(1, L.ptr(dy), L.ptr(x), N, C, C, H, H, L.ptr(ws), L.ptr(dwp), L.stream_ptr()), "c3w"))
        out.append("wgrad %.1f us (%.0f TF-eq)" % (t, gf / t * 1e3))
    if only == "":
        x9 = torch.randn(N, 9 * C, H, H, device=dev) if N * 9 * C * H * H < 2 ** 31 else None
        if x9 is not None:
            w1 = torch.randn(C, 9 * C, 1, 1, device=dev)
            t = timed(lambda: ops.conv1x1(x9, w1, None, 1))
            out.append("1x1 split GEMM of the same M,N,K %.1f us" % t)
            del x9
        t = timed(lambda: F.conv2d(x, w, padding=1))
        out.append("library fwd %.1f us" % t)
    print("N %d C %d %dx%d (%.1f GF): %s" % (N, C, H, H, gf, "   ".join(out)), flush=True)
