"""GPU lab: fp32 GroupNorm through the C ABI directly (no autograd, no per-call allocation), 20 launches per timing -- the kernels'
own time at the step's shapes under the workgroup-size plans of ACR_OPT_GN_PLAN (0 / 1 / 2).  usage: gn_raw_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
N = 32
shapes = [(64, 224, 1, 1), (64, 112, 1, 5), (256, 112, 2, 3), (256, 112, 0, 1), (128, 112, 1, 1), (128, 56, 1, 7),
          (512, 56, 2, 4), (512, 56, 0, 1), (256, 56, 1, 1), (256, 28, 1, 17), (1024, 28, 2, 9), (1024, 28, 0, 1)]
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
tot = {}
for C, S, act, cnt in shapes:
    x = torch.randn(N, C, S, S, device=dev); r = torch.randn(N, C, S, S, device=dev); dy = torch.randn(N, C, S, S, device=dev)
    w, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    y, dx, dr = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    stats = torch.empty(N * 64, device=dev); mask = torch.empty(x.numel() // 4, dtype=torch.uint8, device=dev)
    part = torch.empty(2, N, C, device=dev); dgb = torch.empty(2, C, device=dev)
    st = L.stream_ptr()
    row = "C %4d %3d^2 act %d x%2d:" % (C, S, act, cnt)
    for plan in (0, 2, 3):
        L.set_option("gn_plan", plan)
        if act == 2:
            f = lambda: L.check(lib.acr_groupnorm_fwd_mask_f32(L.ptr(x), L.ptr(r), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(stats), N, C, S * S, 1e-5, L.ptr(mask), st), "f")
            g = lambda: L.check(lib.acr_groupnorm_bwd_mask_f32(L.ptr(dy), L.ptr(x), L.ptr(mask), L.ptr(w), L.ptr(b), L.ptr(stats), L.ptr(dx), L.ptr(dr), L.ptr(part[0]), L.ptr(part[1]), L.ptr(dgb[0]), L.ptr(dgb[1]), N, C, S * S, st), "b")
        else:
            f = lambda: L.check(lib.acr_groupnorm_fwd_f32(L.ptr(x), None, L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(stats), N, C, S * S, 1e-5, act, None, st), "f")
            g = lambda: L.check(lib.acr_groupnorm_bwd_f32(L.ptr(dy), L.ptr(x), None, L.ptr(w), L.ptr(b), L.ptr(stats), L.ptr(dx), None, L.ptr(part[0]), L.ptr(part[1]), L.ptr(dgb[0]), L.ptr(dgb[1]), N, C, S * S, act, st), "b")
        tf, tb = t(f), t(g)
        tot.setdefault(plan, [0.0, 0.0]); tot[plan][0] += cnt * tf; tot[plan][1] += cnt * tb
        row += "  plan %d fwd %6.1f bwd %6.1f us" % (plan, tf, tb)
    print(row, flush=True)
    del x, r, dy, y, dx, dr, mask
for plan, (a, b2) in tot.items():
    print("plan %d per step: fwd %.2f ms  bwd %.2f ms" % (plan, a / 1e3, b2 / 1e3))
L.set_option("gn_plan", 0)
