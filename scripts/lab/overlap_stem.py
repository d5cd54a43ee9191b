"""GPU lab (round 6): do the stem's weight-gradient products (VALU / MFMA work at 2.1-2.4 GHz, not power-limited) hide the GroupNorm
backward passes (pure HBM streams) when the two run on separate streams?  For a few (convolution, norm) pairs of the step: n launches
of each alone, the pairs in series on one stream, and the two loops on two streams.  usage: overlap_stem.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
N = 32
def timed(fn):
    torch.cuda.synchronize(); t0 = time.time(); fn(); torch.cuda.synchronize(); return (time.time() - t0) * 1e3
n = 10
# (cout, cin, S, kind) of the convolution whose weight gradient runs; (C, S, act) of the norm whose backward runs beside it
cases = [((256, 64, 112, 1), (256, 112, 2)), ((64, 64, 112, 3), (64, 112, 1)), ((512, 128, 56, 1), (512, 56, 2)), ((128, 128, 56, 3), (128, 56, 1)),
         ((1024, 256, 28, 1), (1024, 28, 2)), ((256, 256, 28, 3), (256, 28, 1))]
tot = [0.0, 0.0, 0.0, 0.0]
for (co, ci, S, k), (C, S2, act) in cases:
    dy = torch.randn(N, co, S, S, device=dev); x = torch.randn(N, ci, S, S, device=dev)
    if k == 1:
        ws = torch.empty(lib.acr_conv1x1_wgrad_f32_ws_floats(N, co, ci, S * S), device=dev); dw = torch.empty(co, ci, device=dev)
        conv = lambda st: L.check(lib.acr_conv1x1_wgrad_f32(1, L.ptr(dy), L.ptr(x), N, co, ci, S * S, L.ptr(ws), L.ptr(dw), st), "w1")
    else:
        ws = torch.empty(lib.acr_conv3x3_wgrad_ws_floats(N, co, ci, S, S), device=dev); dw = torch.empty(co, 9 * ci, device=dev)
        conv = lambda st: L.check(lib.acr_conv3x3_wgrad_f32(1, L.ptr(dy), L.ptr(x), N, co, ci, S, S, L.ptr(ws), L.ptr(dw), st), "w3")
    gx = torch.randn(N, C, S2, S2, device=dev); gdy = torch.randn(N, C, S2, S2, device=dev); gdx = torch.empty_like(gx); gdr = torch.empty_like(gx)
    mask = torch.randint(0, 16, (gx.numel() // 4,), dtype=torch.uint8, device=dev)
    w, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    stats = torch.zeros(N * 64, device=dev); stats[1::2] = 1.0
    part = torch.empty(2, N, C, device=dev); dgb = torch.empty(2, C, device=dev)
    if act == 2:
        gn = lambda st: L.check(lib.acr_groupnorm_bwd_mask_f32(L.ptr(gdy), L.ptr(gx), L.ptr(mask), L.ptr(w), L.ptr(b), L.ptr(stats), L.ptr(gdx), L.ptr(gdr), L.ptr(part[0]), L.ptr(part[1]), L.ptr(dgb[0]), L.ptr(dgb[1]), N, C, S2 * S2, st), "g")
    else:
        gn = lambda st: L.check(lib.acr_groupnorm_bwd_f32(L.ptr(gdy), L.ptr(gx), None, L.ptr(w), L.ptr(b), L.ptr(stats), L.ptr(gdx), None, L.ptr(part[0]), L.ptr(part[1]), L.ptr(dgb[0]), L.ptr(dgb[1]), N, C, S2 * S2, act, st), "g")
    s0 = L.stream_ptr()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    p1, p2 = L.c_void_p(s1.cuda_stream), L.c_void_p(s2.cuda_stream)
    for f in (conv, gn): f(s0)
    tc = timed(lambda: [conv(s0) for _ in range(n)]) / n
    tg = timed(lambda: [gn(s0) for _ in range(n)]) / n
    ts = timed(lambda: [(conv(s0), gn(s0)) for _ in range(n)]) / n
    def both():
        for _ in range(n): conv(p1)
        for _ in range(n): gn(p2)
    both(); t2 = timed(both) / n
    print("wgrad %dx%d k%d at %d^2 | gn bwd C %d at %d^2 act %d:  conv %.3f  gn %.3f  serial %.3f  two streams %.3f ms" % (co, ci, k, S, C, S2, act, tc, tg, ts, t2), flush=True)
    for i, v in enumerate((tc, tg, ts, t2)): tot[i] += v
print("sum: conv %.2f  gn %.2f  serial %.2f  two streams %.2f ms" % tuple(tot))
