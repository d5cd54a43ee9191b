"""GPU lab: does an HBM-bound kernel (the image pass) hide under an MFMA-bound one (the weight-gradient product on images) when they
are launched on two streams?  Serial time of n (product + pass) pairs vs the two loops on separate streams."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import ops
dev = torch.device("cuda:0")
M, N, K = 25120, 3072, 768
x = torch.randn(M, K, device=dev); dy = torch.randn(M, N, device=dev); dw = torch.empty(N, K, device=dev)
xi, dyi = ops.x3_image(x), ops.x3_image(dy)
big = torch.randn(M, N, device=dev)
n = 20
def gem(): ops.gemm_x3("tn", dyi, xi, dw, M)
def pas(): ops.x3_image(big)
def ln(): torch.nn.functional.layer_norm(big, (N,))
for name, mem in (("image pass", pas), ("torch layer_norm", ln)):
    for f in (gem, mem): f()
    torch.cuda.synchronize()
    def timed(fn):
        torch.cuda.synchronize(); t0 = time.time(); fn(); torch.cuda.synchronize(); return (time.time() - t0) * 1e3
    tg = timed(lambda: [gem() for _ in range(n)])
    tm = timed(lambda: [mem() for _ in range(n)])
    ts = timed(lambda: [(gem(), mem()) for _ in range(n)])
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def conc():
        with torch.cuda.stream(s1):
            for _ in range(n): gem()
        with torch.cuda.stream(s2):
            for _ in range(n): mem()
    conc(); tc = timed(conc)
    print("%-18s: products alone %.2f ms, memory kernel alone %.2f ms, serial %.2f ms, two streams %.2f ms" % (name, tg, tm, ts, tc), flush=True)
