"""Times acr_attn_bwd with and without the head-mean gradient G (isolates the cost of pulling G)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acr_wsss_amd import _lib as L, ops
lib = L.load(); dev = torch.device("cuda:0")
B, H, T = 32, 12, 785
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, T, 3 * H * 64, generator=g).to(dev).bfloat16()
d_o = torch.randn(B, T, H * 64, generator=g).to(dev).bfloat16()
gm = (torch.randn(B, T, ops.pad4(T), generator=g).to(dev) * 1e-3)[:, :, :T]
o = torch.empty(B, T, H * 64, dtype=torch.bfloat16, device=dev)
lse2 = torch.empty(B, H, T, dtype=torch.float32, device=dev)
dqkv = torch.empty_like(qkv); delta = torch.empty(B, H, T, dtype=torch.float32, device=dev)
d = ops._desc(B, H, T, torch.bfloat16)
qp, kp, vp = ops._qkv_ptrs(qkv, H); dqp, dkp, dvp = ops._qkv_ptrs(dqkv, H); st = L.stream_ptr()
L.check(lib.acr_attn_fwd(d, qp, kp, vp, L.ptr(o), L.ptr(lse2), None, 0, 0, st), "fwd")
def run(with_g):
    if with_g:
        L.check(lib.acr_attn_bwd(d, qp, kp, vp, L.ptr(o), L.ptr(d_o), L.ptr(lse2), L.ptr(gm), gm.stride(0), gm.stride(1), dqp, dkp, dvp, L.ptr(delta), st), "bwd")
    else:
        L.check(lib.acr_attn_bwd(d, qp, kp, vp, L.ptr(o), L.ptr(d_o), L.ptr(lse2), None, 0, 0, dqp, dkp, dvp, L.ptr(delta), st), "bwd")
for with_g in (True, False, True, False):
    for _ in range(3): run(with_g)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(with_g)
    e1.record(); torch.cuda.synchronize()
    print("with G" if with_g else "no G  ", "%.1f us per attn_bwd" % (e0.elapsed_time(e1) / 20 * 1e3), flush=True)
