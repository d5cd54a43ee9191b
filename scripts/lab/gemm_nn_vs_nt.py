"""GPU lab: the fp32 input-gradient GEMM dx = dy W as NN (W as stored, (out, in)) vs NT on a transposed copy W^T (in, out)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import ops
dev = torch.device("cuda:0")
M = 25120
def t(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it
for (nout, nin) in ((3072, 768), (768, 3072), (2304, 768), (768, 768)):       # Linear(in -> out); dx is (M, in)
    dy = torch.randn(M, nout, device=dev)
    w = torch.randn(nout, nin, device=dev) * nin ** -0.5
    wt = w.t().contiguous()
    dx1 = torch.empty(M, nin, device=dev); dx2 = torch.empty(M, nin, device=dev)
    a = t(lambda: ops.gemm_f32_raw("nn", dy, w, dx1))
    b = t(lambda: ops.gemm_f32_raw("nt", dy, wt, dx2))
    fl = 2.0 * M * nout * nin
    print("Linear %4d -> %4d  dx: NN %.3f ms (%.1f TF)   NT on W^T %.3f ms (%.1f TF)   max diff %.2e" % (
        nin, nout, a, fl / a / 1e9, b, fl / b / 1e9, (dx1 - dx2).abs().max()))
