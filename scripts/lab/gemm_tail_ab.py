"""GPU lab: the block GEMM shapes of the bench (M = 25 120 tokens) with and without the K-split tail tiles
(ACR_OPT_GEMM_F32_NOTAIL), NT forward with bias and the GELU / GELU' epilogues; max |difference| between the two."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import ops, _lib
dev = torch.device("cuda:0")
def t(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it
M = 25120
OPT = sys.argv[1] if len(sys.argv) > 1 else "notail"      # the option switched off (1) / on (0): notail | scalar_epi
torch.manual_seed(0)
for name, N, K, act in (("qkv", 2304, 768, 0), ("proj", 768, 768, 0), ("fc1+gelu", 3072, 768, 1), ("fc2", 768, 3072, 0),
                        ("fc1 dx", 768, 3072, 0), ("fc2 dx*gelu'", 3072, 768, 2), ("qkv dx", 768, 2304, 0)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    aux = torch.randn(M, N, device=dev) if act != 1 else None
    outs = {}
    for notail in (1, 0):
        _lib.set_option("gemm_f32_" + OPT, notail)
        y = torch.empty(M, N, device=dev); y2 = torch.empty(M, N, device=dev) if act == 1 else None
        fn = lambda: ops.gemm_f32_raw("nt", x, w, y, bias=None if act == 2 else b, aux=aux, act=act, c2=y2)
        ms = t(fn)
        outs[notail] = (ms, y.clone(), None if y2 is None else y2.clone())
    tiles = ((M + 127) // 128) * (N // 128)
    d = float((outs[0][1] - outs[1][1]).abs().max())
    d2 = float((outs[0][2] - outs[1][2]).abs().max()) if act == 1 else 0.0
    fl = 2.0 * M * N * K
    print("%-14s N %4d K %4d: %4d tiles (%.2f rounds, tail %3d)  option=1 %.3f ms %.1f TF   option=0 %.3f ms %.1f TF   max|diff| %.2e %.2e (|y| %.1f)" % (
        name, N, K, tiles, tiles / 512, tiles % 256, outs[1][0], fl / outs[1][0] / 1e9, outs[0][0], fl / outs[0][0] / 1e9, d, d2, float(outs[1][1].abs().max())), flush=True)
