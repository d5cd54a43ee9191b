"""GPU lab: per-phase cycles (s_memtime, wave 0 of every workgroup) of the split-product attention (attn_f32_x3.hip: forward,
dK/dV and dQ bodies).  Needs scripts/lab/_build/libacr_x3tl.so:  scripts/lab/build_variant.sh -H x3tl attn_f32_x3.hip -DLAB_TL  (the hooked round-5 sources)"""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acr_wsss_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "scripts", "lab", "_build", "libacr_x3tl.so")
from acr_wsss_amd import ops
raw = ctypes.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
B, T, H = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 785, 12
qkv = (1.5 * torch.randn(B, T, 3 * H * 64, device=dev)).requires_grad_(True)
do = torch.randn(B, T, H * 64, device=dev)
gst = torch.zeros(B, T, ops.pad4(T), device=dev); gst[:, :, :T] = torch.randn(B, T, T, device=dev) * 1e-3
stack = ops.MeanStack(B, 1, T, dev)
names = ["s_barrier (after the counted vmcnt wait)", "DMA issue + loads", "first product (24 MFMA)", "VALU (scale / exp / dS / split)", "accumulate products (48 / 72 MFMA)"]
n = 16384
def dump(kinds):
    buf = (ctypes.c_ulonglong * (8 * n))()
    raw.acr_lab_x3_read(buf, 8 * n)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.float64)
    for kind, nm, mf in kinds:
        sel = a[a[:, 7] == kind]
        if not len(sel):
            continue
        steps = sel[:, 6]
        tot = (sel[:, :6].sum(1) / steps).mean()
        print("%s: %d workgroups, %.0f steps each, %.0f cycles per step (own MFMA work: %d cycles)" % (nm, len(sel), steps.mean(), tot, mf * 32))
        for i, ph in enumerate(names):
            print("   %-38s %8.0f cycles / step" % (ph, (sel[:, i] / steps).mean()))
        print("   %-38s %8.0f cycles / step" % ("counted vmcnt wait before the barrier", (sel[:, 5] / steps).mean()))
for _ in range(2):
    qkv.grad = None
    o, pm = ops.attention_core(qkv, H, stack, 0, None, 1)
torch.cuda.synchronize()
dump(((3, "forward", 48),))
torch.autograd.backward([o, pm], [do, gst[:, :, :T]])
torch.cuda.synchronize()
dump(((2, "dK/dV body", 72), (1, "dQ body", 48)))
