"""GPU lab: per-phase cycles (s_memtime, wave 0 of every workgroup) of the resident-score fp32 attention backward
(attn_bwd_sres_kernel: dK/dV and dQ bodies).  Needs scripts/lab/_build/libacr_hip_tl.so (scripts/lab/build_variant.sh -H srestl attn_f32_sres.hip -DLAB_TL: the hooked round-5 sources)."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acr_wsss_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "scripts", "lab", "_build", "libacr_srestl.so")
from acr_wsss_amd import ops
raw = ctypes.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
B, T, H = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 785, 12
qkv = (1.5 * torch.randn(B, T, 3 * H * 64, device=dev)).requires_grad_(True)
do = torch.randn(B, T, H * 64, device=dev)
gst = torch.zeros(B, T, ops.pad4(T), device=dev); gst[:, :, :T] = torch.randn(B, T, T, device=dev) * 1e-3
stack = ops.MeanStack(B, 1, T, dev)
for _ in range(3):
    qkv.grad = None
    o, pm = ops.attention_core(qkv, H, stack, 0, None)
    torch.autograd.backward([o, pm], [do, gst[:, :, :T]])
torch.cuda.synchronize()
n = 16384
buf = (ctypes.c_ulonglong * (8 * n))()
raw.acr_lab_attn_read(buf, 8 * n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.float64)
names = ["barrier (vmcnt + s_barrier)", "DMA issue + loads", "dP chain (32 MFMA)", "VALU (exp, dS)", "accumulate chains (64 / 96 MFMA)"]
for kind, nm, mf in ((2, "dK/dV body", 96 + 32), (1, "dQ body", 64 + 32)):
    sel = a[a[:, 7] == kind]
    if not len(sel):
        continue
    steps = sel[:, 6]
    tot = (sel[:, :5].sum(1) / steps).mean()
    print("%s: %d workgroups, %.0f steps each, %.0f cycles per step (own MFMA work: %d cycles)" % (nm, len(sel), steps.mean(), tot, (mf - 32) * 64 + 32 * 64))
    for i, ph in enumerate(names):
        print("   %-34s %8.0f cycles / step" % (ph, (sel[:, i] / steps).mean()))
