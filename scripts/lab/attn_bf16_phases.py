"""GPU lab: per-phase cycles (s_memtime, wave 0 of every workgroup) of the bf16 attention dQ sweep.
Needs scripts/lab/_build/libacr_tlb.so (build_variant.sh -H tlb attn_bf16.hip -DLAB_TLB: the hooked round-5 sources)."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acr_wsss_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "scripts", "lab", "_build", "libacr_tlb.so")
from acr_wsss_amd import ops
raw = ctypes.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
B, T, H = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 785, 12
qkv = (1.5 * torch.randn(B, T, 3 * H * 64, device=dev)).bfloat16().requires_grad_(True)
do = torch.randn(B, T, H * 64, device=dev).bfloat16()
gst = torch.zeros(B, T, ops.pad4(T), device=dev); gst[:, :, :T] = torch.randn(B, T, T, device=dev) * 1e-3
stack = ops.MeanStack(B, 1, T, dev)
for _ in range(3):
    qkv.grad = None
    o, pm = ops.attention_core(qkv, H, stack, 0, None)
    torch.autograd.backward([o, pm], [do, gst[:, :, :T]])
torch.cuda.synchronize()
n = 16384
buf = (ctypes.c_ulonglong * (8 * n))()
raw.acr_lab_attnb_read(buf, 8 * n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.float64)
names = ["issue tile + G loads", "S / dP products (2 x 8 MFMA)", "softmax / dS VALU (2 x 16 elements)", "dQ (+Y) products (2 x 8 MFMA)", "registers -> LDS (waits for the loads)", "barrier"]
sel = a[a[:, 7] == 1]
steps = sel[:, 6]
print("dQ sweep: %d workgroups, %.0f steps each, %.0f cycles per 64-key step" % (len(sel), steps.mean(), (sel[:, :6].sum(1) / steps).mean()))
for i, ph in enumerate(names):
    print("   %-40s %8.0f cycles / step" % (ph, (sel[:, i] / steps).mean()))
