"""Lab (GPU box): per-phase cycles of the fp32 attention forward; needs a library whose attn_f32_dma.hip is the hooked round-5 source: scripts/lab/build_variant.sh -H stamp attn_f32_dma.hip -DLAB_STAMP."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import ops, _lib as L
lib = ctypes.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
B, T, H = 32, 785, 12
qkv = torch.randn(B, T, 3 * H * 64, device=dev)
for _ in range(3):
    o, _ = ops.attention_core(qkv, H, None, 0, None)
torch.cuda.synchronize()
n = B * H * ((T + 127) // 128)
buf = (ctypes.c_ulonglong * (8 * n))()
lib.acr_lab_read_stamps(buf, 8 * n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.float64)
a = a[a[:, 7] > 0]
steps = a[:, 5]
print("workgroups %d, steps/wg %.1f" % (len(a), steps.mean()))
for i, nm in enumerate(["barrier wait", "dma issue", "S chain (32 MFMA)", "softmax VALU", "PV chain (32 MFMA)"]):
    print("%-22s %8.0f cycles / step" % (nm, (a[:, i] / steps).mean()))
print("%-22s %8.0f cycles / step (whole loop / steps)" % ("total", (a[:, 6] / steps).mean()))
