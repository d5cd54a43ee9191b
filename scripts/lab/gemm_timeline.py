"""GPU lab: wall-clock timeline (s_memrealtime, 100 MHz) of every workgroup of one fp32 NT GEMM launch: when it started, when
its first operand chunk had landed, when its K loop ended, when its epilogue stores had drained.  Needs the library built
with -DLAB_TL from the hooked round-5 sources (scripts/lab/build_variant.sh -H hip_tl gemm_f32.hip -DLAB_TL -> scripts/lab/_build/libacr_hip_tl.so).  usage: gemm_timeline.py N K"""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acr_wsss_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "scripts", "lab", "_build", "libacr_hip_tl.so")
from acr_wsss_amd import ops
raw = ctypes.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
M = 25120
N, K = int(sys.argv[1]), int(sys.argv[2])
L.set_option("gemm_f32_notail", 1)
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
for _ in range(5):
    ops.gemm_f32_raw("nt", x, w, y, bias=b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm_f32_raw("nt", x, w, y, bias=b); e1.record(); torch.cuda.synchronize()
n = ((M + 127) // 128) * (N // 128)
buf = (ctypes.c_ulonglong * (4 * n))()
raw.acr_lab_tl_read(buf, 4 * n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.float64) * 0.01      # microseconds
t0 = a[:, 0].min()
a -= t0
print("N %d K %d: %d workgroups, event time %.1f us, first start -> last end %.1f us" % (N, K, n, e0.elapsed_time(e1) * 1e3, a[:, 3].max()))
print("  per workgroup (us): start->first chunk landed %.2f (p90 %.2f)   K loop %.2f (p10 %.2f p90 %.2f)   epilogue %.2f (p90 %.2f)   lifetime %.2f" % (
    (a[:, 1] - a[:, 0]).mean(), np.percentile(a[:, 1] - a[:, 0], 90), (a[:, 2] - a[:, 1]).mean(), np.percentile(a[:, 2] - a[:, 1], 10),
    np.percentile(a[:, 2] - a[:, 1], 90), (a[:, 3] - a[:, 2]).mean(), np.percentile(a[:, 3] - a[:, 2], 90), (a[:, 3] - a[:, 0]).mean()))
order = np.argsort(a[:, 0])
st = a[order, 0]
for lo in range(0, n, 256):
    seg = a[order[lo:lo + 256]]
    print("  workgroups %4d..%4d by start time: start %7.1f .. %7.1f   loop %6.1f   epilogue %5.1f   end %7.1f .. %7.1f" % (
        lo, min(lo + 255, n - 1), seg[:, 0].min(), seg[:, 0].max(), (seg[:, 2] - seg[:, 1]).mean(), (seg[:, 3] - seg[:, 2]).mean(), seg[:, 3].min(), seg[:, 3].max()))
