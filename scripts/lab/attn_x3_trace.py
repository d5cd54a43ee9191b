"""GPU lab: split-product attention at the bench geometry, forward (+ head mean) and backward (row term + sweeps) six times -- run
under rocprofv3 --kernel-trace and summarise with kstats.py.  usage: attn_x3_trace.py [B] [T]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import _lib
if os.environ.get("ACR_LAB_LIB"):
    _lib.LIB_PATH = os.environ["ACR_LAB_LIB"]
from acr_wsss_amd import ops
dev = torch.device("cuda:0")
H = 12
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 785
torch.manual_seed(0)
qkv = (1.5 * torch.randn(B, T, 3 * H * 64, device=dev)).requires_grad_(True)
do = torch.randn(B, T, H * 64, device=dev)
gst = torch.zeros(B, T, ops.pad4(T), device=dev)
gst[:, :, :T] = torch.randn(B, T, T, device=dev) * 1e-3
gpm = gst[:, :, :T]
stack = ops.MeanStack(B, 1, T, dev)
for _ in range(8):
    qkv.grad = None
    o, pm = ops.attention_core(qkv, H, stack, 0, None, 1)
    torch.autograd.backward([o, pm], [do, gpm])
torch.cuda.synchronize()
print("dqkv checksum %.9e" % float(qkv.grad.double().abs().sum()))
