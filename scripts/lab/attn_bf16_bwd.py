"""GPU lab: bf16 attention backward at the bench geometry (B = 32 views, H = 12, T = 785), with and without the head-mean
gradient G: ms per call (HIP events on the launch stream; run under rocprofv3 --kernel-trace --stats for the per-kernel split).
A lab build of the library is picked with ACR_LIB_PATH.  usage: attn_bf16_bwd.py [B] [T] [reps]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import ops
dev = torch.device("cuda:0")
H = 12
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 785
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
torch.manual_seed(0)
qkv = (1.5 * torch.randn(B, T, 3 * H * 64, device=dev)).bfloat16().requires_grad_(True)
do = torch.randn(B, T, H * 64, device=dev).bfloat16()
gst = torch.zeros(B, T, ops.pad4(T), device=dev)
gst[:, :, :T] = torch.randn(B, T, T, device=dev) * 1e-3
gpm = gst[:, :, :T]
stack = ops.MeanStack(B, 1, T, dev)
for with_g in (True, False):
    o, pm = ops.attention_core(qkv, H, stack, 0, None)
    outs, grads = ([o, pm], [do, gpm]) if with_g else ([o], [do])
    for _ in range(3):
        qkv.grad = None
        torch.autograd.backward(outs, grads, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        qkv.grad = None
        torch.autograd.backward(outs, grads, retain_graph=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%s  B %d T %d %s: backward %.3f ms per call (delta + dQ + dK/dV), %.0f TF algorithmic (4 products); |dqkv| %.4e" % (
        os.path.basename(os.environ.get("ACR_LIB_PATH", "libacr_hip.so")), B, T, "with G" if with_g else "no G  ", ms,
        4 * 2.0 * T * T * 64 * B * H / ms * 1e-9, float(qkv.grad.float().abs().mean())), flush=True)
