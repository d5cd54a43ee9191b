"""GPU lab: fp32 attention at the bench geometry (B = 32 views, H = 12, T = 785), recompute generation vs resident scores vs resident scores with split
products (csrc/attn_f32_x3.hip):
forward (+ head mean) and backward (with the head-mean gradient) time per launch, HIP events on the launch stream, plus a
max-abs comparison of the two generations' outputs.  usage: attn_gen.py [B] [T]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import _lib
if os.environ.get("ACR_LAB_LIB"):      # a lab build of the library (scripts/lab/_build/*.so)
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
res = {}
for gen in ("recompute", "scores nw4", "split x3 notail", "split x3"):
    ops.ATTN_F32_SCORES = gen != "recompute"
    _lib.set_option("attn_f32_nosplittail", 1 if gen.endswith("notail") else 0)
    stack = ops.MeanStack(B, 1, T, dev)
    def run():
        qkv.grad = None
        o, pm = ops.attention_core(qkv, H, stack, 0, None, 1 if gen.startswith("split") else 0)
        return o, pm
    o, pm = run()
    torch.autograd.backward([o, pm], [do, gpm])
    torch.cuda.synchronize()
    res[gen] = (o.detach().clone(), pm.detach().clone(), qkv.grad.detach().clone())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    reps, tf, tb = 10, 0.0, 0.0
    for _ in range(reps):
        ev[0].record()
        o, pm = run()
        ev[1].record()
        torch.autograd.backward([o, pm], [do, gpm])
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    prod = 2.0 * T * T * 64 * B * H
    print("%-10s B %d T %d: fwd+pmean %.3f ms (%.1f TF algorithmic)   bwd %.3f ms (%.1f TF algorithmic)" % (
        gen, B, T, tf / reps, 2 * prod / (tf / reps) * 1e-9, tb / reps, 4 * prod / (tb / reps) * 1e-9), flush=True)
a, b = res["scores nw4"], res["split x3"]
for name, x, y in zip(("o", "pmean", "dqkv"), a, b):
    print("%-6s max |scores - split| = %.3e  (max |x| %.3e)  finite %s" % (name, float((x - y).abs().max()), float(x.abs().max()),
                                                                             bool(torch.isfinite(y).all())))
