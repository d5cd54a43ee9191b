"""GPU lab: bf16 attention backward variants (option dq_variant / dkdv_variant) against the default at the bench geometry:
ms per call (HIP events) and max |difference| of dqkv.  usage: attn_bf16_variants.py [B] [T] [reps]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import ops, _lib
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
# the shipped variants; a lab build adds its own (this round: 16 / 163 = dQ on 16-row waves at 4 / 3 waves per SIMD, not kept)
variants = [("default", {}), ("dQ 2-wave sweep", {"dq_variant": 2}), ("dQ 4-wave sweep", {"dq_variant": 4})]
for v in os.environ.get("ACR_LAB_DQ_VARIANTS", "").split(","):
    if v.strip():
        variants.append(("dq_variant %s" % v.strip(), {"dq_variant": int(v)}))
for with_g in (True, False):
    o, pm = ops.attention_core(qkv, H, stack, 0, None)
    outs, grads = ([o, pm], [do, gpm]) if with_g else ([o], [do])
    ref = None
    for rnd in range(2):                                # two rounds: the chip's clock drifts, compare within a round
        for name, opts in variants:
            for kname in ("dq_variant", "bwd16"):
                if kname in _lib.OPTIONS:
                    _lib.set_option(kname, opts.get(kname, 0))
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
            gr = qkv.grad.float()
            if ref is None:
                ref = gr.clone()
            print("%-22s %s: %.3f ms per call; max |dqkv - default| %.3e (max |dqkv| %.3e) finite %s" % (
                name, "with G" if with_g else "no G  ", e0.elapsed_time(e1) / reps, float((gr - ref).abs().max()), float(ref.abs().max()),
                bool(torch.isfinite(gr).all())), flush=True)
for kname in ("dq_variant", "bwd16"):
    if kname in _lib.OPTIONS:
        _lib.set_option(kname, 0)
