"""GPU lab: fp32 attention forward / backward time per image vs batch (how much the last, partly filled round of workgroups
costs) and vs T (query-tile padding): 12 heads, T = 785 unless given."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import ops
dev = torch.device("cuda:0")
H = 12
for T in (785, 768, 896):
    for B in (16, 32, 48, 64, 96):
        qkv = torch.randn(B, T, 3 * H * 64, device=dev).requires_grad_(True)
        o, _ = ops.attention_core(qkv, H, None, 0, None)
        do = torch.randn_like(o)
        o.backward(do)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        reps = 5
        tf = tb = 0.0
        for _ in range(reps):
            qkv.grad = None
            ev[0].record()
            o, _ = ops.attention_core(qkv, H, None, 0, None)
            ev[1].record()
            o.backward(do)
            ev[2].record()
            torch.cuda.synchronize()
            tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
        nwg = B * H * ((T + 127) // 128)
        print("T %4d B %3d: %5d workgroups (%.2f rounds of 768 / %.2f of 512)  fwd %7.1f us (%.2f us/img)  bwd %7.1f us (%.2f us/img)" % (
            T, B, nwg, nwg / 768, nwg / 512, 1e3 * tf / reps, 1e3 * tf / reps / B, 1e3 * tb / reps, 1e3 * tb / reps / B))
