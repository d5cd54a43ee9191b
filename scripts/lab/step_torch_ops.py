"""GPU lab: which stock torch kernels are left in one f32_split training step (torch.profiler, device-side table sorted by calls).
usage: step_torch_ops.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.train import PolyOptimizer, train_step
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False, math="f32_split").to(dev).train()
g = torch.Generator(device="cpu").manual_seed(1000)
img = torch.randn(16, 3, 448, 448, generator=g).to(dev)
label = (torch.rand(16, 20, generator=g) > 0.85).float().to(dev); label[:, 0] = 1.0
opt = PolyOptimizer(model.parameters(), lr=0.05, weight_decay=5e-4, max_step=100000)
for _ in range(3):
    train_step(model, opt, img, label, 125)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    train_step(model, opt, img, label, 125)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_type.name == "CUDA" or not e.name.startswith("aten::"):
        continue
    kt = sum(k.duration for k in e.kernels) if e.kernels else 0
    if e.kernels:
        st = [s for s in (e.stack or []) if "acr_wsss_amd" in s or "bench" in s]
        rows.append((e.name, tuple(str(x) for x in (e.input_shapes or [])), kt, st[0][-70:] if st else ""))
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for n, sh, kt, st in rows:
    agg[(n, " ".join(sh)[:110])][0] += 1; agg[(n, " ".join(sh)[:110])][1] += kt
for (n, st), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
    print("%4d x %-34s %8.1f us  %s" % (c, n, t, st))
