"""GPU lab: which Python lines cause the big elementwise copies / fills / cats of one fp32 training step (torch.profiler)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch.profiler import profile, ProfilerActivity
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.train import PolyOptimizer, train_step
dev = torch.device("cuda:0")
torch.manual_seed(0)
math = sys.argv[1] if len(sys.argv) > 1 else "f32_split"
model = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False, math=math).to(dev).train()
opt = PolyOptimizer(model.parameters(), lr=0.05, weight_decay=5e-4, max_step=1000)
img = torch.randn(16, 3, 448, 448, device=dev)
lab = torch.zeros(16, 20, device=dev); lab[:, 0] = 1
for _ in range(2):
    train_step(model, opt, img, lab, 125)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    train_step(model, opt, img, lab, 125)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.name.startswith("aten::") and e.self_device_time_total > 8:
        st = [s for s in (e.stack or []) if "acr_wsss_amd" in s or "bench" in s][:2]
        rows.append((e.self_device_time_total, e.name, str(e.input_shapes)[:70], " <- ".join(x.split("/")[-1][:60] for x in st)))
rows.sort(reverse=True)
tot = 0
for r in rows[:70]:
    tot += r[0]
    print("%8.1f us  %-14s %-70s %s" % r)
print("total listed %.1f us" % tot)
