"""GPU lab: in which ORDER do the parameters' gradients arrive in a training backward (post-accumulate hooks)?  Prints the last
arrivals and, per GradSync bucket, the parameter that completes it.  usage: grad_arrival_order.py [f32_split|f32]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.dp import GradSync
from acr_wsss_amd.train import PolyOptimizer, train_step
math = sys.argv[1] if len(sys.argv) > 1 else "f32_split"
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False, math=math).to(dev).train()
names = {p: n for n, p in model.named_parameters()}
order = []
for p in model.parameters():
    p.register_post_accumulate_grad_hook(lambda q: order.append(q))
img = torch.randn(4, 3, 448, 448, device=dev)
label = (torch.rand(4, 20, device=dev) > 0.85).float()
opt = PolyOptimizer(model.parameters(), lr=0.05, weight_decay=5e-4, max_step=100000)
sync = GradSync(model.parameters(), late_params=model.late_gradient_parameters())
for _ in range(2):
    order.clear()
    train_step(model, opt, img, label, 125, grad_sync=sync)
pos = {p: i for i, p in enumerate(order)}
print("%d gradients; the last 14 to arrive:" % len(order))
for p in order[-14:]:
    print("   %4d  %s" % (pos[p], names[p]))
for b in sync.buckets:
    have = [p for p in b.params if p in pos]
    last = max(have, key=lambda p: pos[p])
    print("bucket %d (%.1f MB, %d params, %d without gradient): completed by #%d %s" % (
        b.index, b.flat.numel() * 4 / 2 ** 20, len(b.params), len(b.params) - len(have), pos[last], names[last]))
