"""GPU lab: are the gradients of consecutive train_steps at learning rate 0 (same parameters, same batch) bit-identical?
Which parameters differ between a 2-sample shard average and the 4-sample batch?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.train import PolyOptimizer, train_step
torch.manual_seed(5)
m = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False, math="f32_split").to("cuda:0")
with torch.no_grad():
    for blk in m.pretrained.model.blocks:
        blk.attn.qkv.weight.mul_(4.0)
g = torch.Generator().manual_seed(11)
img = torch.randn(4, 3, 64, 64, generator=g).cuda()
label = (torch.rand(4, 20, generator=g) > 0.7).float().cuda()
opt = PolyOptimizer(m.parameters(), lr=0.0, weight_decay=5e-4, max_step=10)
names = [n for n, _ in m.named_parameters()]
def grads():
    return [(p.grad.detach().clone() if p.grad is not None else None) for p in m.parameters()]
runs = {}
for tag, sl in (("full", slice(0, 4)), ("a", slice(0, 2)), ("b", slice(2, 4))):
    gs = []
    for it in range(3):
        train_step(m, opt, img[sl], label[sl], 125)
        gs.append(grads())
    for it in (1, 2):
        bad = [(n, float((x - y).abs().max())) for n, x, y in zip(names, gs[0], gs[it]) if x is not None and not torch.equal(x, y)]
        print("%s: step %d vs step 0: %d tensors differ %s" % (tag, it, len(bad), bad[:5]))
    runs[tag] = gs[0]
worst = []
for n, f, a, b in zip(names, runs["full"], runs["a"], runs["b"]):
    if f is None:
        continue
    d = (0.5 * (a + b) - f).abs().max()
    worst.append((float(d / f.abs().max().clamp_min(1e-30)), float(d), float(f.abs().max()), n))
worst.sort(reverse=True)
for w in worst[:12]:
    print("rel %.3e abs %.3e max %.3e  %s" % w)
gmax = max(w[2] for w in worst)
print("global max |g| %.3e; worst abs diff %.3e" % (gmax, max(w[1] for w in worst)))
