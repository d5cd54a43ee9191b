"""GPU lab: are consecutive training steps at learning rate 0 (same parameters, same batch) bit-identical?  Where does the first
difference appear (forward: loss terms / head-mean maps / logits; backward: per-parameter gradients in module order)?
usage: step_repeat.py [size] [batch] [mode]   mode: step = train_step (optimizer + image refresh), loop = forward/backward only"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.train import PolyOptimizer, acr_loss, refresh_weight_transposes
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mode = sys.argv[3] if len(sys.argv) > 3 else "step"
math = sys.argv[4] if len(sys.argv) > 4 else "f32_split"
if len(sys.argv) > 5 and sys.argv[5] == "det":          # MIOpen held to its deterministic solvers (what tests/conftest.py sets)
    torch.backends.cudnn.deterministic = True
torch.manual_seed(5)
m = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False, math=math).to("cuda:0")
with torch.no_grad():
    for blk in m.pretrained.model.blocks:
        blk.attn.qkv.weight.mul_(4.0)
g = torch.Generator().manual_seed(11)
img = torch.randn(batch, 3, size, size, generator=g).cuda()
label = (torch.rand(batch, 20, generator=g) > 0.7).float().cuda()
opt = PolyOptimizer(m.parameters(), lr=0.0, weight_decay=5e-4, max_step=10)
names = [n for n, _ in m.named_parameters()]
p0 = [p.detach().clone() for p in m.parameters()]
recs = []
m.train()
for it in range(4):
    opt.zero_grad(set_to_none=True)
    cl, al = m.forward_mirror(img, img.flip(-1))
    loss, terms = acr_loss(cl, al, label, size // 16, 125)
    loss.backward()
    rec = {"loss": loss.detach().clone(), "cls_align": terms["cls_align"].detach().clone(), "aff_align": terms["aff_align"].detach().clone(),
           "logits": torch.cat([c.detach().reshape(-1) for c in cl[:4]]).clone(), "maps": al.stacked.detach().clone()}
    rec["grads"] = [(p.grad.detach().clone() if p.grad is not None else None) for p in m.parameters()]
    recs.append(rec)
    if mode == "step":
        opt.step()
        refresh_weight_transposes(m)
moved = [n for n, a, p in zip(names, p0, m.parameters()) if not torch.equal(a, p.detach())]
print("det %s size %d batch %d mode %s math %s: parameters moved by the lr-0 steps: %d %s" % (torch.backends.cudnn.deterministic, size, batch, mode, math, len(moved), moved[:3]))
for it in range(1, 4):
    r, q = recs[it], recs[0]
    fw = {k: bool(torch.equal(r[k], q[k])) for k in ("loss", "cls_align", "aff_align", "logits", "maps")}
    bad = [(n, float((x - y).abs().max()), float(y.abs().max())) for n, x, y in zip(names, r["grads"], q["grads"]) if x is not None and not torch.equal(x, y)]
    print("step %d vs 0: forward equal %s; %d gradient tensors differ" % (it, fw, len(bad)))
    if not fw["maps"]:
        d = (r["maps"] - q["maps"]).abs().amax(dim=(0, 2, 3))
        print("   maps: per-layer max diff", [float("%.2e" % v) for v in d.tolist()])
    for n, d, mx in (bad[-6:] if len(bad) > 6 else bad):
        print("   %-70s max diff %.3e of %.3e" % (n, d, mx))
