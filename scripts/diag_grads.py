import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from conftest import load_golden, recipe_sd
from recipe import make_inputs
from oracle import acr_oracle as O
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.train import acr_loss
dev = "cuda:0"
fx = load_golden("train_hybrid_64_b2")
sd = recipe_sd("hybrid")
m = ACR(20, "vitb_hybrid", use_pretrain=False); m.load_state_dict(sd); m.to(dev).train()
img, label = make_inputs(2, 64, 20, 1)
cl, al = m.forward_mirror(img.to(dev), img.flip(-1).to(dev))
loss, t = acr_loss(cl, al, label.to(dev), 4, 125); loss.backward()
# oracle on GPU (pure torch ops, same backend libs)
sdg = {k: v.to(dev).requires_grad_(True) for k, v in sd.items()}
l2, t2 = O.train_step(sdg, O.HYBRID_BASE, img.to(dev), label.to(dev), 125); l2.backward()
# oracle on GPU in float64
sdd = {k: v.to(dev).double().requires_grad_(True) for k, v in sd.items()}
l3, t3 = O.train_step(sdd, O.HYBRID_BASE, img.to(dev).double(), label.to(dev).double(), 125); l3.backward()
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
P = dict(m.named_parameters())
print("loss", float(loss), float(l2), float(l3), float(fx["loss"]))
for k in fx:
    if k.startswith("grad:"):
        n = k[5:]; g = torch.from_numpy(fx[k]).to(dev)
        print("%-60s hip-vs-cpuref %.2e  gpuoracle-vs-cpuref %.2e  hip-vs-gpuoracle %.2e | f64: hip %.2e gpuoracle %.2e cpuref %.2e" % (
            n, rel(P[n].grad, g), rel(sdg[n].grad, g), rel(P[n].grad, sdg[n].grad), rel(P[n].grad, sdd[n].grad), rel(sdg[n].grad, sdd[n].grad), rel(g, sdd[n].grad)))
