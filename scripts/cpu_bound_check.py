"""How much host time does one training step need to ENQUEUE its kernels (vs the GPU time of the step)?"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.train import MasterWeights, PolyOptimizer, train_step
dev = "cuda:0"
torch.manual_seed(0)
m = ACR(20, "vitb_hybrid", use_pretrain=False).to(dev).train()
opt = MasterWeights(m, lambda ps: PolyOptimizer(ps, lr=0.05, weight_decay=5e-4, max_step=1000))
g = torch.Generator().manual_seed(0)
B, S = int(os.environ.get("B", 16)), int(os.environ.get("S", 448))
img = torch.randn(B, 3, S, S, generator=g).to(dev).bfloat16()
lab = (torch.rand(B, 20, generator=g) > 0.85).float().to(dev); lab[:, 0] = 1
for _ in range(3): train_step(m, opt, img, lab, 125)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train_step(m, opt, img, lab, 125)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print("enqueue %.1f ms   step (sync) %.1f ms" % (1e3 * sum(enq) / len(enq), 1e3 * sum(tot) / len(tot)))
