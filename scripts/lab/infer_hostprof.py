"""GPU lab: host-side profile (cProfile) of infer_cam_images at batch 8, scale 1."""
import sys, os, time, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.infer_cam import infer_cam_images
dev = "cuda:0"
torch.manual_seed(0)
m = ACR(20, "vitb_hybrid", use_pretrain=False).to(dev).eval()
g = torch.Generator().manual_seed(0)
img = torch.randn(1, 3, 384, 384, generator=g).to(dev)
lab = torch.zeros(1, 20); lab[0, 3] = 1; lab[0, 11] = 1
imgs8, labs8 = img.repeat(8, 1, 1, 1), lab.repeat(8, 1)
f = lambda: infer_cam_images(m, imgs8, labs8, [(375, 500)] * 8)
f(); f(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5): f()
torch.cuda.synchronize(); print("%.1f ms per batch of 8" % ((time.time() - t0) / 5 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): f()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
