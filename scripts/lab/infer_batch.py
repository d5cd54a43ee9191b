"""GPU lab: CAM generation over scales {0.5,1,1.5,2} for batches of same-sized images (infer_cam_images): img/s by batch size."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.infer_cam import infer_cam_images
dev = "cuda:0"
torch.manual_seed(0)
m = ACR(20, "vitb_hybrid", use_pretrain=False).to(dev).eval()
g = torch.Generator().manual_seed(0)
scales = (0.5, 1.0, 1.5, 2.0)
for B in (1, 2, 4, 8):
    imgs = torch.randn(B, 3, 384, 384, generator=g).to(dev)
    lab = torch.zeros(B, 20); lab[:, 3] = 1; lab[:, 11] = 1
    sizes = [(375, 500)] * B
    infer_cam_images(m, imgs, lab, sizes, scales=scales)
    torch.cuda.synchronize(); t0 = time.time()
    n = 3
    for _ in range(n):
        infer_cam_images(m, imgs, lab, sizes, scales=scales)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print("batch %d: %.1f ms per batch, %.1f img/s, peak memory %.1f GB" % (B, dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 2**30), flush=True)
