"""CAM-generation throughput (BASELINE configs[3], informational): infer_cam_image at scales {0.5,1,1.5,2} of 384^2,
2 positive classes per image, fp32 (parity precision), one GPU."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.infer_cam import infer_cam_image
dev = "cuda:0"
torch.manual_seed(0)
m = ACR(20, "vitb_hybrid", use_pretrain=False).to(dev).eval()
dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float32
m = m.to(dt)
g = torch.Generator().manual_seed(0)
img = torch.randn(1, 3, 384, 384, generator=g).to(dev).to(dt)
lab = torch.zeros(1, 20); lab[0, 3] = 1; lab[0, 11] = 1
for scales in ((1,), (0.5, 1.0, 1.5, 2.0)):
    for _ in range(2):
        infer_cam_image(m, img, lab, (375, 500), scales=scales)
    torch.cuda.synchronize(); t0 = time.time(); n = 5
    for _ in range(n):
        cam, _ = infer_cam_image(m, img, lab, (375, 500), scales=scales)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print("scales %-22s %.1f ms/image  %.2f img/s  (cam %s, peak mem %.1f GB)" % (scales, dt * 1e3, 1 / dt, cam[3].shape, torch.cuda.max_memory_allocated() / 2**30))

from acr_wsss_amd.infer_cam import infer_cam_images
for B in (4, 8):
    imgs = img.repeat(B, 1, 1, 1); labs = lab.repeat(B, 1); sizes = [(375, 500)] * B
    for _ in range(2): infer_cam_images(m, imgs, labs, sizes)
    torch.cuda.synchronize(); t0 = time.time(); n = 3
    for _ in range(n): infer_cam_images(m, imgs, labs, sizes)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print("batch %d, scale 1: %.1f ms/batch  %.1f img/s  (peak mem %.1f GB)" % (B, dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 2**30))

# the optional CRF stage of infer_cam.py:218-225 on the GPU: both alphas for the 2-class image above
import numpy as np
from acr_wsss_amd.crf import crf_with_alpha
orig = np.random.default_rng(0).integers(0, 256, (375, 500, 3)).astype(np.uint8)
cam, _ = infer_cam_image(m, img, lab, (375, 500))
for _ in range(2):
    crf_with_alpha(cam, 1, orig)
torch.cuda.synchronize(); t0 = time.time(); n = 5
for _ in range(n):
    for alpha in (1, 12):
        crf_with_alpha(cam, alpha, orig)
torch.cuda.synchronize()
print("CRF (2 classes + background, alphas 1 and 12, 375x500): %.1f ms/image" % ((time.time() - t0) / n * 1e3))
