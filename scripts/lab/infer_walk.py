"""GPU lab: CAM generation over scales {0.5,1,1.5,2}, list walk with one image in flight (launch_cam_images / collect) vs one
image at a time, and batches of 8.  usage: infer_walk.py [nimg]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.infer_cam import infer_cam_image, infer_cam_images, launch_cam_images
dev = "cuda:0"
torch.manual_seed(0)
m = ACR(20, "vitb_hybrid", use_pretrain=False, math=os.environ.get("ACR_MATH", "f32")).to(dev).eval()
g = torch.Generator().manual_seed(0)
img = torch.randn(1, 3, 384, 384, generator=g).to(dev)
lab = torch.zeros(1, 20); lab[0, 3] = 1; lab[0, 11] = 1
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = (0.5, 1.0, 1.5, 2.0)
def t(fn, reps):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / reps
def walk():
    pend = None
    for _ in range(n):
        c = launch_cam_images(m, img, lab, [(375, 500)], scales=sc)
        if pend is not None: pend()
        pend = c
    pend()
print("one at a time : %.1f ms/image" % (t(lambda: infer_cam_image(m, img, lab, (375, 500), scales=sc), n) * 1e3), flush=True)
print("list walk     : %.1f ms/image" % (t(walk, 2) / n * 1e3), flush=True)
imgs8, labs8 = img.repeat(8, 1, 1, 1), lab.repeat(8, 1)
print("batch 8, 4 sc : %.1f ms/image" % (t(lambda: infer_cam_images(m, imgs8, labs8, [(375, 500)] * 8, scales=sc), 2) / 8 * 1e3), flush=True)
print("batch 8, sc 1 : %.1f ms/image" % (t(lambda: infer_cam_images(m, imgs8, labs8, [(375, 500)] * 8), 3) / 8 * 1e3), flush=True)
print("scale 1 alone : %.1f ms/image" % (t(lambda: infer_cam_image(m, img, lab, (375, 500)), n) * 1e3), flush=True)
