"""GPU lab: the CAM list walk the bench's `infer.value` measures -- acr_wsss_amd.infer_cam.infer_cam_list with its defaults (batches of 8
same-sized images, scales {0.5, 1, 1.5, 2}, split products) -- for a kernel trace.  usage: infer_list_busy.py [batches]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.infer_cam import infer_cam_list
dev = "cuda:0"
torch.manual_seed(0)
m = ACR(20, "vitb_hybrid", use_pretrain=False, math=os.environ.get("ACR_MATH", "f32_split")).to(dev).eval()
g = torch.Generator().manual_seed(0)
img = torch.randn(1, 3, 384, 384, generator=g).to(dev)
lab = torch.zeros(1, 20); lab[0, 3] = 1; lab[0, 11] = 1
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
items = [("img%03d" % i, img, lab, (375, 500)) for i in range(8 * nb)]
infer_cam_list(m, items[:16], scales=(0.5, 1.0, 1.5, 2.0))          # first sighting eager, second captures the pass graphs
torch.cuda.synchronize(); t0 = time.time()
infer_cam_list(m, items, scales=(0.5, 1.0, 1.5, 2.0))
torch.cuda.synchronize()
dt = time.time() - t0
print("infer_cam_list, %d images in batches of 8, 4 scales: %.1f ms/image = %.1f img/s" % (len(items), dt / len(items) * 1e3, len(items) / dt), flush=True)
