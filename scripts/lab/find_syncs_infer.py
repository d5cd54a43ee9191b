"""GPU lab: host-synchronising calls inside infer_cam_image (torch sync debug mode)."""
import sys, os, warnings, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.infer_cam import infer_cam_image
dev = "cuda:0"
torch.manual_seed(0)
m = ACR(20, "vitb_hybrid", use_pretrain=False).to(dev).eval()
img = torch.randn(1, 3, 384, 384, device=dev)
lab = torch.zeros(1, 20); lab[0, 3] = 1; lab[0, 11] = 1
for _ in range(2):
    infer_cam_image(m, img, lab, (375, 500))
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    infer_cam_image(m, img, lab, (375, 500))
torch.cuda.set_sync_debug_mode("default")
import collections
c = collections.Counter((str(x.message)[:90], os.path.basename(x.filename), x.lineno) for x in w)
for k, n in c.most_common(20):
    print(n, k)
