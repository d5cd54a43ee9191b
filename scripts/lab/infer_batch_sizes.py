"""GPU lab: infer_cam_list over 48 same-sized images (384^2 base, scales {0.5,1,1.5,2}, 2 classes, split products) at batch sizes 4 / 8 / 12 / 16."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.infer_cam import infer_cam_list
dev = "cuda:0"
torch.manual_seed(0)
m = ACR(20, "vitb_hybrid", use_pretrain=False, math="f32_split").to(dev).eval()
g = torch.Generator().manual_seed(0)
img = torch.randn(1, 3, 384, 384, generator=g).to(dev)
lab = torch.zeros(1, 20); lab[0, 3] = 1; lab[0, 11] = 1
items = [("img%03d" % i, img, lab, (375, 500)) for i in range(48)]
for bs in (4, 8, 12, 16):
    infer_cam_list(m, items[:2 * bs], scales=(0.5, 1.0, 1.5, 2.0), batch_size=bs)
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); t0 = time.time()
    infer_cam_list(m, items, scales=(0.5, 1.0, 1.5, 2.0), batch_size=bs)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("batch %2d: %.1f ms/image = %.1f img/s, peak %.1f GB" % (bs, dt / 48 * 1e3, 48 / dt, torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)
    m.pretrained.model.__dict__.pop("_pass_graphs", None); m.pretrained.model.__dict__.pop("_graph_sightings", None)
    torch.cuda.empty_cache()
