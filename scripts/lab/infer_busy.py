"""GPU lab: CAM generation over scales {0.5,1,1.5,2} -- wall time per image (run under rocprofv3 --stats to compare with the
summed kernel time: how launch-bound is the path?).  usage: infer_busy.py [nimg] [scale | all]  (one scale, or the four together, only: for a kernel trace)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.infer_cam import infer_cam_image
dev = "cuda:0"
torch.manual_seed(0)
m = ACR(20, "vitb_hybrid", use_pretrain=False, math=os.environ.get("ACR_MATH", "f32")).to(dev).eval()
g = torch.Generator().manual_seed(0)
img = torch.randn(1, 3, 384, 384, generator=g).to(dev)
lab = torch.zeros(1, 20); lab[0, 3] = 1; lab[0, 11] = 1
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sets = ((0.5, 1.0, 1.5, 2.0), (0.5,), (1.0,), (1.5,), (2.0,)) if len(sys.argv) < 3 else (((0.5, 1.0, 1.5, 2.0),) if sys.argv[2] == "all" else ((float(sys.argv[2]),),))
for scales in sets:
    infer_cam_image(m, img, lab, (375, 500), scales=scales)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        infer_cam_image(m, img, lab, (375, 500), scales=scales)
    torch.cuda.synchronize()
    print("scales %-22s wall %.1f ms/image over %d images (+1 warm-up)" % (scales, (time.time() - t0) / n * 1e3, n), flush=True)
