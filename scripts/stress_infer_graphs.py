"""GPU stress: CAM generation over more input geometries than VisionTransformer.max_prefix_graphs (LRU eviction of captured
prefixes), graphs + streams on vs eager one-stream launches: results must stay bit-identical and memory bounded."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.infer_cam import infer_cam_image
dev = "cuda:0"
torch.backends.cudnn.deterministic = True      # MIOpen's default fp32 solvers for the strided stem convolutions are not run-to-run
torch.manual_seed(0)                            # deterministic (scripts/lab/determinism_infer.py); the hand-written kernels are
m = ACR(20, "vitb_hybrid", use_pretrain=False).to(dev).eval()
vit = m.pretrained.model
lab = torch.zeros(1, 20); lab[0, 3] = 1; lab[0, 11] = 1
g = torch.Generator().manual_seed(1)
# multiples of 32, so that the 1.5x pass is a multiple of the 16-pixel patch too (as in the reference: forward_flex sizes the
# position embedding with h // 16 while the SAME-padded stem yields ceil(h / 16) rows)
sizes = [(96, 96), (96, 128), (128, 96), (128, 128), (160, 96), (96, 160), (160, 160), (160, 128), (128, 160), (192, 192), (192, 96), (96, 192)]
peak = []
for rnd in range(3):
    for (h, w) in sizes:
        img = torch.randn(1, 3, h, w, generator=g).to(dev)
        vit.graph_prefix = False; vit.graph_pass = False
        if rnd == 0:                                        # first touch of a geometry: MIOpen picks its solvers here
            infer_cam_image(m, img, lab, (50, 60), scales=(1.0, 1.5), concurrent_scales=False)
        ref = infer_cam_image(m, img, lab, (50, 60), scales=(1.0, 1.5), concurrent_scales=False)
        ref2 = infer_cam_image(m, img, lab, (50, 60), scales=(1.0, 1.5), concurrent_scales=False)
        vit.graph_prefix = True; vit.graph_pass = True
        got = infer_cam_image(m, img, lab, (50, 60), scales=(1.0, 1.5), concurrent_scales=True)
        for c in (3, 11):
            e2 = float(np.abs(ref[0][c] - ref2[0][c]).max())
            eg = float(np.abs(ref[0][c] - got[0][c]).max())
            if e2 or eg:
                print("  %dx%d class %d: eager vs eager %.3e, eager vs graph %.3e" % (h, w, c, e2, eg), flush=True)
            assert eg <= max(e2, 0.0) + 1e-6, (rnd, h, w, c, e2, eg)
    torch.cuda.synchronize()
    peak.append((len(vit.__dict__.get("_pass_graphs", ())) + len(vit.__dict__.get("_prefix_graphs", ())), torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30))
    print("round %d: %d graphs cached, allocated %.2f GB, reserved %.2f GB" % ((rnd,) + peak[-1]), flush=True)
assert peak[-1][0] <= vit.max_prefix_graphs and peak[-1][2] <= peak[0][2] * 1.5 + 0.5, peak
print("ok")
