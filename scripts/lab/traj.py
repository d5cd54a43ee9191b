"""GPU lab: fp32 vs bf16 loss trajectories for a few learning rates (which setting gives a non-chaotic fp32 curve?)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.train import MasterWeights, PolyOptimizer, train_step
DEV = "cuda:0"
size, B = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator(device="cpu").manual_seed(1000)
img = torch.randn(B, 3, size, size, generator=g).to(DEV)
label = (torch.rand(B, 20, generator=g) > 0.85).float(); label[:, 0] = 1.0; label = label.to(DEV)
for lr in (0.05, 0.01, 0.002):
    curves = {}
    for mode in ("f32", "bf16"):
        torch.manual_seed(0)
        m = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False).to(DEV).train()
        if mode == "bf16":
            opt = MasterWeights(m, lambda ps: PolyOptimizer(ps, lr=lr, weight_decay=5e-4, max_step=100000)); x = img.bfloat16()
        else:
            opt = PolyOptimizer(m.parameters(), lr=lr, weight_decay=5e-4, max_step=100000); x = img
        curves[mode] = np.array([float(train_step(m, opt, x, label, 125)[0]) for _ in range(20)])
        del m, opt; torch.cuda.empty_cache()
    rel = np.abs(curves["bf16"] - curves["f32"]) / np.abs(curves["f32"])
    print("lr %g size %d B %d\n  f32 %s\n  bf16 %s\n  max rel %.3e" % (lr, size, B, np.round(curves["f32"], 3), np.round(curves["bf16"], 3), rel.max()), flush=True)
