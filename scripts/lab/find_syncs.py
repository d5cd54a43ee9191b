"""GPU lab: host-synchronising calls inside one training step (torch.cuda.set_sync_debug_mode("warn")), fp32 and bf16."""
import sys, os, warnings, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.train import MasterWeights, PolyOptimizer, train_step
dev = torch.device("cuda:0")
for mode in ("f32", "bf16"):
    torch.manual_seed(0)
    model = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False).to(dev).train()
    img = torch.randn(4, 3, 448, 448, device=dev)
    lab = torch.zeros(4, 20, device=dev); lab[:, 0] = 1
    if mode == "bf16":
        opt = MasterWeights(model, lambda ps: PolyOptimizer(ps, lr=0.05, weight_decay=5e-4, max_step=1000))
        img = img.bfloat16()
    else:
        opt = PolyOptimizer(model.parameters(), lr=0.05, weight_decay=5e-4, max_step=1000)
    for _ in range(2):
        train_step(model, opt, img, lab, 125)
    torch.cuda.synchronize()
    print("== %s: synchronising calls in one steady step ==" % mode, flush=True)
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        train_step(model, opt, img, lab, 125)
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    for x in w:
        print("  ", str(x.message)[:160], "@", x.filename.split("/")[-1], x.lineno)
    print("   total", len(w))
