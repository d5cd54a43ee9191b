"""GPU lab: which hooked kernel shapes one fp32 / bf16 training step launches (ops.KernelTimer keys), with their in-step times."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd import ops
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.train import PolyOptimizer, train_step
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False).to(dev).train()
opt = PolyOptimizer(model.parameters(), lr=0.05, weight_decay=5e-4, max_step=1000)
img = torch.randn(16, 3, 448, 448, device=dev)
lab = torch.zeros(16, 20, device=dev); lab[:, 0] = 1
for _ in range(2):
    train_step(model, opt, img, lab, 125)
t = ops.KernelTimer()
ops.KERNEL_TIMER = t
for _ in range(3):
    t.next_step()
    train_step(model, opt, img, lab, 125)
ops.KERNEL_TIMER = None
for k, v in sorted(t.collect().items()):
    print("%-40s %.3f ms" % (k, sum(v) / len(v)))
