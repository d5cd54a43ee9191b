"""GPU lab: run-to-run bit reproducibility of one training step's loss and gradients (same weights, same inputs, one process).
The HIP kernels use no atomics; MIOpen's convolution weight gradients may.  usage: determinism.py [bf16|f32] [iters] [size]"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import json
from recipe import recipe_state_dict, make_inputs
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.train import acr_loss
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
size = int(sys.argv[3]) if len(sys.argv) > 3 else 96
dev = torch.device("cuda:0")
if os.environ.get("DET", "0") == "1":
    torch.backends.cudnn.deterministic = True          # MIOpen: no split-K-with-atomics solvers
layout = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_layout.json")))
model = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False)
model.load_state_dict(recipe_state_dict(layout, 0), strict=True)
model = model.to(dev).train()
img, label = make_inputs(1, size, 20, 3)
x = img.to(dev)
if mode == "bf16":
    model = model.bfloat16(); x = x.bfloat16()
names = ["cls_head.weight", "pretrained.model.blocks.0.attn.qkv.weight", "pretrained.model.blocks.11.mlp.fc1.weight",
         "pretrained.model.patch_embed.backbone.stem.conv.weight", "pretrained.model.blocks.5.norm1.weight"]
params = dict(model.named_parameters())
ref = None
bad = 0
for it in range(iters):
    model.zero_grad(set_to_none=True)
    cl, al = model.forward_mirror(x, x.flip(-1))
    loss, terms = acr_loss(cl, al, label.to(dev), size // 16, 125)
    loss.backward()
    sig = [loss.detach().float().item()] + [float(t.detach().float()) for t in (terms["cls_loss_1"], terms["cls_loss_2"], terms["cls_align"], terms["aff_align"])]
    sig += [params[n].grad.float().double().sum().item() for n in names if n in params]
    if ref is None:
        ref = sig
        print("reference:", ["%.9g" % v for v in sig])
    elif sig != ref:
        bad += 1
        diff = [i for i, (a, b) in enumerate(zip(sig, ref)) if a != b]
        print("iter %d differs at %s: %s" % (it, diff, ["%.9g" % sig[i] for i in diff]))
print("%s %dx%d: %d of %d repeats differ from the first" % (mode, size, size, bad, iters - 1))
