import sys, os, traceback, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
from conftest import recipe_sd
from recipe import make_inputs
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd import backbone
m = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False)
m.load_state_dict(recipe_sd("hybrid"), strict=True)
m = m.to("cuda:0").eval()
for p in m.parameters(): p.requires_grad_(False)
vit = m.pretrained.model
for a in vit.blocks: a.attn.keep_state_in_training = True
m.truncate_at = 10
for math in ("f32", "f32_split"):
    m.set_math(math)
    for s in (48, 96, 144, 192):
        inp = torch.randn(2, 3, s, s, device="cuda:0")
        vit.graph_prefix = False
        try:
            g = backbone.PassGraph(m, inp, 10, "grad")
            print(math, s, "captured ok")
        except Exception:
            print(math, s, "FAILED")
            traceback.print_exc()
            torch.cuda.synchronize()
