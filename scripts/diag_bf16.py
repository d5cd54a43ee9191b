import sys, os, copy, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from conftest import load_golden, recipe_sd
from recipe import make_inputs
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.train import acr_loss
from acr_wsss_amd import backbone, _lib
dev = "cuda:0"
fx = load_golden("train_hybrid_96_b1")
size, batch, ncls, alpha, seed = [int(v) for v in fx["meta"]]
img, label = make_inputs(batch, size, ncls, seed)
sd = recipe_sd("hybrid")
def run(tag, bf16, fused_gn, hip_lin, f32math=False, autocast=False):
    backbone.GroupNormAct.fused = fused_gn
    backbone.Attention.hip_linear = hip_lin
    _lib.BF16_F32MATH = f32math
    m = ACR(20, "vitb_hybrid", use_pretrain=False); m.load_state_dict(sd); m.to(dev).train()
    x = img.to(dev)
    if bf16 and not autocast:
        m = m.bfloat16(); x = x.bfloat16()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        cl, al = m.forward_mirror(x, x.flip(-1))
    loss, t = acr_loss(cl, al, label.to(dev), size // 16, alpha)
    print("%-44s loss %.4f cls1 %.4f cls2 %.4f cls_align %.5f aff_align %.5f" % (tag, float(loss), float(t["cls_loss_1"]), float(t["cls_loss_2"]), float(t["cls_align"]), float(t["aff_align"])))
print("%-44s loss %.4f cls1 %.4f cls2 %.4f cls_align %.5f aff_align %.5f" % ("golden fp32 (reference)", fx["loss"], fx["cls_loss_1"], fx["cls_loss_2"], fx["cls_align"], fx["aff_align"]))
run("fp32 hip", False, False, False)
run("bf16 stock GN/linear, f32math attn", True, False, False, True)
run("bf16 stock GN/linear, bf16 attn", True, False, False)
run("bf16 fused GN", True, True, False)
run("bf16 hip linear", True, False, True)
run("bf16 all hip", True, True, True)
run("autocast bf16 (stock GN/linear)", True, False, False, False, True)
