"""GPU lab: which stage of an (eager) CAM pass is not run-to-run deterministic?  Two passes on the same input, bitwise compare."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
m = ACR(20, "vitb_hybrid", use_pretrain=False).to(dev).eval()
vit = m.pretrained.model
vit.graph_prefix = False; vit.graph_pass = False
for p in m.parameters():
    p.requires_grad_(False)
for blk in vit.blocks:
    blk.attn.keep_state_in_training = True
g = torch.Generator().manual_seed(1)
for (h, w) in ((96, 96), (144, 144), (384, 384)):
    img = torch.randn(2, 3, h, w, generator=g).to(dev)
    outs = []
    for rep in range(3):
        m.truncate_at = 10
        with torch.enable_grad():
            cls_pred, xp, attn, patch_cam = m.forward_cam(img)
            taps = {k: v.detach().clone() for k, v in m.pretrained.activations.items()}
            tgt = cls_pred[0, 3] + cls_pred[1, 3]
            (gin,) = torch.autograd.grad(tgt, vit.trunc_input, retain_graph=True)
            cam, _, rows = m.getam(0, start_layer=10, func="grad")
        outs.append(dict(cls_pred=cls_pred.detach().clone(), patch_cam=patch_cam.detach().clone(), attn=attn.detach().clone(),
                         gin=gin.clone(), cam=cam.clone(), **{"tap" + k: v for k, v in taps.items()}))
    for k in outs[0]:
        same = all(torch.equal(outs[0][k], o[k]) for o in outs[1:])
        d = max(float((outs[0][k].float() - o[k].float()).abs().max()) for o in outs[1:])
        print("%dx%d %-10s %s  max diff %.3e (max |x| %.3e)" % (h, w, k, "bit-identical" if same else "DIFFERS", d, float(outs[0][k].abs().max())), flush=True)
