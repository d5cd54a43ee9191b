"""GPU lab: names of the parameters whose gradient is not bit-reproducible over repeats (deterministic MIOpen solvers)."""
import sys, os, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from recipe import recipe_state_dict, make_inputs
from acr_wsss_amd.DPT.ACR import ACR
from acr_wsss_amd.train import acr_loss
torch.backends.cudnn.deterministic = True
dev = torch.device("cuda:0")
layout = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_layout.json")))
for dtype in (torch.bfloat16, torch.float32):
    model = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False)
    model.load_state_dict(recipe_state_dict(layout, 0), strict=True)
    model = model.to(dev).to(dtype).train()
    img, label = make_inputs(2, 96, 20, 17)
    x = img.to(dev).to(dtype)
    first = None
    bad = {}
    for it in range(8):
        model.zero_grad(set_to_none=True)
        cl, al = model.forward_mirror(x, x.flip(-1))
        loss, _ = acr_loss(cl, al, label.to(dev), 6, 125)
        loss.backward()
        g = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        if first is None:
            first = g
        else:
            for n in g:
                if not torch.equal(g[n], first[n]):
                    bad[n] = max(bad.get(n, 0.0), float((g[n].float() - first[n].float()).abs().max() / (first[n].float().abs().max() + 1e-30)))
    print(dtype, "parameters with non-reproducible gradients:", len(bad))
    for n, v in sorted(bad.items())[:40]:
        print("   %-70s shape %-18s max rel diff %.2e" % (n, tuple(first[n].shape), v))
