"""smoke(): one tiny ACR training step and one CAM read-out on cuda:0, checked against the CPU oracle.
(Imports ``oracle`` -- allowed only here, in tests/ and in bench.py's cpu_baseline leg.)"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _recipe_model(dev):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from recipe import recipe_state_dict
    from .DPT.ACR import ACR
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_layout.json")) as f:
        layout = json.load(f)
    sd = recipe_state_dict(layout, 0)
    model = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False)
    model.load_state_dict(sd, strict=True)
    return model.to(dev), sd


def run_smoke():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)                                # `oracle` lives at the repo root
    from .train import acr_loss
    from .infer_cam import infer_cam_image
    from oracle import acr_oracle as O
    dev = torch.device("cuda:0")
    model, sd = _recipe_model(dev)
    from recipe import make_inputs
    img, label = make_inputs(2, 64, 20, 1)
    model.train()
    cls_list, attn_list = model.forward_mirror(img.to(dev), img.flip(-1).to(dev))
    loss, terms = acr_loss(cls_list, attn_list, label.to(dev), 4, 125)
    loss.backward()
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_loss, ref_terms = O.train_step(sdg, O.HYBRID_BASE, img, label, 125)
    ref_loss.backward()
    for k in ("loss", "cls_align", "aff_align"):
        a, b = float(terms[k]), float(ref_terms[k])
        assert abs(a - b) <= 2e-4 * abs(b), (k, a, b)
    g = model.cls_head.weight.grad.cpu()
    gr = sdg["cls_head.weight"].grad
    assert (g - gr).abs().max() <= 1e-3 * gr.abs().max()
    lab = torch.zeros(1, 20)
    lab[0, 3] = 1
    cam_dict, _ = infer_cam_image(model, img[:1].to(dev), lab, (50, 70))
    ref_dict, _, _ = O.infer_image(sd, O.HYBRID_BASE, img[:1], lab, (50, 70))
    assert np.abs(cam_dict[3] - ref_dict[3]).max() <= 2e-3
    print("smoke ok: loss %.6f (oracle %.6f)" % (float(loss), float(ref_loss)))
