"""Distribution of max |CAM(batched flips) - CAM(sequential)| over repeated runs (fp32), to tell tolerance from race."""
import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from recipe import make_inputs
from acr_wsss_amd.infer_cam import infer_cam_image
from __graft_entry__ import _recipe_model
dev = torch.device("cuda:0")
model, _ = _recipe_model(dev); model.eval()
img, _ = make_inputs(1, 96, 20, 5)
label = torch.zeros(1, 20); label[0, [2, 11]] = 1
ref, pref = infer_cam_image(model, img.to(dev), label, (50, 41), scales=(1.0, 1.5), batch_flips=False)
for it in range(12):
    a, pa = infer_cam_image(model, img.to(dev), label, (50, 41), scales=(1.0, 1.5), batch_flips=True)
    b, pb = infer_cam_image(model, img.to(dev), label, (50, 41), scales=(1.0, 1.5), batch_flips=False)
    print("run %2d: batched vs seq %.2e | seq vs first seq %.2e | patch %.2e" % (
        it, max(np.abs(a[c] - b[c]).max() for c in a), max(np.abs(b[c] - ref[c]).max() for c in b), max(np.abs(pa[c] - pb[c]).max() for c in pa)), flush=True)
