"""GETAM / CAM generation -- the per-image body of the reference's infer_cam.py:128-249 on the HIP path.

Differences from the reference that do not change results:
  * ``scales`` is a real list (the reference's loop is ``for scale in [1]`` with the scale variable already
    threaded through all shape arithmetic, infer_cam.py:145-156,186; BASELINE.json names {0.5,1,1.5,2});
  * the backward per class stops at block ``start_layer`` (nothing below it is read by getam, DPT/ACR.py:207)
    and computes no weight gradients (``torch.autograd.grad`` w.r.t. the tokens entering that block);
  * patch-CAM / GETAM read-outs, the affinity product and both bilinear resizes run as HIP kernels and stay
    on the device; one D2H copy per image instead of one per class (infer_cam.py:188).
Rank r of ``world`` processes handles images r::world (embarrassingly parallel, no collective).
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import ops


def infer_cam_image(model, img, label, out_hw, start_layer=10, func="grad", aff=True, scales=(1,), truncate=True,
                    batch_flips=True):
    """CAMs of one image.  img (1,3,h,w) normalised fp32 on the GPU; label (1,C) multi-hot; out_hw = (W,H) =
    (image height, image width) as infer_cam.py:138 names them.  Returns (cam_dict, patch_cam_dict):
    {class index: float32 (W,H) numpy array}, min-max normalised over the summed passes (:201-215)."""
    W, H = out_hw
    dev = img.device
    C = label.shape[1]
    label = label.to(dev).float()
    classes = [c for c in range(C) if float(label[0, c]) > 1e-5]
    b, _, h, w = img.shape
    assert b == 1, "infer_cam processes one image per step (infer_cam.py:123 chunker(...,1))"
    vit = model.pretrained.model
    cam_acc = torch.zeros((len(classes), W, H), dtype=torch.float32, device=dev)
    patch_acc = torch.zeros((C, W, H), dtype=torch.float32, device=dev)
    old_trunc = model.truncate_at
    model.truncate_at = start_layer if truncate else None
    try:
        for scale in scales:
            base = F.interpolate(img, size=(int(h * scale), int(w * scale)), mode="bilinear", align_corners=False)
            ph, pw = int((h * scale) // 16), int((w * scale) // 16)
            # the two passes of a scale (h-flipped first, then plain: infer_cam.py:147-153) are independent samples: run
            # them as ONE batch of 2 -- every per-sample operator (GroupNorm, LayerNorm, attention) is batch-invariant --
            # and back-propagate both class logits at once; at batch 1 the step is launch-bound, not GPU-bound
            passes = ((True, False),) if batch_flips else ((True,), (False,))
            for flips in passes:
                inp = torch.cat([base.flip(-1) if f else base for f in flips], dim=0)
                with torch.enable_grad():
                    cls_pred, _, attn, patch_cam = model.forward_cam(inp)
                    for i, flipped in enumerate(flips):
                        # patch-token CAM: (1,N,C) -> (C,ph,pw) -> bilinear(align_corners=False) * label, un-flip, sum
                        ops.bilinear_resize(patch_cam[i].detach().float().reshape(ph, pw, C), (W, H), False,
                                            chan_mul=label[0], hflip=flipped, out=patch_acc, channels_last=True)
                    rows = [[] for _ in flips]
                    for c in classes:
                        tgt = cls_pred[:, c].sum()                  # samples are independent: one backward serves both
                        if truncate:
                            torch.autograd.grad(tgt, vit.trunc_input, retain_graph=True)
                        else:
                            model.zero_grad()
                            tgt.backward(retain_graph=True)
                        for i in range(len(flips)):
                            cam, _, _ = model.getam(i, start_layer=start_layer, func=func)
                            rows[i].append(cam)
                for i, flipped in enumerate(flips):
                    cams = torch.cat(rows[i], dim=0).contiguous()                    # (n_cls, N)
                    if aff:
                        cams = ops.aff_refine(attn[i].detach().contiguous(), cams)    # patch_aff @ cam (:164-165,183-184)
                    ops.bilinear_resize(cams.reshape(len(classes), ph, pw), (W, H), True, hflip=flipped, out=cam_acc)
    finally:
        model.truncate_at = old_trunc
    cmin, cmax = cam_acc.amin((1, 2), keepdim=True), cam_acc.amax((1, 2), keepdim=True)
    norm_cam = ((cam_acc - cmin) / (cmax - cmin + 1e-6)).cpu().numpy()
    pmin, pmax = patch_acc.amin((1, 2), keepdim=True), patch_acc.amax((1, 2), keepdim=True)
    patch_norm = ((patch_acc - pmin) / (pmax - pmin + 1e-5)).cpu().numpy()
    cam_dict = {c: norm_cam[i] for i, c in enumerate(classes)}
    patch_dict = {c: patch_norm[c] for c in classes}
    return cam_dict, patch_dict


def seeds_from_cam_dict(cam_dict, threshold, num_cls=21):
    """evaluation.py:27-33: background plane = threshold, argmax over (21,h,w) -> uint8 seed map."""
    h, w = next(iter(cam_dict.values())).shape
    tensor = np.zeros((num_cls, h, w), np.float32)
    for k, v in cam_dict.items():
        tensor[k + 1] = v
    tensor[0] = threshold
    return np.argmax(tensor, axis=0).astype(np.uint8)


def infer_cam_list(model, items, out_cam=None, rank=0, world=1, **kw):
    """Shard ``items`` -- an indexable of (name, img (1,3,h,w), label (1,C), (W,H)) -- over ranks and write
    ``<out_cam>/<name>.npy`` in the reference's wire format: a pickled {class: float32 (W,H)} dict
    (infer_cam.py:227-228, read back by evaluation.py:23-25).  Returns {name: cam_dict} of this rank."""
    dev = next(model.parameters()).device
    model.eval()
    results = {}
    for i in range(rank, len(items), world):
        name, img, label, out_hw = items[i]
        cam_dict, _ = infer_cam_image(model, img.to(dev), label, out_hw, **kw)
        if out_cam is not None:
            os.makedirs(out_cam, exist_ok=True)
            np.save(os.path.join(out_cam, name + ".npy"), cam_dict)
        results[name] = cam_dict
    return results
