#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the *reference itself* on CPU.

Run in the build container only (``/root/reference`` does not travel to the GPU box):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz + state_dict_layout.json

What is real reference code and what is restated here
-----------------------------------------------------
* ``DPT.ACR.ACR`` (and everything below it: ``DPT/blocks.py``, ``DPT/vit.py``, ``models/*``) is imported
  unmodified from ``/root/reference``.  The only shim is a constants-only ``timm`` package written to a
  temp dir (``timm`` is not installed; ``models/__init__.py`` star-imports families that want
  ``timm.data`` constants and ``timm.models.*`` re-exports) -- see SURVEY.md 8(c).
* ``train_acr.py`` / ``infer_cam.py`` / ``tool/torchutils.py`` cannot be imported (cv2, torchvision,
  pydensecrf at module top; hot code inline behind NCCL/CUDA calls).  Their hot blocks are restated
  below line-for-line *as a harness around the real model*:
    - train_step()      <- train_acr.py:135-174  (img.flip(-1) == RandomHorizontalFlip(p=1))
    - PolySGD           <- tool/torchutils.py:10-31 (including the positional-momentum quirk)
    - infer_one_image() <- infer_cam.py:141-215
    - seeds()           <- evaluation.py:27-33
Weights come from tests/golden/recipe.py (use_pretrain=False always; no network access is attempted).
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("ACR_REFERENCE", "/root/reference")
sys.path.insert(0, HERE)
from recipe import fill_state_dict, make_inputs, weights_checksum  # noqa: E402


def _install_timm_stub():
    d = tempfile.mkdtemp(prefix="timm_stub_")
    os.makedirs(os.path.join(d, "timm", "data"))
    os.makedirs(os.path.join(d, "timm", "models"))
    with open(os.path.join(d, "timm", "__init__.py"), "w") as f:
        f.write("__version__ = '0.4.5'\n")
    with open(os.path.join(d, "timm", "data", "__init__.py"), "w") as f:
        f.write(
            "IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)\nIMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)\n"
            "IMAGENET_INCEPTION_MEAN = (0.5, 0.5, 0.5)\nIMAGENET_INCEPTION_STD = (0.5, 0.5, 0.5)\n"
            "IMAGENET_DPN_MEAN = (124 / 255, 117 / 255, 104 / 255)\nIMAGENET_DPN_STD = tuple([1 / (.0167 * 255)] * 3)\n")
    with open(os.path.join(d, "timm", "models", "__init__.py"), "w") as f:
        f.write("")
    for sub in ("helpers", "layers", "registry", "vision_transformer"):
        with open(os.path.join(d, "timm", "models", sub + ".py"), "w") as f:
            f.write("import importlib\n_m = importlib.import_module('models.%s')\n"
                    "def __getattr__(name):\n    return getattr(_m, name)\n" % sub)
    sys.path.insert(0, d)
    sys.path.insert(0, REF)


# ----------------------------------------------------------------------------------------------
# restated harness blocks
# ----------------------------------------------------------------------------------------------
class PolySGD(torch.optim.SGD):
    """tool/torchutils.py:10-31.  NB: SGD(params, lr, weight_decay) passes weight_decay positionally
    into the *momentum* slot -- reproduced on purpose."""

    def __init__(self, params, lr, weight_decay, max_step, momentum=0.9):
        super().__init__(params, lr, weight_decay)
        self.global_step = 0
        self.max_step = max_step
        self.momentum = momentum
        self._initial_lr = [g["lr"] for g in self.param_groups]

    def step(self, closure=None):
        if self.global_step < self.max_step:
            mult = (1 - self.global_step / self.max_step) ** self.momentum
            for i, g in enumerate(self.param_groups):
                g["lr"] = self._initial_lr[i] * mult
        super().step(closure)
        self.global_step += 1


def train_step(model, img, label, alpha, forward_mirror):
    """train_acr.py:135-168 verbatim semantics (in-place block flips on the view-2 maps)."""
    b, c, h, w = img.shape
    img2 = img.flip(-1)
    cls_list, attn_list = forward_mirror(img, img2)
    attn1, attn2 = attn_list[0], attn_list[1]
    raw1, raw2 = attn1.detach().clone(), attn2.detach().clone()
    x1, x2 = cls_list[0], cls_list[1]
    attn1_cls = attn1[:, :, 0, 1:].unsqueeze(2)
    attn2_cls = attn2[:, :, 0, 1:].unsqueeze(2)
    attn1_aff = attn1[:, :, 1:, 1:]
    attn2_aff = attn2[:, :, 1:, 1:]
    p = h // 16
    for i in range(p):
        attn2_cls[:, :, :, i * p:i * p + p] = attn2_cls[:, :, :, i * p:i * p + p].flip(3)
    for i in range(p):
        attn2_aff[:, :, i * p:i * p + p, :] = attn2_aff[:, :, i * p:i * p + p, :].flip(2)
    for i in range(p):
        attn2_aff[:, :, :, i * p:i * p + p] = attn2_aff[:, :, :, i * p:i * p + p].flip(3)
    cls_align = F.l1_loss(attn1_cls, attn2_cls, reduction="mean")
    aff_align = F.l1_loss(attn1_aff, attn2_aff, reduction="mean")
    cls1 = F.multilabel_soft_margin_loss(x1, label)
    cls2 = F.multilabel_soft_margin_loss(x2, label)
    loss = cls1 + cls2 + cls_align * alpha + aff_align * alpha
    out = dict(x_cls_1=x1, x_cls_2=x2, x_p_cls_1=cls_list[2], x_p_cls_2=cls_list[3],
               attn1=raw1, attn2=raw2, cls_align=cls_align, aff_align=aff_align,
               cls_loss_1=cls1, cls_loss_2=cls2, loss=loss)
    return loss, out


def seeds(cam_dict, h, w, t, num_cls=21):
    """evaluation.py:27-33: background plane = threshold, argmax -> uint8."""
    tensor = np.zeros((num_cls, h, w), np.float32)
    for k, v in cam_dict.items():
        tensor[k + 1] = v
    tensor[0] = t
    return np.argmax(tensor, axis=0).astype(np.uint8)


def infer_one_image(model, img, label, WH, start_layer, func, aff, scales=(1,), num_classes=20):
    """infer_cam.py:141-215 (W = image height, H = image width as in :138)."""
    W, H = WH
    cam_list, patch_cam_list = [], []
    b, c, h, w = img.shape
    getam_rows = []
    for scale in scales:
        for hflip in [1, 2]:
            cam_matrix = torch.zeros((b, num_classes, W, H))
            model.zero_grad()
            inp = F.interpolate(img, size=(int(h * scale), int(w * scale)), mode="bilinear", align_corners=False)
            if hflip % 2 == 1:
                inp = inp.flip(-1)
            cls_pred, _, attn, patch_cam = model.forward_cam(inp)
            ph, pw = int((h * scale) // 16), int((w * scale) // 16)
            patch_cam = patch_cam.permute(0, 2, 1).reshape(1, num_classes, ph, pw)
            patch_cam = F.interpolate(patch_cam, [W, H], mode="bilinear", align_corners=False)[0]
            patch_cam = patch_cam.detach().cpu().numpy() * label[0, :].cpu().clone().view(num_classes, 1, 1).numpy()
            if hflip % 2 == 1:
                patch_cam = np.flip(patch_cam, axis=-1)
            patch_cam_list.append(patch_cam)
            patch_aff = torch.sum(attn[:, :, 1:, 1:], dim=1)
            cur_label = label[0, :]
            output = cls_pred[0, :]
            for class_index in range(num_classes):
                if cur_label[class_index] > 1e-5:
                    one_hot = np.zeros((1, output.size()[-1]), dtype=np.float32)
                    one_hot[0, class_index] = 1
                    one_hot = torch.from_numpy(one_hot).requires_grad_(True)
                    one_hot = torch.sum(one_hot * output)
                    model.zero_grad()
                    one_hot.backward(retain_graph=True)
                    cam, _, _ = model.getam(0, start_layer=start_layer, func=func)
                    getam_rows.append(cam.detach().clone().numpy())
                    if aff:
                        cam = torch.matmul(patch_aff, cam.unsqueeze(2))
                    cam = cam.reshape(ph, pw)
                    cam = F.interpolate(cam.unsqueeze(0).unsqueeze(0), (W, H), mode="bilinear", align_corners=True)
                    cam_matrix[0, class_index, :, :] = cam
            cam_up_single = cam_matrix[0].cpu().data.numpy()
            if hflip % 2 == 1:
                cam_up_single = np.flip(cam_up_single, axis=2)
            cam_list.append(cam_up_single)
    patch_sum = np.sum(patch_cam_list, axis=0)
    pmin, pmax = np.min(patch_sum, (1, 2), keepdims=True), np.max(patch_sum, (1, 2), keepdims=True)
    patch_norm = (patch_sum - pmin) / (pmax - pmin + 1e-5)
    sum_cam = np.sum(cam_list, axis=0)
    cmin, cmax = np.min(sum_cam, (1, 2), keepdims=True), np.max(sum_cam, (1, 2), keepdims=True)
    norm_cam = (sum_cam - cmin) / (cmax - cmin + 1e-6)
    cam_dict = {c: norm_cam[c] for c in range(num_classes) if label[0, c] > 1e-5}
    patch_dict = {c: patch_norm[c] for c in range(num_classes) if label[0, c] > 1e-5}
    rows_out = np.stack(getam_rows) if len({r.shape for r in getam_rows}) == 1 else getam_rows
    return cam_dict, patch_dict, rows_out


# ----------------------------------------------------------------------------------------------
# assembled ViT-tiny oracle (BASELINE config 1; SURVEY 8(c): no backbone_dict entry, cls_head is
# hard-wired to 768 in ACR.py:88, so it is assembled from reference parts)
# ----------------------------------------------------------------------------------------------
def build_tiny(num_classes):
    from models.factory import create_model
    from DPT.vit import _make_vit_b16_backbone
    from DPT.ACR import DPT

    vit = create_model("vit_deit_tiny_patch16_224", pretrained=False)
    m = nn.Module()
    m.pretrained = _make_vit_b16_backbone(vit, features=[48, 96, 192, 192], hooks=[2, 5, 8, 11],
                                          vit_features=192, seg=False)
    m.cls_head = nn.Linear(192, num_classes)
    m.channels_last = False
    m.forward_cls = types.MethodType(DPT.forward_cls, m)

    def forward_mirror(x1, x2):
        a = m.forward_cls(x1)
        b = m.forward_cls(x2)
        return [a[0], b[0], a[1], b[1], None, None], [a[2], b[2]]
    m.forward_mirror = forward_mirror
    return m


def np_(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def grads_pick(model, names):
    out = {}
    for n, prm in model.named_parameters():
        if n in names:
            out["grad:" + n] = np_(prm.grad if prm.grad is not None else torch.zeros_like(prm))
    return out


def run_train_case(tag, model, fm, size, batch, num_classes, alpha, seed, grad_names, sub=None, opt_param=None):
    img, label = make_inputs(batch, size, num_classes, seed)
    model.train()
    model.zero_grad()
    loss, out = train_step(model, img, label, alpha, fm)
    params = [p for p in model.parameters()]
    opt = PolySGD(params, lr=0.05, weight_decay=5e-4, max_step=100)
    opt.zero_grad()
    loss.backward()
    fx = {k: np_(v) for k, v in out.items()}
    if sub is not None:                                    # big maps: keep a strided subsample + row 0
        for k in ("attn1", "attn2"):
            a = fx.pop(k)
            fx[k + "_sub"] = a[:, :, ::sub[0], ::sub[1]].copy()
            fx[k + "_row0"] = a[:, :, 0, :].copy()
            fx[k + "_sum"] = a.astype(np.float64).sum(axis=(2, 3))
        fx["sub"] = np.array(sub)
    fx.update(grads_pick(model, grad_names))
    if opt_param is not None:
        prm = dict(model.named_parameters())[opt_param]
        opt.step()
        fx["after_step:" + opt_param] = np_(prm)
        fx["lr_after_step"] = np.array(opt.param_groups[0]["lr"])
    fx["img"] = np_(img) if img.numel() < 200000 else np.zeros(0, np.float32)
    fx["label"] = np_(label)
    fx["meta"] = np.array([size, batch, num_classes, alpha, seed])
    fx["weights_checksum"] = np.array(weights_checksum(model.state_dict()))
    path = os.path.join(HERE, tag + ".npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, "loss=%.6f cls_align=%.6e aff_align=%.6e" % (
        float(fx["loss"]), float(fx["cls_align"]), float(fx["aff_align"])), flush=True)


def run_infer_case(tag, model, size, WH, seed, num_classes=20):
    img, _ = make_inputs(1, size, num_classes, seed)
    label = torch.zeros(1, num_classes)
    label[0, 3] = 1.0
    label[0, 11] = 1.0
    model.eval()
    fx = {"img": np_(img), "label": np_(label), "meta": np.array([size, WH[0], WH[1], seed])}
    for func in ("grad", "cam_grad", "grad_s", "cam_grad_s"):
        for start_layer in (0, 10):
            for aff in (True, False):
                if func != "grad" and not (start_layer == 10 and aff):
                    continue                               # full sweep only for the shipped func
                cam_dict, patch_dict, rows = infer_one_image(model, img, label, WH, start_layer, func, aff)
                key = "%s_s%d_a%d" % (func, start_layer, int(aff))
                for c, v in cam_dict.items():
                    fx["cam:%s:%d" % (key, c)] = v.astype(np.float32)
                fx["getam_rows:" + key] = rows
                for t in (0.2, 0.4):
                    fx["seed:%s:%.1f" % (key, t)] = seeds(cam_dict, WH[0], WH[1], t)
                if func == "grad" and start_layer == 10 and aff:
                    for c, v in patch_dict.items():
                        fx["patch_cam:%d" % c] = v.astype(np.float32)
    fx["weights_checksum"] = np.array(weights_checksum(model.state_dict()))
    path = os.path.join(HERE, tag + ".npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, flush=True)


def run_infer_case_big(tag, model, size, WH, seed, scales, combos, classes, num_classes=20):
    """Real-geometry CAM generation (infer_cam.py:141-215): full-resolution normalised CAMs (needed to prove that a
    seed pixel that differs is an fp tie), getam rows and argmax seeds at t in {0.2, 0.4}.  ``combos`` = list of
    (func, start_layer, aff); ``scales`` = the scale list BASELINE configs[3] names when longer than (1,)."""
    img, _ = make_inputs(1, size, num_classes, seed)
    label = torch.zeros(1, num_classes)
    for c in classes:
        label[0, c] = 1.0
    model.eval()
    fx = {"label": np_(label), "meta": np.array([size, WH[0], WH[1], seed]), "scales": np.array(scales, np.float64)}
    for func, start_layer, aff in combos:
        cam_dict, patch_dict, rows = infer_one_image(model, img, label, WH, start_layer, func, aff, scales=scales,
                                                     num_classes=num_classes)
        key = "%s_s%d_a%d" % (func, start_layer, int(aff))
        for c, v in cam_dict.items():
            fx["cam:%s:%d" % (key, c)] = v.astype(np.float32)
        for i, r in enumerate(rows):                          # ragged over scales: one entry per (scale, flip, class)
            fx["getam_row:%s:%d" % (key, i)] = r.astype(np.float32)
        for t in (0.2, 0.4):
            fx["seed:%s:%.1f" % (key, t)] = seeds(cam_dict, WH[0], WH[1], t, num_cls=num_classes + 1)
        if (func, start_layer, aff) == combos[0]:
            for c, v in patch_dict.items():
                fx["patch_cam:%d" % c] = v.astype(np.float32)
        print("  ", tag, key, "done", flush=True)
    fx["weights_checksum"] = np.array(weights_checksum(model.state_dict()))
    path = os.path.join(HERE, tag + ".npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, flush=True)


def run_getam_case(tag, model, size, seed, classes, num_classes=20):
    """The part of infer_cam.py:155-180 that works for EVERY backbone of DPT/ACR.py:155-160: forward_cam, one backward per
    positive class, ACR.getam for all four funcs.  (For `deit_distilled` the rest of the reference loop cannot run -- the
    patch-CAM reshape :156 and the affinity product :183-184 assume T = N + 1 -- but getam has a dedicated branch for it,
    `cams[:, 0, 2:]`, DPT/ACR.py:210-211, which is what this fixture pins.)"""
    img, _ = make_inputs(1, size, num_classes, seed)
    model.eval()
    model.zero_grad()
    cls_pred, x_patch_cls, attn, patch_cam = model.forward_cam(img)
    fx = {"img": np_(img), "meta": np.array([size, seed]), "classes": np.array(classes), "cls_pred": np_(cls_pred),
          "x_patch_cls": np_(x_patch_cls), "attn": np_(attn), "patch_cam": np_(patch_cam)}
    for c in classes:
        model.zero_grad()
        cls_pred[0, c].backward(retain_graph=True)
        for func in ("grad", "cam_grad", "grad_s", "cam_grad_s"):
            for start_layer in (0, 10):
                cam, _, _ = model.getam(0, start_layer=start_layer, func=func)
                fx["getam:%s_s%d:%d" % (func, start_layer, c)] = np_(cam)
    fx["weights_checksum"] = np.array(weights_checksum(model.state_dict()))
    path = os.path.join(HERE, tag + ".npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, {k: v.shape for k, v in fx.items() if hasattr(v, "shape") and k.startswith(("attn", "getam:grad_s10"))}, flush=True)


def main():
    _install_timm_stub()
    torch.set_num_threads(8)
    torch.manual_seed(0)
    from DPT.ACR import ACR

    which = set(sys.argv[1:]) or {"layout", "hyb64", "hyb96", "tiny224", "infer64", "infer96", "hyb448", "infer384", "ms96", "ms384",
                                  "coco512", "distil96", "layouts"}

    model = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False)
    if "layout" in which:
        layout = {k: list(v.shape) for k, v in model.state_dict().items()}
        with open(os.path.join(HERE, "state_dict_layout.json"), "w") as f:
            json.dump(layout, f, indent=0)
        print("layout: %d tensors, %.2f M params" % (len(layout), sum(v.numel() for v in model.state_dict().values()) / 1e6))
    fill_state_dict(model, seed=0)

    gnames = {"cls_head.weight", "cls_head.bias",
              "pretrained.model.blocks.11.attn.qkv.bias", "pretrained.model.blocks.0.attn.qkv.bias",
              "pretrained.model.blocks.5.attn.proj.bias", "pretrained.model.blocks.0.norm1.weight",
              "pretrained.model.cls_token", "pretrained.model.patch_embed.backbone.stem.norm.bias",
              "pretrained.model.patch_embed.proj.bias"}
    if "hyb64" in which:
        run_train_case("train_hybrid_64_b2", model, model.forward_mirror, 64, 2, 20, 125, 1, gnames,
                       opt_param="cls_head.bias")
        fill_state_dict(model, seed=0)
    if "hyb96" in which:
        run_train_case("train_hybrid_96_b1", model, model.forward_mirror, 96, 1, 20, 125, 2, gnames)
    if "infer64" in which:
        run_infer_case("infer_hybrid_64", model, 64, (50, 70), 3)
    if "infer96" in which:
        run_infer_case("infer_hybrid_96", model, 96, (75, 61), 4)
    if "hyb448" in which:
        run_train_case("train_hybrid_448_b1", model, model.forward_mirror, 448, 1, 20, 125, 5, gnames, sub=(97, 89))

    if "infer384" in which:          # real inference geometry of train_acr.sh:26-37 (crop 384 -> T = 577), VOC-sized output
        run_infer_case_big("infer_hybrid_384", model, 384, (375, 500), 7, (1,), [("grad", 10, True)], [14])
    if "ms96" in which:              # BASELINE configs[3] scale set at a small base: T in {10, 37, 82, 145}
        run_infer_case_big("infer_ms_hybrid_96", model, 96, (75, 61), 8, (0.5, 1.0, 1.5, 2.0),
                           [("grad", 10, True), ("cam_grad", 10, True), ("grad_s", 10, True), ("cam_grad_s", 10, True),
                            ("grad", 0, False)], [3, 11])
    if "ms384" in which:             # BASELINE configs[3] at the real base size: T in {145, 577, 1297, 2305}
        run_infer_case_big("infer_ms_hybrid_384", model, 384, (188, 250), 9, (0.5, 1.0, 1.5, 2.0), [("grad", 10, True)],
                           [6, 14])
    if "coco512" in which:           # BASELINE configs[4]: train_acr_coco.py:91 ACR(num_classes=80), 512^2 (T = 1025)
        coco = ACR(num_classes=80, backbone_name="vitb_hybrid", use_pretrain=False)
        fill_state_dict(coco, seed=0)
        run_train_case("train_coco_512_b1", coco, coco.forward_mirror, 512, 1, 80, 125, 10, gnames, sub=(113, 101))
        del coco

    if "distil96" in which:          # DPT/ACR.py:155-160 'deit_distilled' (cls + dist token: T = N + 2), getam's [:, 2:] branch
        dist_model = ACR(num_classes=20, backbone_name="deit_distilled", use_pretrain=False)
        layout = {k: list(v.shape) for k, v in dist_model.state_dict().items()}
        with open(os.path.join(HERE, "state_dict_layout_distil.json"), "w") as f:
            json.dump(layout, f, indent=0)
        fill_state_dict(dist_model, seed=0)
        run_getam_case("getam_distil_96", dist_model, 96, 13, [3, 11])
        del dist_model

    if "layouts" in which:           # state-dict layouts (key -> shape) of the other backbones of DPT/ACR.py:155-160: data only
        for name in ("vitb", "deit", "vitl"):
            mdl = ACR(num_classes=20, backbone_name=name, use_pretrain=False)
            layout = {k: list(v.shape) for k, v in mdl.state_dict().items()}
            with open(os.path.join(HERE, "state_dict_layout_%s.json" % name), "w") as f:
                json.dump(layout, f, indent=0)
            print("layout %s: %d tensors, %.1f M params" % (name, len(layout), sum(v.numel() for v in mdl.state_dict().values()) / 1e6))
            del mdl

    if "tiny224" in which:
        tiny = build_tiny(20)
        layout = {k: list(v.shape) for k, v in tiny.state_dict().items()}
        with open(os.path.join(HERE, "state_dict_layout_tiny.json"), "w") as f:
            json.dump(layout, f, indent=0)
        fill_state_dict(tiny, seed=0)
        tnames = {"cls_head.weight", "cls_head.bias", "pretrained.model.blocks.11.attn.qkv.bias",
                  "pretrained.model.blocks.0.attn.qkv.bias", "pretrained.model.cls_token",
                  "pretrained.model.patch_embed.proj.bias"}
        run_train_case("train_tiny_224_b2", tiny, tiny.forward_mirror, 224, 2, 20, 125, 6, tnames, sub=(13, 11))


if __name__ == "__main__":
    main()
