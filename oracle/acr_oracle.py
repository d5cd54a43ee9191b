"""CPU ORACLE for the ACR hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (fp32, CPU) functional restatement of the reference algorithm for the path named by
BASELINE.json's north_star.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; the product (``acr_wsss_amd``) never does and fails
loudly when its HIP library is missing.

Parity status: PINNED.  Every function below is checked (tests/test_oracle_golden.py) against golden
vectors produced by importing and running the reference itself on CPU in the build container
(tests/golden/make_golden.py, committed next to the fixtures).

Everything is a pure function of a reference-layout ``state_dict`` (315 tensors for hybrid-base; see
tests/golden/state_dict_layout.json) -- there are no nn.Modules here, which keeps the oracle
independent of the product's module tree.  Citations are ``file:line`` under the reference root.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# configs
# --------------------------------------------------------------------------------------------
HYBRID_BASE = dict(embed_dim=768, depth=12, heads=12, hybrid=True, patch=16, start_index=1,
                   stage_depths=(3, 4, 9), stage_chs=(256, 512, 1024), prefix="pretrained.model.")
# DPT/ACR.py:155-160 'deit_distilled' -> DPT/vit.py deitb16_distil_384: plain 16x16 patch embedding, cls + distillation
# token (start_index 2, models/vision_transformer.py:466-472)
DEIT_DISTILLED = dict(embed_dim=768, depth=12, heads=12, hybrid=False, patch=16, start_index=2, distilled=True,
                      prefix="pretrained.model.")
# the plain-ViT backbones of DPT/ACR.py:155-160: 'vitb' / 'deit' (vitb16_384, deitb16_384: identical encoders) and 'vitl'
VIT_BASE = dict(embed_dim=768, depth=12, heads=12, hybrid=False, patch=16, start_index=1, prefix="pretrained.model.")
VIT_LARGE = dict(embed_dim=1024, depth=24, heads=16, hybrid=False, patch=16, start_index=1, prefix="pretrained.model.")
VIT_TINY = dict(embed_dim=192, depth=12, heads=3, hybrid=False, patch=16, start_index=1,
                prefix="pretrained.model.")


# --------------------------------------------------------------------------------------------
# optional activation rounding: the same algorithm with every activation tensor (and its gradient) stored in bf16 -- the error
# budget a bf16 training mode has BY CONSTRUCTION, independent of any kernel (tests/test_model_gpu.py compares the HIP bf16
# mode's deviation from the fp32 oracle with this emulation's).  Off (None) everywhere else: _r is then the identity.
# --------------------------------------------------------------------------------------------
_ROUND = None


class _RoundBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class bf16_activations:
    """with bf16_activations(): ...  -- every tensor the functions below hand from one layer to the next is rounded to bf16
    (forward value and gradient); products, norm statistics, softmax and the loss stay in fp32."""

    def __enter__(self):
        global _ROUND
        _ROUND = _RoundBf16.apply

    def __exit__(self, *a):
        global _ROUND
        _ROUND = None


def _r(x):
    return x if _ROUND is None else _ROUND(x)


# --------------------------------------------------------------------------------------------
# ResNetV2 stem (hybrid patch embedding)
# --------------------------------------------------------------------------------------------
def _same_pad_amount(n, k, s):
    """models/layers/padding.py:18-19 (dilation 1)."""
    return max((math.ceil(n / s) - 1) * s + (k - 1) + 1 - n, 0)


def pad_same(x, k, s, value=0.0):
    """TF 'SAME': the odd pixel goes to the right/bottom.  models/layers/padding.py:28-33."""
    ph = _same_pad_amount(x.shape[-2], k, s)
    pw = _same_pad_amount(x.shape[-1], k, s)
    if ph or pw:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], value=value)
    return x


def std_conv_same(x, w, stride):
    """Weight-standardised conv with dynamic SAME padding.  models/layers/std_conv.py:40-65:
    w_hat = (w - mean) / (std + 1e-5) per output channel (biased std), recomputed every forward."""
    std, mean = torch.std_mean(w, dim=[1, 2, 3], keepdim=True, unbiased=False)
    w_hat = (w - mean) / (std + 1e-5)
    k = w.shape[-1]
    if stride == 1 and (k - 1) % 2 == 0:          # static case, padding.py:22-24,44-46
        return _r(F.conv2d(x, _r(w_hat), None, 1, (k - 1) // 2))
    return _r(F.conv2d(pad_same(x, k, stride), _r(w_hat), None, stride, 0))


def gn(x, sd, name, relu):
    """GroupNorm(32, eps 1e-5) [+ ReLU].  models/layers/norm_act.py:69-85."""
    y = F.group_norm(x, 32, sd[name + ".weight"], sd[name + ".bias"], 1e-5)
    return _r(F.relu(y) if relu else y)


def bottleneck(x, sd, pre, stride, has_down):
    """Non-preact bottleneck.  models/resnetv2.py:171-216, DownsampleConv :219-228."""
    shortcut = x
    if has_down:
        shortcut = gn(std_conv_same(x, sd[pre + "downsample.conv.weight"], stride), sd, pre + "downsample.norm", False)
    y = gn(std_conv_same(x, sd[pre + "conv1.weight"], 1), sd, pre + "norm1", True)
    y = gn(std_conv_same(y, sd[pre + "conv2.weight"], stride), sd, pre + "norm2", True)
    y = gn(std_conv_same(y, sd[pre + "conv3.weight"], 1), sd, pre + "norm3", False)
    return _r(F.relu(y + shortcut))


def resnetv2_features(x, sd, pre, cfg):
    """models/resnetv2.py:277-308 ('same' stem), :311-383 with layers=(3,4,9), preact=False."""
    x = gn(std_conv_same(x, sd[pre + "stem.conv.weight"], 2), sd, pre + "stem.norm", True)
    x = F.max_pool2d(pad_same(x, 3, 2, value=-float("inf")), 3, 2)      # pool2d_same.py:34-38
    for si, depth in enumerate(cfg["stage_depths"]):
        for bi in range(depth):
            stride = 2 if (si > 0 and bi == 0) else 1
            x = bottleneck(x, sd, "%sstages.%d.blocks.%d." % (pre, si, bi), stride, bi == 0)
    return x                                                              # norm = Identity, head = identity


# --------------------------------------------------------------------------------------------
# ViT
# --------------------------------------------------------------------------------------------
def resize_pos_embed(pos, gh, gw, start_index):
    """models/vision_transformer.py:490-504 (bilinear, align_corners=False)."""
    tok, grid = pos[:, :start_index], pos[0, start_index:]
    g0 = int(math.sqrt(grid.shape[0]))
    grid = grid.reshape(1, g0, g0, -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(gh, gw), mode="bilinear")
    grid = grid.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)
    return torch.cat([tok, grid], dim=1)


def attention(x, sd, pre, heads):
    """models/vision_transformer.py:198-214.  Returns (out, P) with P = softmax(q k^T d^-0.5) of shape
    (B, H, T, T) -- the tensor the reference stores on the module and hooks for its gradient."""
    B, T, C = x.shape
    d = C // heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(B, T, 3, heads, d).permute(2, 0, 3, 1, 4)
    qkv = _r(qkv)
    q, k, v = qkv[0], qkv[1], qkv[2]
    P = ((q @ k.transpose(-2, -1)) * (d ** -0.5)).softmax(dim=-1)
    if P.requires_grad:
        P.retain_grad()
    out = _r((_r(P) @ v).transpose(1, 2).reshape(B, T, C))        # (bf16 mode: P enters the second product rounded)
    return _r(F.linear(out, sd[pre + "proj.weight"], sd[pre + "proj.bias"])), P


def block(x, sd, pre, heads):
    """models/vision_transformer.py:230-233; LayerNorm eps 1e-6 (:299); Mlp :158-164 (exact GELU)."""
    C = x.shape[-1]
    a, P = attention(_r(F.layer_norm(x, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-6)), sd, pre + "attn.", heads)
    x = _r(x + a)
    h = _r(F.layer_norm(x, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-6))
    h = _r(F.linear(_r(F.gelu(_r(F.linear(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])))),
                    sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"]))
    return _r(x + h), P


def forward_flex(x, sd, cfg):
    """models/vision_transformer.py:449-486.  Returns (layer_4, [P_0..P_{L-1}]) where layer_4 is the
    output of the last block *before* the final norm (DPT/vit.py:431 hook "4", hooks[3] = 11)."""
    pre = cfg["prefix"]
    b, c, h, w = x.shape
    pos = resize_pos_embed(sd[pre + "pos_embed"], h // cfg["patch"], w // cfg["patch"], cfg["start_index"])
    if cfg["hybrid"]:
        x = resnetv2_features(x, sd, pre + "patch_embed.backbone.", cfg)
        x = _r(F.conv2d(x, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"]))
    else:
        x = F.conv2d(x, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"], stride=cfg["patch"])
    x = x.flatten(2).transpose(1, 2)
    toks = [sd[pre + "cls_token"].expand(b, -1, -1)]
    if cfg.get("distilled"):                                 # models/vision_transformer.py:466-472
        toks.append(sd[pre + "dist_token"].expand(b, -1, -1))
    x = _r(torch.cat(toks + [x], dim=1) + pos)
    maps = []
    for i in range(cfg["depth"]):
        x, P = block(x, sd, "%sblocks.%d." % (pre, i), cfg["heads"])
        maps.append(P)
    return x, maps


def forward_cls(x, sd, cfg):
    """DPT/ACR.py:92-116 -> (x_cls, x_patch_cls, attn (B,L,T,T) head-mean stack, per-layer P list)."""
    layer_4, maps = forward_flex(x, sd, cfg)
    W, bcls = sd["cls_head.weight"], sd["cls_head.bias"]
    x_cls = F.linear(layer_4[:, 0, :], W, bcls)
    x_patch_cls = F.linear(layer_4[:, 1:, :].mean(dim=1), W, bcls)
    attn = torch.stack([P.mean(dim=1) for P in maps], dim=1)
    return x_cls, x_patch_cls, attn, maps, layer_4


def forward_cam(x, sd, cfg):
    """DPT/ACR.py:118-143: forward_cls + x_patch_cam = relu(cls_head(patch tokens)) (B,N,C)."""
    x_cls, x_patch_cls, attn, maps, layer_4 = forward_cls(x, sd, cfg)
    patch_cam = F.relu(F.linear(layer_4[:, 1:, :], sd["cls_head.weight"], sd["cls_head.bias"]))
    return x_cls, x_patch_cls, attn, patch_cam, maps


# --------------------------------------------------------------------------------------------
# ACR loss (train_acr.py:140-168) -- literal restatement with the in-place block flips
# --------------------------------------------------------------------------------------------
def acr_loss_inline(attn1, attn2, x1, x2, label, p, alpha):
    attn2 = attn2.clone()                      # the reference mutates the stack output in place
    a1_cls = attn1[:, :, 0, 1:].unsqueeze(2)
    a2_cls = attn2[:, :, 0, 1:].unsqueeze(2)
    a1_aff = attn1[:, :, 1:, 1:]
    a2_aff = attn2[:, :, 1:, 1:]
    for i in range(p):
        a2_cls[:, :, :, i * p:i * p + p] = a2_cls[:, :, :, i * p:i * p + p].flip(3)
    for i in range(p):
        a2_aff[:, :, i * p:i * p + p, :] = a2_aff[:, :, i * p:i * p + p, :].flip(2)
    for i in range(p):
        a2_aff[:, :, :, i * p:i * p + p] = a2_aff[:, :, :, i * p:i * p + p].flip(3)
    cls_align = F.l1_loss(a1_cls, a2_cls, reduction="mean")
    aff_align = F.l1_loss(a1_aff, a2_aff, reduction="mean")
    cls1 = F.multilabel_soft_margin_loss(x1, label)
    cls2 = F.multilabel_soft_margin_loss(x2, label)
    loss = cls1 + cls2 + cls_align * alpha + aff_align * alpha
    return loss, dict(cls_align=cls_align, aff_align=aff_align, cls_loss_1=cls1, cls_loss_2=cls2, loss=loss)


def flip_perm(p, device=None):
    """pi(i*p + j) = i*p + (p-1-j): the token permutation induced by a horizontal image flip."""
    idx = torch.arange(p * p, device=device).reshape(p, p).flip(1).reshape(-1)
    return idx


def acr_align_perm(attn1, attn2, p):
    """Permutation form of the two alignment terms (bit-exact with the in-place flips)."""
    pi = flip_perm(p, attn1.device)
    a2_cls = attn2[:, :, 0, 1:][:, :, pi]
    a2_aff = attn2[:, :, 1:, 1:][:, :, pi][:, :, :, pi]
    return (attn1[:, :, 0, 1:] - a2_cls).abs().mean(), (attn1[:, :, 1:, 1:] - a2_aff).abs().mean()


def train_step(sd, cfg, img, label, alpha):
    """train_acr.py:135-168 on the functional model (view 2 = horizontal flip)."""
    a = forward_cls(img, sd, cfg)
    b = forward_cls(img.flip(-1), sd, cfg)
    p = img.shape[2] // 16
    loss, terms = acr_loss_inline(a[2], b[2], a[0], b[0], label, p, alpha)
    terms.update(x_cls_1=a[0], x_cls_2=b[0], x_p_cls_1=a[1], x_p_cls_2=b[1], attn1=a[2], attn2=b[2])
    return loss, terms


def poly_sgd_step(params, grads, bufs, step, max_step, lr0, wt_dec):
    """tool/torchutils.py:10-31.  SGD(params, lr, weight_decay) binds weight_decay to SGD's *momentum*
    slot: effective momentum = wt_dec, weight decay = 0, dampening 0; lr = lr0 (1-step/max)^0.9."""
    lr = lr0 * (1 - step / max_step) ** 0.9 if step < max_step else lr0
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        if bufs[i] is None:
            bufs[i] = g.clone()
        else:
            bufs[i].mul_(wt_dec).add_(g)
        p.data.add_(bufs[i], alpha=-lr)
    return lr


# --------------------------------------------------------------------------------------------
# GETAM (DPT/ACR.py:177-215) and the infer_cam.py per-image loop
# --------------------------------------------------------------------------------------------
def getam(maps, grads, batch, start_layer=0, func="grad", distilled=False):
    cams = []
    for P, G in zip(maps, grads):
        cam, grad = P[batch], G[batch]                       # (H,T,T)
        if func == "cam_grad_s":
            cam = (grad * cam).clamp(min=0).mean(dim=0) * grad.clamp(min=0).mean(dim=0)
        elif func == "cam_grad":
            cam = (grad * cam).clamp(min=0).mean(dim=0)
        elif func == "grad":
            cam = grad.clamp(min=0).mean(dim=0)
        elif func == "grad_s":
            cam = grad.clamp(min=0).mean(dim=0)
            cam = cam * cam
        else:
            raise ValueError(func)
        cams.append(cam.unsqueeze(0))
    tot = torch.stack(cams[start_layer:]).sum(dim=0)
    return torch.relu(tot[:, 0, 2:] if distilled else tot[:, 0, 1:])


def infer_image(sd, cfg, img, label, out_hw, start_layer=10, func="grad", aff=True, scales=(1,)):
    """infer_cam.py:141-215 for one image.  ``out_hw`` = (W, H) of the reference = (image height, width).
    Returns (cam_dict, patch_cam_dict, getam_rows)."""
    W, H = out_hw
    C = label.shape[1]
    b, c, h, w = img.shape
    cam_list, patch_list, rows = [], [], []
    sdg = {k: v.detach().requires_grad_(True) if v.is_floating_point() else v for k, v in sd.items()}
    for scale in scales:
        for hflip in (1, 2):
            cam_matrix = torch.zeros((C, W, H))
            inp = F.interpolate(img, size=(int(h * scale), int(w * scale)), mode="bilinear", align_corners=False)
            if hflip % 2 == 1:
                inp = inp.flip(-1)
            cls_pred, _, attn, patch_cam, maps = forward_cam(inp, sdg, cfg)
            ph, pw = int((h * scale) // 16), int((w * scale) // 16)
            pc = patch_cam.permute(0, 2, 1).reshape(1, C, ph, pw)
            pc = F.interpolate(pc, [W, H], mode="bilinear", align_corners=False)[0]
            pc = pc.detach().numpy() * label[0].view(C, 1, 1).numpy()
            if hflip % 2 == 1:
                pc = np.flip(pc, axis=-1)
            patch_list.append(pc)
            patch_aff = attn[:, :, 1:, 1:].sum(dim=1).detach()
            for ci in range(C):
                if label[0, ci] > 1e-5:
                    for P in maps:
                        P.grad = None
                    cls_pred[0, ci].backward(retain_graph=True)
                    cam = getam([P.detach() for P in maps], [P.grad for P in maps], 0, start_layer, func)
                    rows.append(cam.numpy().copy())
                    if aff:
                        cam = torch.matmul(patch_aff, cam.unsqueeze(2))
                    cam = cam.reshape(ph, pw)
                    cam = F.interpolate(cam[None, None], (W, H), mode="bilinear", align_corners=True)
                    cam_matrix[ci] = cam[0, 0]
            cm = cam_matrix.numpy()
            if hflip % 2 == 1:
                cm = np.flip(cm, axis=2)
            cam_list.append(cm)
    psum = np.sum(patch_list, axis=0)
    pmin, pmax = psum.min((1, 2), keepdims=True), psum.max((1, 2), keepdims=True)
    pnorm = (psum - pmin) / (pmax - pmin + 1e-5)
    csum = np.sum(cam_list, axis=0)
    cmin, cmax = csum.min((1, 2), keepdims=True), csum.max((1, 2), keepdims=True)
    cnorm = (csum - cmin) / (cmax - cmin + 1e-6)
    keep = [ci for ci in range(C) if label[0, ci] > 1e-5]
    rows_out = np.stack(rows) if len({r.shape for r in rows}) == 1 else rows      # ragged over scales
    return {ci: cnorm[ci] for ci in keep}, {ci: pnorm[ci] for ci in keep}, rows_out


# --------------------------------------------------------------------------------------------
# evaluation.py:13-85 (seed argmax + IoU counters)
# --------------------------------------------------------------------------------------------
def seeds_from_cam_dict(cam_dict, t, num_cls=21):
    h, w = next(iter(cam_dict.values())).shape
    tensor = np.zeros((num_cls, h, w), np.float32)
    for k, v in cam_dict.items():
        tensor[k + 1] = v
    tensor[0] = t
    return np.argmax(tensor, axis=0).astype(np.uint8)


def iou_counts(pred, gt, num_cls=21):
    cal = gt < 255
    mask = (pred == gt) * cal
    P = np.array([np.sum((pred == i) * cal) for i in range(num_cls)])
    T = np.array([np.sum((gt == i) * cal) for i in range(num_cls)])
    TP = np.array([np.sum((gt == i) * mask) for i in range(num_cls)])
    return TP, P, T


def miou(TP, P, T):
    iou = TP / (T + P - TP + 1e-10)
    return float(np.mean(iou) * 100.0)
