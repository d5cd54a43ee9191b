"""Encoder of the ACR model: ResNetV2 stem (hybrid patch embedding) + ViT whose attention runs in HIP.

Module / parameter names reproduce the reference's state-dict layout exactly (315 tensors for hybrid-base,
tests/golden/state_dict_layout.json) so reference checkpoints load with ``strict=True``:
  models/resnetv2.py:171-216,250-383   Bottleneck / ResNetStage / ResNetV2 (layers=(3,4,9), preact=False, 'same' stem)
  models/layers/std_conv.py:40-65      StdConv2dSame (weight standardisation + TF SAME padding)
  models/layers/norm_act.py:69-85      GroupNormAct
  models/vision_transformer_hybrid.py:67-106   HybridEmbed
  models/vision_transformer.py:148-233,262-504 Mlp / Attention / Block / VisionTransformer.forward_flex

Convolutions, GroupNorm/LayerNorm and the MLP GEMMs stay on stock PyTorch-ROCm ops (MIOpen / hipBLASLt) as
SURVEY 7 scopes them; the attention core -- softmax(q k^T) v with the head-mean side output the ACR loss
consumes, and its backward -- is ``ops.attention_core`` (hand-written gfx950 kernels, include/acr_hip.h).
"""
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def set_math(module, math):
    """Select how the fp32 matrix products of ``module`` (a whole model or any part of it) are evaluated -- a property of the
    MODEL (its modules carry it into every call of the C ABI), not of the process:
      "f32"        exact-fp32 MFMA, the reference's arithmetic (default; what CAM inference and the parity fixtures run);
      "f32_split"  the same fp32 tensors with every product of the block Linears, the stem's 1x1 convolutions and the
                   attention as six bf16-MFMA terms of a three-way operand split (include/acr_hip.h: acr_math) --
                   fp32-accurate, 24 mantissa bits per operand, fp32 accumulate.
    Has no effect on a bf16 model.  Returns the module."""
    from . import _lib
    code = _lib.MATH[math] if isinstance(math, str) else int(math)
    for m in module.modules():
        if hasattr(type(m), "acr_math"):
            m.acr_math = code
    return module


# ------------------------------------------------------------------------------------------------
# ResNetV2 pieces
# ------------------------------------------------------------------------------------------------
def _same_pad(n, k, s):
    return max((math.ceil(n / s) - 1) * s + (k - 1) + 1 - n, 0)


def pad_same(x, k, s, value=0.0):
    ph, pw = _same_pad(x.shape[-2], k, s), _same_pad(x.shape[-1], k, s)
    if ph > 0 or pw > 0:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], value=value)
    return x


class StdConv2dSame(nn.Conv2d):
    """Weight-standardised conv, TF 'SAME' padding (asymmetric: the odd pixel goes right/bottom)."""

    def __init__(self, cin, cout, kernel_size, stride=1, eps=1e-5):
        static = stride == 1 and (kernel_size - 1) % 2 == 0
        super().__init__(cin, cout, kernel_size, stride=stride, padding=(kernel_size - 1) // 2 if static else 0, bias=False)
        self.dynamic_pad = not static
        self.eps = eps

    def standardized_weight(self):
        std, mean = torch.std_mean(self.weight, dim=[1, 2, 3], keepdim=True, unbiased=False)
        return (self.weight - mean) / (std + self.eps)

    def forward(self, x):
        w_hat = self._w_hat if self._w_hat is not None else self.standardized_weight()
        if self.hip_3x3 and self.stride[0] == 2 and self.kernel_size[0] in (3, 7) and ops.conv_s2_fusable(x, w_hat, 2, self.acr_math):
            # the stem's 7x7 and the two stride-2 3x3 convolutions under split products: space-to-depth + tap-table implicit GEMM
            # (SAME padding = the taps' validity masks; no padded copy, no library kernel, no layout transposes)
            return ops.conv_s2(x, w_hat, self._w_imgs)
        if self.dynamic_pad:
            x = pad_same(x, self.kernel_size[0], self.stride[0])
        if self.hip_1x1 and self.hip_1x1_strided and self.kernel_size == (1, 1) and self.stride[0] == 2 and x.is_cuda:
            # the two stride-2 1x1 convolutions (the shortcuts of stages 1 and 2; SAME padding is empty for a 1x1 kernel:
            # y[i][j] = W x[2i][2j]): subsample first, then the same NCHW GEMM kernels as every other 1x1 -- the library ran
            # them as layout transposes + a Tensile GEMM; autograd's slice backward scatters the input gradient back
            xs = ops.subsample2(x)
            if ops.conv1x1_fusable(xs, w_hat, 1):
                return ops.conv1x1(xs, w_hat, self._w_hat_t, self.acr_math, self._w_imgs)
        if self.hip_1x1 and ops.conv1x1_fusable(x, w_hat, self.stride[0]):
            return ops.conv1x1(x, w_hat, self._w_hat_t, self.acr_math, self._w_imgs)   # NCHW 1x1 conv = per-sample MFMA GEMM, no layout transposes
        if self.hip_3x3 and not self.dynamic_pad and ops.conv3x3_fusable(x, w_hat, self.stride[0], self.acr_math):
            return ops.conv3x3(x, w_hat, self._w_imgs)                     # split-product implicit GEMM, no layout transposes
        if (self.hip_3x3 and self.pad_narrow and not self.dynamic_pad and x.dim() == 4 and 4 <= x.shape[3] < 16 and x.shape[3] % 4 == 0
                and self.kernel_size == (3, 3)):
            # maps narrower than the kernels' 16-pixel rows (CAM generation at scale 0.5: 12 x 12 in the last stage): zero columns on
            # the right ARE the SAME padding of the last real column, so the convolution of the widened map, cut back, is the result
            xp = F.pad(x, (0, 16 - x.shape[3]))
            if ops.conv3x3_fusable(xp, w_hat, self.stride[0], self.acr_math):
                return ops.conv3x3(xp, w_hat, self._w_imgs)[..., :x.shape[3]].contiguous()
        return F.conv2d(x, w_hat, None, self.stride, self.padding)

    def forward_skip(self, x):
        """(conv(x), x_skip): x_skip is what a parallel branch (the shortcut) should read -- on the HIP 1x1 path its
        gradient is then added inside the input-gradient GEMM."""
        w_hat = self._w_hat if self._w_hat is not None else self.standardized_weight()
        if self.hip_1x1 and x.requires_grad and ops.conv1x1_fusable(x, w_hat, self.stride[0]):
            return ops.conv1x1_skip(x, w_hat, self._w_hat_t, self.acr_math, self._w_imgs)
        return self.forward(x), x

    hip_1x1 = True
    hip_1x1_strided = True      # A/B: stride-2 1x1 convolutions as subsample + HIP GEMM
    hip_3x3 = True      # A/B: the stem's 3x3 convolutions under f32_split on csrc/conv3x3.hip
    pad_narrow = True      # A/B: maps of 4 / 8 / 12 columns widened to 16 instead of the library
    acr_math = 0            # _lib.MATH code of the fp32 products (set_math)

    _w_hat = None           # set for one forward by ResNetV2 when all weights are standardised in one fused launch
    _w_hat_t = None         # bf16 1x1 convolutions: its (cin, cout) copy, written by the same launch
    _w_imgs = None          # split products: (image for the forward, image for the input gradient) of _w_hat, made by ResNetV2 for
                            # all convolutions of a group in ONE launch (ops.x3_image_many)

    def image_specs(self, w_hat):
        """The two strided views of the standardised weight whose split-product images this convolution's forward and input
        gradient multiply by (ops.x3_image_many) -- or None when it does not run on the image kernels."""
        if self.acr_math != 1 or w_hat.dtype != torch.float32 or not w_hat.is_cuda:
            return None
        co, ci, k, _ = w_hat.shape
        if k == 1 and self.hip_1x1 and ops.CONV1X1_WIMG and ops.F32_HIP_CONV1X1 and ci % 32 == 0 and co % 32 == 0 and (
                self.stride[0] == 1 or (self.stride[0] == 2 and self.hip_1x1_strided)):
            return ((w_hat, 0, co, ci, ci, ci, 0, 1),               # W (co x ci)
                    (w_hat, 0, ci, co, 1, co, 0, ci))               # W^T (ci x co): the input gradient's operand
        if k == 3 and self.stride[0] == 1 and not self.dynamic_pad and self.hip_3x3 and ops.CONV3X3_WIMG and ci % 16 == 0 and co % 16 == 0:
            return ((w_hat, 0, co, 9 * ci, 9 * ci, ci, 1, 9),        # packed w[co][t * ci + c]
                    (w_hat, 8, ci, 9 * co, 9, co, -1, 9 * ci))       # input-gradient pack w[o][c][8 - t'] as (ci x 9 co)
        if self.stride[0] == 2 and self.hip_3x3 and ops.CONV_S2_HIP and co % 16 == 0:
            # stride 2 (ops.ConvS2Fn), for inputs of even height and width: the forward pack, and for the 3x3s the four pixel-phase
            # packs of the input gradient (two small gather / permute copies of the weight first)
            if k == 3 and ci % 16 == 0:
                plan = ops.conv_s2_plan(3, ci, 32, 32, w_hat.device)
                return ((w_hat, 0, co, 9 * ci, 9 * ci, ci, 1, 9),) + plan.dgrad_specs(plan.pack_dgrad(w_hat))
            if k == 7 and 4 * ci <= 16:
                w16 = ops.conv_s2_plan(7, ci, 32, 32, w_hat.device).pack(w_hat)
                return ((w16, 0, co, w16.shape[1], w16.shape[1], w16.shape[1], 0, 1),)
        return None


class GroupNormAct(nn.GroupNorm):
    """GroupNorm(32) [+ ReLU]; ``forward(x, resid)`` additionally fuses ``relu(gn(x) + resid)`` (the tail of a
    bottleneck).  bf16 NCHW tensors on the GPU run the fused HIP kernel (one read + one write forward), everything
    else the stock torch ops with identical semantics."""
    fused = True

    def __init__(self, channels, apply_act=True):
        super().__init__(32, channels, eps=1e-5)
        self.apply_act = apply_act

    def forward(self, x, resid=None):
        if self.fused and ops.groupnorm_fusable(x, resid):
            act = "add_relu" if resid is not None else ("relu" if self.apply_act else "none")
            return ops.groupnorm_act(x, self.weight, self.bias, act, resid, self.eps)
        x = F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
        if resid is not None:
            return F.relu(x + resid)
        return F.relu(x) if self.apply_act else x


class MaxPool2dSame(nn.Module):
    hip_pool = True

    def forward(self, x):
        if self.hip_pool and x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.dim() == 4 and not torch.is_autocast_enabled():
            ph, pw = _same_pad(x.shape[-2], 3, 2), _same_pad(x.shape[-1], 3, 2)
            return ops.maxpool3x3s2_same(x, ph // 2, pw // 2, ph, pw)       # -inf SAME padding folded into the kernel
        return F.max_pool2d(pad_same(x, 3, 2, value=-float("inf")), 3, 2)


class DownsampleConv(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv = StdConv2dSame(cin, cout, 1, stride=stride)
        self.norm = GroupNormAct(cout, apply_act=False)

    def forward(self, x):
        return self.norm(self.conv(x))


class Bottleneck(nn.Module):
    def __init__(self, cin, cout, stride, downsample):
        super().__init__()
        mid = cout // 4
        self.downsample = DownsampleConv(cin, cout, stride) if downsample else None
        self.conv1 = StdConv2dSame(cin, mid, 1)
        self.norm1 = GroupNormAct(mid)
        self.conv2 = StdConv2dSame(mid, mid, 3, stride=stride)
        self.norm2 = GroupNormAct(mid)
        self.conv3 = StdConv2dSame(mid, cout, 1)
        self.norm3 = GroupNormAct(cout, apply_act=False)

    def forward(self, x):
        x, skip = self.conv1.forward_skip(x)
        shortcut = skip if self.downsample is None else self.downsample(skip)
        x = self.norm1(x)
        x = self.norm2(self.conv2(x))
        return self.norm3(self.conv3(x), shortcut)          # relu(gn(conv3) + shortcut), fused on the bf16 path


class ResNetStage(nn.Module):
    def __init__(self, cin, cout, stride, depth):
        super().__init__()
        self.blocks = nn.Sequential(*[Bottleneck(cin if i == 0 else cout, cout, stride if i == 0 else 1, i == 0)
                                      for i in range(depth)])

    def forward(self, x):
        return self.blocks(x)


class ResNetV2(nn.Module):
    """Non-preact ResNetV2 feature extractor, layers (3,4,9), channels (256,512,1024), output stride 16."""

    def __init__(self, layers=(3, 4, 9), channels=(256, 512, 1024), in_chans=3, stem_chs=64):
        super().__init__()
        self.stem = nn.Sequential(OrderedDict([
            ("conv", StdConv2dSame(in_chans, stem_chs, 7, stride=2)),
            ("norm", GroupNormAct(stem_chs)),
            ("pool", MaxPool2dSame())]))
        stages, prev = [], stem_chs
        for i, (d, c) in enumerate(zip(layers, channels)):
            stages.append(ResNetStage(prev, c, 1 if i == 0 else 2, d))
            prev = c
        self.stages = nn.Sequential(*stages)
        self.num_features = prev
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    fused_weight_std = True
    _frozen = None                # (key, w_hats, transposed) of the last forward with frozen weights
    frozen_generation = 0         # bumped whenever that cache is rebuilt (PrefixGraph holds pointers into it)

    def _standardised(self, convs, x):
        """The standardised weights of all convolutions: one launch per forward while training; with frozen weights
        (CAM generation: requires_grad off) they are kept until a weight changes (tensor version or storage)."""
        frozen = not any(c.weight.requires_grad for c in convs)
        if frozen:
            key = (x.dtype,) + tuple((c.weight.data_ptr(), c.weight._version, c.acr_math) for c in convs)
            if self._frozen is not None and self._frozen[0] == key:
                return self._frozen[1], self._frozen[2], self._frozen[3]
        if torch.cuda.is_current_stream_capturing():
            # a miss of the frozen cache inside a hipGraph capture: the standardisation / image launches upload descriptor
            # tables from temporary pinned memory, which a replay would read again after it was recycled
            raise RuntimeError("ResNetV2: frozen standardised-weight cache missed inside a hipGraph capture")
        # One launch per GROUP of convolutions (stem + stages 0-1 | last stage), not one for all 52 (see forward)
        w_hats, wts, imgs = [None] * len(convs), [None] * len(convs), [None] * len(convs)
        for idx in self._wstd_groups(convs):
            self._standardise_group(convs, idx, w_hats, wts, imgs)
        if frozen:
            self._frozen = (key, w_hats, wts, imgs)
            self.frozen_generation += 1
        return w_hats, wts, imgs

    @staticmethod
    def _standardise_group(convs, idx, w_hats, wts, imgs=None):
        outs = ops.weight_std_all([convs[i].weight for i in idx], convs[0].eps)
        outs_t = ops.WeightStdAllFn.last_transposed
        for j, i in enumerate(idx):
            w_hats[i], wts[i] = outs[j], outs_t[j]
        if imgs is not None:
            # split products: the images of the group's standardised weights -- for the forward AND for the input gradient -- in
            # ONE launch (they were ~130 launches of a few microseconds per step: an image pass and, for the 3x3 weights, a
            # permute copy per convolution and direction)
            specs, owners = [], []
            for i in idx:
                sp = convs[i].image_specs(w_hats[i].detach())
                if sp is not None:
                    owners.append((i, len(specs), len(sp)))
                    specs += list(sp)
            if specs:
                made = ops.x3_image_many(specs, w_hats[idx[0]].device)
                for i, first, n in owners:
                    imgs[i] = tuple(made[first:first + n])

    def _wstd_groups(self, convs):
        """Index lists into ``convs`` (module order): everything up to and including stage 1, and the last stage."""
        last = set(id(m) for m in self.stages[-1].modules() if isinstance(m, StdConv2dSame))
        early = [i for i, c in enumerate(convs) if id(c) not in last]
        late = [i for i, c in enumerate(convs) if id(c) in last]
        return [g for g in (early, late) if g]

    def late_gradient_parameters(self):
        """The parameters whose gradients only exist once backward has run through the whole stem (the convolutions of the stem
        and of every stage but the last: their weight-standardisation backward is one launch at the end): dp.GradSync keeps
        them in a bucket of their own so that nothing else waits for them."""
        last = set(id(p) for p in self.stages[-1].parameters())
        return [p for p in self.parameters() if id(p) not in last]

    def refresh_frozen(self, dtype):
        """Bring the frozen standardised-weight cache up to date on the current stream (no-op while a weight wants a gradient)."""
        convs = [m for m in self.modules() if isinstance(m, StdConv2dSame)]
        w = convs[0].weight
        if self.fused_weight_std and w.is_cuda and w.dtype == dtype and all(c.weight.dtype == dtype for c in convs) \
                and not any(c.weight.requires_grad for c in convs):
            self._standardised(convs, w)

    def forward(self, x, taps=None):
        convs = None
        if self.fused_weight_std and x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and not torch.is_autocast_enabled():
            # one HIP launch standardises all 52 conv weights (and one more in backward) instead of ~10 tiny
            # kernels per convolution and direction
            convs = [m for m in self.modules() if isinstance(m, StdConv2dSame)]
            late = None
            if all(c.weight.dtype == x.dtype for c in convs):
                if any(c.weight.requires_grad for c in convs):
                    # Training: the weight-standardisation launches are autograd nodes, and the engine runs ready nodes in
                    # reverse order of their CREATION.  One launch for all 52 weights at the top of the forward therefore ran
                    # its backward as the very last kernel of the step: every gradient bucket holding a conv weight waited for
                    # it (profiles/r04_gradsync_timeline.json: 86 MB launchable 0.1 ms before backward ended).  Two groups:
                    # stem + all stages but the last here, the last stage's (42 of the 48 MB) right before that stage runs --
                    # its backward launch then follows that stage's backward directly, ~25 ms before the step's end.
                    groups = self._wstd_groups(convs)
                    w_hats, wts, imgs = [None] * len(convs), [None] * len(convs), [None] * len(convs)
                    self._standardise_group(convs, groups[0], w_hats, wts, imgs)
                    late = groups[1] if len(groups) > 1 else None
                else:
                    w_hats, wts, imgs = self._standardised(convs, x)
                for c, w_hat, wt, im in zip(convs, w_hats, wts, imgs):
                    c._w_hat, c._w_hat_t, c._w_imgs = w_hat, wt, im
            else:
                convs = None
        try:
            x = self.stem(x)
            for i, st in enumerate(self.stages):
                if convs is not None and late is not None and i == len(self.stages) - 1:
                    self._standardise_group(convs, late, w_hats, wts, imgs)
                    for j in late:
                        convs[j]._w_hat, convs[j]._w_hat_t, convs[j]._w_imgs = w_hats[j], wts[j], imgs[j]
                x = st(x)
                if taps is not None and i < 2:
                    taps[str(i + 1)] = x           # DPT/vit.py:426-431 forward hooks "1", "2"
        finally:
            if convs is not None:
                for c in convs:
                    c._w_hat = c._w_hat_t = c._w_imgs = None
        return x


class HybridEmbed(nn.Module):
    def __init__(self, backbone, embed_dim):
        super().__init__()
        self.backbone = backbone
        self.proj = nn.Conv2d(backbone.num_features, embed_dim, kernel_size=1, stride=1)


class PatchEmbed(nn.Module):
    def __init__(self, patch, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch, stride=patch)


# ------------------------------------------------------------------------------------------------
# ViT
# ------------------------------------------------------------------------------------------------
class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x, resid=None):
        """fc2(gelu(fc1(x))) [+ resid].  On the bf16 path fc2 runs on the hand-written GEMM with the block's residual
        add fused into its epilogue (ties hipBLASLt on this long-K shape and saves the add kernel); fc1 and the
        input gradients stay on hipBLASLt, which is faster on the short-K / wide-N shapes (scripts/bench_gemm.py)."""
        # fc1: forward and input gradient on hipBLASLt (faster on this short-K / wide-N shape), weight and bias gradient
        # on the hand-written split-M TN GEMM / column-sum kernels (scripts/bench_gemm.py)
        if Mlp.fused and Attention.hip_linear and isinstance(self.act, nn.GELU) and ops.mlp_fusable(x, self.fc1, self.fc2):
            return ops.mlp(x, self.fc1, self.fc2, resid)     # GELU / GELU' inside the GEMM epilogues
        if (Mlp.fused and Attention.hip_linear and isinstance(self.act, nn.GELU) and ops.mlp_f32_usable(x, self.fc1, self.fc2)
                and not torch.is_autocast_enabled()):
            return ops.mlp_f32(x, self.fc1, self.fc2, resid, self.acr_math)  # reference precision: fp32 GEMMs, same fusion
        lib = Mlp.mlp_on_lib                                 # A/B: fc1 forward and the MLP input gradients on hipBLASLt
        h = self.act(ops.linear_or_hip(x, self.fc1, None, Attention.hip_linear, hip_dx=not lib, hip_fwd=not lib, math=self.acr_math))
        return ops.linear_or_hip(h, self.fc2, resid, Attention.hip_linear, hip_dx=not lib, hip_fwd=Mlp.fc2_hip_fwd, math=self.acr_math)

    acr_math = 0            # _lib.MATH code of the fp32 products (set_math)

    fused = True
    mlp_on_lib = False
    fc2_hip_fwd = True     # A/B: fc2 forward on the hand-written GEMM (fused residual)


class Attention(nn.Module):
    """models/vision_transformer.py:167-214 with the (B,H,T,T) softmax never written to HBM.

    ``get_attn()`` / ``get_attn_gradients()`` keep working for code that reaches into the blocks
    (DPT/ACR.py:108-111,182-184): the per-head maps are recomputed on demand from the saved q, k, row
    log-sum-exp (resp. dO, v, plus G/H when the loss used the head-mean maps) by the HIP kernels instead of being retained
    after every forward.  One documented deviation: the reference stores P whenever ``x.requires_grad``
    (vision_transformer.py:207-209), i.e. also in every training step; here a ``train()``-mode forward drops the state
    (it would pin qkv + dO of all 12 layers between steps) unless ``keep_state_in_training`` is set on the module -- then
    the API returns the reference's values after a training backward too (tests/test_model_gpu.py::test_attention_state_api_after_training_backward)."""

    def __init__(self, dim, num_heads):
        super().__init__()
        assert dim % num_heads == 0 and dim // num_heads == ops.HEAD_DIM, "HIP attention is built for head_dim 64"
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        self._saved = None          # (qkv, lse2, heads) of the last forward
        self._saved_do = None       # dO of the last backward
        self._saved_gpm = None      # dLoss/d(head-mean map) of the last backward (None when the loss did not use the maps)
        self._override = {}
        self.last_pm = None         # (B,T,T) head-mean map of the last forward (slice of the MeanStack)

    hip_linear = True       # qkv / proj on the hand-written MFMA GEMMs (acr_linear_bf16 / acr_gemm_f32)
    acr_math = 0            # _lib.MATH code of the fp32 products: Linears and, in training, the attention core (set_math)
    keep_state_in_training = False      # True: get_attn() / get_attn_gradients() also work after a train()-mode forward

    def forward(self, x, stack=None, layer=0, resid=None, x_image=None):
        """Returns proj(attention(qkv(x))) (+ resid when given: the block's residual add is fused into the
        proj GEMM epilogue on the bf16 path).  ``x_image``: x is the output of ops.layer_norm_image (it exists only as that image)."""
        self._override = {}
        qkv = ops.linear_or_hip(x, self.qkv, None, self.hip_linear, math=self.acr_math, x_image=x_image)  # packed (B, T, 3*H*64): no permute copy
        if (self.acr_math == 1 and ops.ATTN_O_IMAGE and self.hip_linear and ops.X3_IMAGES and ops.linear_f32_usable(qkv, self.proj.weight)
                and not torch.is_autocast_enabled() and ops._f32_ok(self.proj.weight, self.proj.bias, resid)):
            # split products: o leaves the attention forward as proj's operand image (no image pass over o)
            o, self.last_pm, oimg = ops.attention_core_oimg(qkv, self.num_heads, stack, layer, self, self.acr_math)
            return ops.linear_or_hip(o, self.proj, resid, self.hip_linear, math=self.acr_math, x_image=oimg)
        o, self.last_pm = ops.attention_core(qkv, self.num_heads, stack, layer, self, self.acr_math)
        return ops.linear_or_hip(o, self.proj, resid, self.hip_linear, math=self.acr_math)

    # -- reference state API (vision_transformer.py:186-196) --
    def get_attn(self):
        if "attn" in self._override:
            return self._override["attn"]
        if self._saved is None:
            return None
        qkv, lse2, heads = self._saved
        return ops.attn_probs(qkv.detach(), lse2, heads)

    def save_attn(self, attn):
        self._override["attn"] = attn

    def get_attn_gradients(self):
        if "grad" in self._override:
            return self._override["grad"]
        if self._saved is None or self._saved_do is None:
            return None
        qkv, _, heads = self._saved
        g = ops.attn_dprobs(qkv.detach(), self._saved_do, heads)
        # what the reference's hook on P holds (vision_transformer.py:207-209): the gradient through attn @ v PLUS, after a
        # training backward, the one through `P.mean(dim=1)` of DPT/ACR.py:109 -- G / H on every head
        if self._saved_gpm is not None:
            g += (self._saved_gpm.detach().float() / heads).unsqueeze(1)
        return g

    def save_attn_gradients(self, g):
        self._override["grad"] = g

    def saved_for_getam(self):
        """(qkv, dO, lse2, heads) of the last forward/backward, for the fused GETAM row kernel."""
        if self._saved is None or self._saved_do is None:
            return None
        qkv, lse2, heads = self._saved
        return qkv.detach(), self._saved_do, lse2, heads


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, stack=None, layer=0):
        math = self.attn.acr_math
        if (Attention.hip_linear and Mlp.fused and isinstance(self.mlp.act, nn.GELU) and ops.ln_image_usable(x, self.norm1, self.attn.qkv, math, self.hip_norm)
                and ops.ln_image_usable(x, self.norm2, self.mlp.fc1, self.mlp.acr_math, self.hip_norm) and ops.mlp_f32_usable(x, self.mlp.fc1, self.mlp.fc2)):
            # split products: LN(x) is read by one Linear only, as an image -- the LayerNorm writes that image, no fp32 copy
            h, skip, hi = ops.layer_norm_image(x, self.norm1)
            x = self.attn(h, stack, layer, resid=skip, x_image=hi)
            h, skip, hi = ops.layer_norm_image(x, self.norm2)
            return ops.mlp_f32(h, self.mlp.fc1, self.mlp.fc2, skip, self.mlp.acr_math, hi)
        h, skip = ops.layer_norm_skip(x, self.norm1, self.hip_norm)         # skip aliases x (gradient fused in LN bwd)
        x = self.attn(h, stack, layer, resid=skip)
        h, skip = ops.layer_norm_skip(x, self.norm2, self.hip_norm)
        return self.mlp(h, resid=skip)

    hip_norm = True         # bf16 mode: LayerNorm on acr_layernorm_*_bf16


class PrefixGraph:
    """Stem + token embedding + blocks[:k] for one input geometry, captured once as a hipGraph and replayed per pass.

    CAM generation (infer_cam.py:123-215) only differentiates blocks >= start_layer; everything below is a fixed chain of
    ~700 small launches per pass that the host cannot issue as fast as the GPU retires them at batch 2.  The graph owns
    its input, its MeanStack (the head-mean maps of blocks < k land in it on every replay) and the tokens it returns;
    the attention state the blocks expose (``get_attn`` / ``last_pm``) is re-pointed to this graph's buffers on replay.
    What the captured launches READ is fixed at capture: parameter storages, the frozen standardised conv weights (a cached
    copy, ResNetV2._standardised) and -- under split products -- the cached weight IMAGES of the prefix Linears
    (ops.weight_image: the warm-up passes fill that cache, the capture hits it).  The graph keeps references to those images
    (they cannot return to the allocator under it) and ``valid`` compares parameter addresses, every parameter's autograd
    VERSION (an in-place update -- load_state_dict into the same storages, an optimizer step in eval mode -- makes the graph
    stale: exact fp32 would have seen it through the storages, the images would not) and the stem cache's generation.
    Writes through ``.data`` are invisible to all of these: ``ACR.invalidate_caches()`` / ``train.refresh_weight_transposes``
    after such a write."""

    def __init__(self, vit, x, k):
        self.k = k
        self.x = x.detach().clone(memory_format=torch.preserve_format)
        b, _, h, w = x.shape
        T = (h // vit.patch_size[1]) * (w // vit.patch_size[0]) + vit.num_tokens
        self.stack = ops.MeanStack(b, vit.depth, T, x.device)
        self.taps = {}
        stem = vit.patch_embed.backbone if isinstance(vit.patch_embed, HybridEmbed) else None
        side = torch.cuda.Stream(device=x.device)          # warm-up off the caller's stream, as torch.cuda.graph asks
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):                 # MIOpen find / library workspaces / the frozen weight cache happen here, uncaptured
                t, _ = vit.embed_tokens(self.x, None)
                vit.run_blocks(t, self.stack, None, 0, k)
        torch.cuda.current_stream(x.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            t, self.res_features = vit.embed_tokens(self.x, self.taps)
            self.out = vit.run_blocks(t, self.stack, self.taps, 0, k)
        self._blocks = list(vit.blocks[:k])
        self.state = [(blk.attn._saved, blk.attn.last_pm) for blk in self._blocks]
        self.stem_generation = stem.frozen_generation if stem is not None else 0
        self.addresses = self._addresses(vit)
        self.versions = self._versions(vit)
        # the weight images the captured products read (cache entries of the prefix Linears as the capture found them)
        self._images = [m.__dict__[key][2] for blk in self._blocks for m in blk.modules() if isinstance(m, nn.Linear)
                        for key in ("_acr_x3_w_img", "_acr_x3_wt_img") if key in m.__dict__]

    @staticmethod
    def _addresses(vit):
        return tuple(p.data_ptr() for p in vit.parameters())

    @staticmethod
    def _versions(vit):
        return tuple(p._version for p in vit.parameters())

    def valid(self, vit):
        stem = vit.patch_embed.backbone if isinstance(vit.patch_embed, HybridEmbed) else None
        if self.addresses != self._addresses(vit) or self.versions != self._versions(vit):
            return False
        if stem is not None:
            # rebuilds the cache when a conv weight changed in place; then the generation moves and the graph is stale
            convs = [m for m in stem.modules() if isinstance(m, StdConv2dSame)]
            if all(c.weight.dtype == self.x.dtype for c in convs):
                stem._standardised(convs, self.x)
            return stem.frozen_generation == self.stem_generation
        return True

    def replay(self, x, taps):
        self.x.copy_(x)
        self.graph.replay()
        for blk, (saved, pm) in zip(self._blocks, self.state):
            blk.attn._saved, blk.attn.last_pm = saved, pm
            blk.attn._saved_do = blk.attn._saved_gpm = None
            blk.attn._override = {}
        if taps is not None:
            taps.update(self.taps)
        return self.out, self.res_features


class PassGraph:
    """One whole GETAM pass for one input geometry as TWO captured hipGraphs that share a memory pool (VERDICT r4 #9b):

      forward graph   the model's ``forward_cam`` on a static input (stem, all blocks, heads; gradients enabled, backward
                      truncated at ``start_layer``) -> class logits, head-mean stack, patch CAMs;
      class graph     d(sum(logits * mask)) / d(tokens entering block start_layer) + ``getam_all`` for a static (samples, C)
                      mask -> the GETAM rows of every sample.  Replayed once per class rank with another mask.

    A one-image pass over four scales used to be ~890 host-side launches (the PrefixGraph only covered the gradient-free
    prefix: every Function of blocks >= start_layer and of each class's backward was issued from Python) with the GPU idle
    13 % of the span; now it is 1 + kmax replays per geometry.  The autograd graph of the captured forward (its saved
    activations live in the pool) is kept alive by ``cls_pred``; the class graph was captured against exactly those buffers.
    Same kernels in the same order as eager launches: results are bit-identical
    (tests/test_model_gpu.py::test_infer_pass_graph_equals_eager_launches).  Validity is PrefixGraph's: parameter addresses
    and versions, the stem cache's generation; the cached weight images the captured products read are held here."""

    def __init__(self, model, inp, start_layer, func):
        vit = model.pretrained.model
        dev = inp.device
        self.inp = inp.detach().clone(memory_format=torch.preserve_format)
        self.mask = torch.zeros((inp.shape[0], model.num_class), dtype=torch.float32, device=dev)
        stem = vit.patch_embed.backbone if isinstance(vit.patch_embed, HybridEmbed) else None

        def forward():
            with torch.enable_grad():
                cls_pred, _, attn, patch_cam = model.forward_cam(self.inp)
            return cls_pred, attn, patch_cam.detach().float(), vit.trunc_input

        def classes(cls_pred, trunc):
            with torch.enable_grad():
                tgt = (cls_pred.float() * self.mask).sum()
                torch.autograd.grad(tgt, trunc, retain_graph=True)
            return model.getam_all(start_layer=start_layer, func=func)

        side = torch.cuda.Stream(device=dev)               # warm-up off the caller's stream, as torch.cuda.graph asks
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):                 # MIOpen find / library workspaces / the frozen weight cache / weight images: uncaptured
                cls_pred, _, _, trunc = forward()
                classes(cls_pred, trunc)
            del cls_pred, trunc
        torch.cuda.current_stream(dev).wait_stream(side)
        self.fwd_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd_graph):
            self.cls_pred, self.attn, self.patch, self.trunc = forward()
        self.cls_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.cls_graph, pool=self.fwd_graph.pool()):
            self.rows = classes(self.cls_pred, self.trunc)
        self._blocks = list(vit.blocks)
        self.state = [(blk.attn._saved, blk.attn.last_pm, blk.attn._saved_do) for blk in self._blocks]
        self.taps = dict(model.pretrained.activations)
        self.stem_generation = stem.frozen_generation if stem is not None else 0
        self.addresses = tuple(p.data_ptr() for p in model.parameters())
        self.versions = tuple(p._version for p in model.parameters())
        self._images = [m.__dict__[key][2] for m in model.modules() if isinstance(m, nn.Linear)
                        for key in ("_acr_x3_w_img", "_acr_x3_wt_img") if key in m.__dict__]

    def valid(self, model):
        vit = model.pretrained.model
        stem = vit.patch_embed.backbone if isinstance(vit.patch_embed, HybridEmbed) else None
        if (self.addresses != tuple(p.data_ptr() for p in model.parameters())
                or self.versions != tuple(p._version for p in model.parameters())):
            return False
        if stem is not None:
            convs = [m for m in stem.modules() if isinstance(m, StdConv2dSame)]
            if all(c.weight.dtype == self.inp.dtype for c in convs):
                stem._standardised(convs, self.inp)
            return stem.frozen_generation == self.stem_generation
        return True

    def run(self, model, inp, masks):
        """Replay the pass on ``inp`` and the class graph once per mask.  Returns (patch CAMs, head-mean stack, [rows per mask]):
        the stack is the graph's own buffer (valid until the next replay of this geometry ON THE SAME STREAM), the others are
        copies."""
        self.inp.copy_(inp)
        self.fwd_graph.replay()
        rows = []
        for m in masks:
            self.mask.copy_(m)
            self.cls_graph.replay()
            rows.append(self.rows.clone())
        for blk, (saved, pm, sdo) in zip(self._blocks, self.state):      # the attention-state API reads this pass
            blk.attn._saved, blk.attn.last_pm, blk.attn._saved_do = saved, pm, sdo
            blk.attn._saved_gpm = None
            blk.attn._override = {}
        model.pretrained.activations.clear()
        model.pretrained.activations.update(self.taps)
        return self.patch.clone(), self.attn, rows


def pass_graph(model, inp, start_layer, func):
    """The PassGraph for passes shaped like ``inp`` -- or None when such a pass cannot be replayed (a parameter or the input wants
    a gradient, a kernel timer is recording, the switch is off, or an earlier capture of this geometry failed)."""
    vit = model.pretrained.model
    if (not getattr(vit, "graph_pass", False) or not start_layer or not inp.is_cuda or inp.requires_grad or ops.KERNEL_TIMER is not None
            or torch.is_autocast_enabled() or torch.cuda.is_current_stream_capturing() or model.truncate_at != start_layer
            or any(p.requires_grad for p in model.parameters())):
        return None
    key = (tuple(inp.shape), inp.dtype, inp.device, inp.is_contiguous(memory_format=torch.channels_last), start_layer, func, model.training,
           vit.acr_math, vit.blocks[0].attn.acr_math, model.num_class)
    cache = vit.__dict__.setdefault("_pass_graphs", OrderedDict())
    g = cache.get(key)
    if g is not None and g is not False and not g.valid(model):
        del cache[key]
        g = None
    if g is None and not _sighted(vit, key):
        return None                            # first pass of this geometry: eager launches (see VisionTransformer.graph_sightings)
    if g is None:
        keep = vit.graph_prefix
        vit.graph_prefix = False               # the pass is captured whole: no nested replay of a prefix graph
        try:
            g = PassGraph(model, inp, start_layer, func)
        except RuntimeError as e:              # an op that cannot be captured on this build: stay on eager launches
            import traceback
            import warnings
            where = [l.strip() for l in traceback.format_exc().splitlines() if l.strip().startswith("File")][-3:]
            warnings.warn("hipGraph capture of the GETAM pass failed (%s; %s); running it eagerly" % (str(e).splitlines()[0], " <- ".join(reversed(where))))
            torch.cuda.synchronize()
            g = False
        finally:
            vit.graph_prefix = keep
        cache[key] = g
        while len(cache) > vit.max_pass_graphs:
            cache.popitem(last=False)
    cache.move_to_end(key)
    return g or None


def _sighted(vit, key):
    """True once ``key`` (an input geometry) has been asked for ``vit.graph_sightings`` times: a capture costs two warm-up passes,
    the capture itself and a private pool holding the pass's activations -- about four passes of work -- so a geometry seen ONCE
    (natural-size images, the last short batch of a list) runs eagerly and only a recurring one is captured.  The
    record survives a graph's invalidation (a weight update re-captures at once) and is bounded."""
    seen = vit.__dict__.setdefault("_graph_sightings", OrderedDict())
    n = seen.get(key, 0) + 1
    seen[key] = n
    seen.move_to_end(key)
    while len(seen) > 64:
        seen.popitem(last=False)
    return n >= vit.graph_sightings


class _PosEmbedResizeFn(torch.autograd.Function):
    """F.interpolate(grid, (gs_h, gs_w), mode="bilinear") of the position-embedding grid (vision_transformer.py:490-504) with a
    DETERMINISTIC backward.  The stock backward scatters with float atomics: when several output pixels read one input pixel
    (any up-scaling -- 24^2 -> 28^2 at the BASELINE geometry) the gradient of `pos_embed` came out in a different summation
    order from run to run: the one tensor of a 448^2 step that did not repeat bit for bit (scripts/lab/step_repeat.py, round
    6).  Here every INPUT pixel gathers its (at most `m`) contributing output pixels through a table built once per geometry
    from the operator itself (the images of the unit vectors under the very same F.interpolate) and sums them in ascending
    output order.  Forward values are the stock op's, bit for bit."""
    _tables = {}

    @staticmethod
    def tables(gs_old_h, gs_old_w, gs_h, gs_w, device, dtype):
        key = (gs_old_h, gs_old_w, gs_h, gs_w, str(device), dtype)
        t = _PosEmbedResizeFn._tables.get(key)
        if t is None:
            n_in = gs_old_h * gs_old_w
            with torch.no_grad():
                eye = torch.eye(n_in, device=device, dtype=torch.float32).reshape(n_in, 1, gs_old_h, gs_old_w)
                R = F.interpolate(eye, size=(gs_h, gs_w), mode="bilinear").reshape(n_in, gs_h * gs_w)       # R[input pixel, output pixel]
                nz = R != 0
                m = max(1, int(nz.sum(1).max()))
                # stable: the non-zero columns of every row first, in ascending output order
                idx = torch.sort(nz.to(torch.int8), dim=1, descending=True, stable=True).indices[:, :m].contiguous()
                w = R.gather(1, idx).to(dtype)                                                           # zero weight where a row has fewer than m
            t = _PosEmbedResizeFn._tables[key] = (idx, w)
        return t

    @staticmethod
    def forward(ctx, grid, gs_h, gs_w):
        ctx.geom = (grid.shape[2], grid.shape[3], gs_h, gs_w)
        return F.interpolate(grid, size=(gs_h, gs_w), mode="bilinear")

    @staticmethod
    def backward(ctx, g):
        oh, ow, gs_h, gs_w = ctx.geom
        idx, w = _PosEmbedResizeFn.tables(oh, ow, gs_h, gs_w, g.device, g.dtype)
        b, c = g.shape[0], g.shape[1]
        gf = g.reshape(b * c, gs_h * gs_w)
        gi = (gf[:, idx] * w).sum(-1)                        # (b c, input pixels, m) -> fixed-order sum over m
        return gi.reshape(b, c, oh, ow), None, None


class VisionTransformer(nn.Module):
    def __init__(self, embed_dim=768, depth=12, num_heads=12, hybrid=True, patch=16, img_size=384,
                 num_classes=1000, distilled=False, in_chans=3):
        super().__init__()
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.num_tokens = 2 if distilled else 1
        self.start_index = self.num_tokens
        self.patch_size = [16, 16]
        if hybrid:
            self.patch_embed = HybridEmbed(ResNetV2(in_chans=in_chans), embed_dim)
        else:
            self.patch_embed = PatchEmbed(patch, in_chans, embed_dim)
        n_patches = (img_size // 16) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.bkg_token = nn.Parameter(torch.zeros(1, 1, embed_dim))          # vision_transformer.py:307 (unused)
        self.dist_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if distilled else None
        self.pos_embed = nn.Parameter(torch.zeros(1, n_patches + self.num_tokens, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes)                        # in the layout, never used by ACR
        self.head_dist = nn.Linear(embed_dim, num_classes) if distilled else None
        self._init_weights()

    def _init_weights(self):
        for t in (self.pos_embed, self.cls_token, self.bkg_token):
            nn.init.trunc_normal_(t, std=0.02)
        if self.dist_token is not None:
            nn.init.trunc_normal_(self.dist_token, std=0.02)
        for name, m in self.named_modules():
            if name.startswith("patch_embed.backbone"):
                continue
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Conv2d):
                fan_in = m.weight[0].numel()
                nn.init.trunc_normal_(m.weight, std=math.sqrt(1.0 / fan_in) / 0.87962566103423978)
                nn.init.zeros_(m.bias)

    def _resize_pos_embed(self, posemb, gs_h, gs_w):
        tok, grid = posemb[:, :self.start_index], posemb[0, self.start_index:]
        gs_old = int(math.sqrt(grid.shape[0]))
        if gs_old == gs_h and gs_old == gs_w:
            return posemb
        grid = grid.reshape(1, gs_old, gs_old, -1).permute(0, 3, 1, 2)
        grid = _PosEmbedResizeFn.apply(grid, gs_h, gs_w)
        grid = grid.permute(0, 2, 3, 1).reshape(1, gs_h * gs_w, -1)
        return torch.cat([tok, grid], dim=1)

    def embed_tokens(self, x, taps=None):
        """Stem (hybrid) or patch convolution, class/distillation tokens, position embedding: (B,3,h,w) -> ((B,T,D), stem features)."""
        b, c, h, w = x.shape
        if isinstance(self.patch_embed, HybridEmbed):
            x = self.patch_embed.backbone(x, taps)
        res_features = x
        # (after the stem on purpose: the autograd engine runs ready nodes in reverse order of their creation -- resized before
        # the stem, the position embedding's gradient was the LAST one of the whole backward to arrive and held its 17 MB
        # gradient bucket until the step's end, scripts/lab/grad_arrival_order.py)
        pos = self._resize_pos_embed(self.pos_embed, h // self.patch_size[1], w // self.patch_size[0])
        pe = self.patch_embed.proj
        if (isinstance(self.patch_embed, HybridEmbed) and StdConv2dSame.hip_1x1 and pe.bias is not None
                and ops.conv1x1_fusable(x, pe.weight, pe.stride[0])):
            x = ops.conv1x1(x, pe.weight, None, self.acr_math)                                   # 1024 -> 768 projection on the NCHW GEMM kernels
            prefix = self.cls_token[0] if self.dist_token is None else torch.cat([self.cls_token[0], self.dist_token[0]], 0)
            if x.shape[2] * x.shape[3] + self.num_tokens == pos.shape[1] and ops.tokens_fusable(x, pe.bias, prefix, pos):
                # bias add, transpose to token-major, class token, position embedding: one pass (and one backward pass) instead of four
                return ops.tokens(x, pe.bias, prefix, pos), res_features
            x = x + pe.bias.view(1, -1, 1, 1)
        else:
            x = pe(x)
        x = x.flatten(2).transpose(1, 2)
        if x.shape[1] + self.num_tokens != pos.shape[1]:
            # the reference fails at the addition below with a bare size mismatch (vision_transformer.py:449-467 sizes the
            # position embedding with h // 16 while the SAME-padded stem yields ceil(h / 16) rows)
            raise ValueError("input %dx%d is not a multiple of the %d-pixel patch: %d patch tokens but a %d-entry position "
                             "embedding" % (h, w, self.patch_size[0], x.shape[1], pos.shape[1] - self.num_tokens))
        toks = [self.cls_token.expand(b, -1, -1)]
        if self.dist_token is not None:
            toks.append(self.dist_token.expand(b, -1, -1))
        return torch.cat(toks + [x], dim=1) + pos, res_features

    def run_blocks(self, x, stack, taps, lo, hi):
        for i in range(lo, hi):
            x = self.blocks[i](x, stack, i)
            if taps is not None:
                if i == self.tap3:
                    taps["3"] = x
                if i == self.tap4:
                    taps["4"] = x
        return x

    def forward_flex(self, x, stack=None, taps=None, truncate_at=None, prefix=None):
        """models/vision_transformer.py:449-486.  ``stack`` (ops.MeanStack) receives the head-mean maps,
        ``taps`` the DPT activations dict; ``truncate_at`` = k detaches the tokens entering block k so a
        later backward stops there (GETAM only needs gradients of blocks >= start_layer).  ``prefix`` (a PrefixGraph
        from ``prefix_graph``, whose MeanStack must be the ``stack`` passed here) replays everything below block k as
        one captured hipGraph instead of launching it kernel by kernel."""
        k = 0 if truncate_at is None else truncate_at
        if prefix is not None:
            assert truncate_at is not None and prefix.k == truncate_at and stack is prefix.stack
            x, res_features = prefix.replay(x, taps)
        else:
            x, res_features = self.embed_tokens(x, taps)
            x = self.run_blocks(x, stack, taps, 0, k)
        if truncate_at is not None:
            x = x.detach().requires_grad_(True)
            self.trunc_input = x
        x = self.run_blocks(x, stack, taps, k, len(self.blocks))
        return self.norm(x) if truncate_at is None else None, res_features

    acr_math = 0            # _lib.MATH code of the patch-embedding projection's fp32 products (set_math)
    graph_prefix = os.environ.get("ACR_INFER_GRAPH", "1") != "0"      # A/B: hipGraph replay of the gradient-free prefix
    graph_pass = os.environ.get("ACR_INFER_PASS_GRAPH", "1") != "0"   # A/B: whole GETAM passes as captured graphs (PassGraph)
    max_prefix_graphs = 8   # LRU bound of the captured prefixes ...
    max_pass_graphs = 8     # ... and, separately, of the captured whole passes (each holds its forward's saved activations)
    graph_sightings = 2     # a geometry is captured the second time it is seen (1 = at once); variable-size inputs: INTEGRATION.md

    def train(self, mode=True):
        if mode:                               # back to training: the captured prefixes' private pools (activations of every
            self.__dict__.pop("_prefix_graphs", None)      # geometry seen) go back to the allocator
            self.__dict__.pop("_pass_graphs", None)
            self.__dict__.pop("_graph_sightings", None)
        return super().train(mode)

    def prefix_graph(self, x, k):
        """The PrefixGraph for inputs shaped like ``x`` and a backward truncated at block ``k`` -- or None when the prefix
        cannot be replayed: a parameter or the input wants a gradient, a kernel timer is recording, k == 0, the switch is
        off, or an earlier capture of this geometry failed."""
        if (not self.graph_prefix or not k or not x.is_cuda or x.requires_grad or ops.KERNEL_TIMER is not None
                or torch.is_autocast_enabled() or torch.cuda.is_current_stream_capturing()
                or any(p.requires_grad for p in self.parameters())):
            return None
        key = (tuple(x.shape), x.dtype, x.device, x.is_contiguous(memory_format=torch.channels_last), k, self.training,
               self.blocks[0].attn.keep_state_in_training, self.acr_math, self.blocks[0].attn.acr_math)
        cache = self.__dict__.setdefault("_prefix_graphs", OrderedDict())
        g = cache.get(key)
        if g is not None and g is not False and not g.valid(self):
            del cache[key]                     # the parameters moved (.to / .float / load into new storage): capture again
            g = None
        if g is None and not _sighted(self, ("prefix",) + key):
            return None
        if g is None:
            try:
                g = PrefixGraph(self, x, k)
            except RuntimeError as e:          # an op that cannot be captured on this build: stay on eager launches
                import warnings
                warnings.warn("hipGraph capture of the inference prefix failed (%s); running it eagerly" % str(e).splitlines()[0])
                torch.cuda.synchronize()
                g = False
            cache[key] = g
            while len(cache) > self.max_prefix_graphs:
                cache.popitem(last=False)
        cache.move_to_end(key)
        return g or None

    tap3, tap4 = 8, 11
