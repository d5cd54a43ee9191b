"""Kernel-level parity on the GPU: every C-ABI entry point against plain torch math (fp64 on device) on
seeded inputs, incl. ragged sizes (T not a multiple of any tile), the maximum T of the inference config and
the edge cases the path has (p = 1 -> T = 2, masked tails, no head-mean gradient)."""
import math

import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _ref_attn(qkv, H, want_grad=False):
    B, T, _ = qkv.shape
    q, k, v = qkv.double().reshape(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    P = ((q @ k.transpose(-2, -1)) * 64 ** -0.5).softmax(-1)
    o = (P @ v).transpose(1, 2).reshape(B, T, H * 64)
    return o, P


def _flip_perm(p, dev):
    return torch.arange(p * p, device=dev).reshape(p, p).flip(1).reshape(-1)


@pytest.mark.parametrize("B,L,p", [(1, 1, 1), (2, 3, 4), (1, 12, 14), (2, 2, 28), (1, 2, 37), (1, 3, 32)])   # p = 32: COCO 512^2
def test_consistency(B, L, p):
    from acr_wsss_amd import ops
    dev = _dev()
    T = p * p + 1
    g = torch.Generator(device="cpu").manual_seed(p * 7 + B)
    a = torch.rand(2 * B, L, T, T, generator=g).to(dev)
    a[0, 0, 1:, 1:] = a[B, 0, 1:, 1:][_flip_perm(p, dev)][:, _flip_perm(p, dev)]   # exact zeros in d -> sign 0
    a.requires_grad_(True)
    cls, aff = ops.consistency(a, p)
    w = torch.tensor([1.7, -0.6], device=dev)
    (cls * w[0] + aff * w[1]).backward()
    ad = a.detach().double().requires_grad_(True)
    pi = _flip_perm(p, dev)
    a1, a2 = ad[:B], ad[B:]
    rc = (a1[:, :, 0, 1:] - a2[:, :, 0, 1:][:, :, pi]).abs().mean()
    ra = (a1[:, :, 1:, 1:] - a2[:, :, 1:, 1:][:, :, pi][:, :, :, pi]).abs().mean()
    (rc * w[0].double() + ra * w[1].double()).backward()
    assert abs(float(cls) - float(rc)) <= 2e-6 * abs(float(rc)) + 1e-9
    assert abs(float(aff) - float(ra)) <= 2e-6 * abs(float(ra)) + 1e-9
    torch.testing.assert_close(a.grad.double(), ad.grad, rtol=1e-5, atol=1e-12)
    # in-place-flip form of the reference (train_acr.py:151-158) gives the same numbers
    b2 = a.detach()[B:].clone()
    c2, f2 = b2[:, :, 0, 1:].unsqueeze(2), b2[:, :, 1:, 1:]
    for i in range(p):
        c2[:, :, :, i * p:i * p + p] = c2[:, :, :, i * p:i * p + p].flip(3)
    for i in range(p):
        f2[:, :, i * p:i * p + p, :] = f2[:, :, i * p:i * p + p, :].flip(2)
    for i in range(p):
        f2[:, :, :, i * p:i * p + p] = f2[:, :, :, i * p:i * p + p].flip(3)
    ref_aff = torch.nn.functional.l1_loss(a.detach()[:B, :, 1:, 1:], f2)
    assert abs(float(aff) - float(ref_aff)) <= 5e-6 * abs(float(ref_aff)) + 1e-9


def test_consistency_full_size_is_a_mean_over_samples():
    """The loss at BASELINE size (16 images x 2 views, 12 layers, T = 785: a 946 MB stack) through a size-independent property:
    both terms are means over equally sized per-sample blocks, so they equal the mean of the 16 one-sample results, and sample
    i's gradient is 1/16 of its one-sample gradient (checked on the first and the last sample)."""
    from acr_wsss_amd import ops
    dev = _dev()
    B, L, p = 16, 12, 28
    T = p * p + 1
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.rand(2 * B, L, T, T, generator=g).to(dev).requires_grad_(True)
    cls, aff = ops.consistency(a, p)
    (cls * 1.7 - aff * 0.6).backward()
    singles = []
    for i in range(B):
        ai = torch.stack([a.detach()[i], a.detach()[B + i]]).requires_grad_(i in (0, B - 1))
        ci, fi = ops.consistency(ai, p)
        if ai.requires_grad:
            (ci * 1.7 - fi * 0.6).backward()
            for view, row in ((0, i), (1, B + i)):
                torch.testing.assert_close(a.grad[row], ai.grad[view] / B, rtol=1e-5, atol=1e-12)
        singles.append((float(ci.detach()), float(fi.detach())))
    mc, mf = sum(c for c, _ in singles) / B, sum(f for _, f in singles) / B
    assert abs(float(cls.detach()) - mc) <= 2e-6 * abs(mc) and abs(float(aff.detach()) - mf) <= 2e-6 * abs(mf)


@pytest.mark.parametrize("B,T,H", [(2, 2, 1), (1, 17, 12), (2, 197, 3), (1, 785, 12), (1, 1025, 2),
                                   (1, 2305, 3), (1, 3137, 1)])          # multi-scale inference: 768^2 and 896^2
@pytest.mark.parametrize("with_g", [True, False])
@pytest.mark.parametrize("gen", ["scores", "recompute", "split"])
def test_attention_f32(B, T, H, with_g, gen, monkeypatch):
    """The fp32 generations (resident scores: csrc/attn_f32_sres.hip, the default; recompute: csrc/attn_f32_dma.hip; split:
    csrc/attn_f32_x3.hip -- resident scores with S, PV, dP, dQ, dK, dV as six bf16-MFMA terms of a three-way operand split,
    math = "f32_split") against fp64 math of models/vision_transformer.py:203-211 + the head mean of DPT/ACR.py:107-112, all
    three at the SAME tolerances."""
    from acr_wsss_amd import ops
    monkeypatch.setattr(ops, "ATTN_F32_SCORES", gen != "recompute")
    math = 1 if gen == "split" else 0
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(T * 3 + H)
    qkv = (1.5 * torch.randn(B, T, 3 * H * 64, generator=g)).to(dev).requires_grad_(True)
    Ly = 2
    stack = ops.MeanStack(B, Ly, T, dev)
    stack.buf.fill_(float("nan"))
    o, pm = ops.attention_core(qkv, H, stack, 1, None, math)
    d_o = torch.randn(B, T, H * 64, generator=g).to(dev)
    gpm = torch.randn(B, T, T, generator=g).to(dev) if with_g else None
    loss = (o * d_o).sum() + ((pm * gpm).sum() if with_g else 0.0)
    loss.backward()
    if gen == "split":                                       # deterministic: a second run repeats bit for bit
        q2 = qkv.detach().clone().requires_grad_(True)
        stack2 = ops.MeanStack(B, Ly, T, dev)
        o2, pm2 = ops.attention_core(q2, H, stack2, 1, None, math)
        ((o2 * d_o).sum() + ((pm2 * gpm).sum() if with_g else 0.0)).backward()
        assert torch.equal(o2, o) and torch.equal(pm2, pm) and torch.equal(q2.grad, qkv.grad)

    qd = qkv.detach().double().requires_grad_(True)
    o_ref, P = _ref_attn(qd, H)
    pm_ref = P.mean(1)
    loss_ref = (o_ref * d_o.double()).sum() + ((pm_ref * gpm.double()).sum() if with_g else 0.0)
    loss_ref.backward()
    torch.testing.assert_close(o.double(), o_ref, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(pm.double(), pm_ref, rtol=1e-4, atol=1e-7)
    assert torch.isnan(stack.buf[:, 0]).all()            # the other layer's slice is untouched
    scale = qd.grad.abs().max()
    assert (qkv.grad.double() - qd.grad).abs().max() <= 3e-5 * scale, (qkv.grad.double() - qd.grad).abs().max() / scale


@pytest.mark.parametrize("math", [0, 1])
@pytest.mark.parametrize("B,T,H", [(2, 785, 12), (1, 197, 4), (2, 64, 8), (1, 2305, 4), (1, 577, 24)])
def test_attention_delta_kernels_agree(B, T, H, math):
    """The delta pass of the resident-score backward (delta_i = O_i . dO_i + 1/H sum_j P_ij G_ij) with four heads per workgroup and
    the gradient block staged through LDS (the default when H % 4 == 0) does the same operations in the same order as the
    one-wave-per-head kernel it replaced: the whole backward repeats bit for bit under the A/B switch."""
    from acr_wsss_amd import ops, _lib
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(T + H)
    qkv0 = (1.5 * torch.randn(B, T, 3 * H * 64, generator=g)).to(dev)
    d_o = torch.randn(B, T, H * 64, generator=g).to(dev)
    gpm = torch.randn(B, T, T, generator=g).to(dev)
    grads = []
    try:
        for one_head in (0, 1):
            _lib.set_option("attn_delta_1head", one_head)
            qkv = qkv0.clone().requires_grad_(True)
            stack = ops.MeanStack(B, 1, T, dev)
            o, pm = ops.attention_core(qkv, H, stack, 0, None, math)
            ((o * d_o).sum() + (pm * gpm).sum()).backward()
            grads.append(qkv.grad)
    finally:
        _lib.set_option("attn_delta_1head", 0)
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize("B,T,H", [(1, 2, 1), (2, 17, 2), (2, 197, 3), (1, 785, 12), (1, 1025, 2)])
@pytest.mark.parametrize("with_g", [True, False])
@pytest.mark.parametrize("f32math", [False, True])
def test_attention_bf16(B, T, H, with_g, f32math):
    """bf16 tensors: the bf16-MFMA kernels (training precision) and the exact-fp32-math kernels with bf16 I/O,
    both against fp64 math on the same bf16-rounded inputs.  Tolerances: bf16 has 8 significant bits; outputs are
    rounded once (rel 2^-9), P/dS are rounded to bf16 before the second products on the MFMA path."""
    from acr_wsss_amd import ops, _lib
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(T * 5 + H)
    qkv = torch.randn(B, T, 3 * H * 64, generator=g).to(dev).bfloat16().requires_grad_(True)
    d_o = torch.randn(B, T, H * 64, generator=g).to(dev).bfloat16()
    gpm = (torch.randn(B, T, T, generator=g) * 0.5).to(dev) if with_g else None
    stack = ops.MeanStack(B, 1, T, dev)
    _lib.BF16_F32MATH = f32math
    try:
        o, pm = ops.attention_core(qkv, H, stack, 0, None)
        ((o.float() * d_o.float()).sum() + ((pm * gpm).sum() if with_g else 0.0)).backward()
    finally:
        _lib.BF16_F32MATH = False
    qd = qkv.detach().double().requires_grad_(True)
    o_ref, P = _ref_attn(qd, H)
    ((o_ref * d_o.double()).sum() + ((P.mean(1) * gpm.double()).sum() if with_g else 0.0)).backward()
    assert (o.double() - o_ref).abs().max() <= 1.5e-2 * o_ref.abs().max()
    torch.testing.assert_close(pm.double(), P.mean(1), rtol=2e-3, atol=1e-6)
    err = (qkv.grad.double() - qd.grad).abs()
    scale = qd.grad.abs().max()
    assert err.max() <= (1e-2 if f32math else 2.5e-2) * scale, float(err.max() / scale)
    assert err.mean() <= (1.5e-3 if f32math else 3e-3) * scale, float(err.mean() / scale)


@pytest.mark.parametrize("B,T,dtype,math", [(32, 785, torch.float32, 0), (32, 785, torch.float32, 1), (32, 785, torch.bfloat16, 0),
                                             (16, 2305, torch.float32, 0), (16, 2305, torch.float32, 1)])
def test_attention_full_size_is_per_sample(B, T, dtype, math):
    """BASELINE-size launches (32 views x 12 heads x 785 tokens: one training step; 16 samples x 2305 tokens: eight images at scale 2,
    a 4 GB score buffer, byte offsets beyond 2^32) checked through a size-independent property: attention is per sample, so the
    LAST sample of the big batch must come out exactly as when it is run alone -- forward, head mean and every gradient.  (The
    small-shape tests above pin the values against fp64; this pins the indexing at full size.)  math = 1: the split-product
    kernels the bench's headline launches (their bf16 planes live BEHIND the score buffer: offsets past 983 MB resp. 4 GB)."""
    from acr_wsss_amd import ops
    dev = _dev()
    H = 12
    g = torch.Generator(device="cpu").manual_seed(B + T)
    qkv = (1.5 * torch.randn(B, T, 3 * H * 64, generator=g)).to(dev).to(dtype)
    d_o = torch.randn(B, T, H * 64, generator=g).to(dev).to(dtype)
    gst = torch.zeros(B, T, ops.pad4(T), device=dev)
    gst[:, :, :T] = torch.randn(B, T, T, generator=g).to(dev) * 1e-2
    outs = []
    for sl in (slice(0, B), slice(B - 1, B)):
        x = qkv[sl].clone().requires_grad_(True)
        n = x.shape[0]
        stack = ops.MeanStack(n, 1, T, dev)
        o, pm = ops.attention_core(x, H, stack, 0, None, math)
        assert o.grad_fn.math == math
        torch.autograd.backward([o, pm], [d_o[sl].contiguous(), gst[sl][:, :, :T]])
        outs.append((o.detach()[-1].clone(), pm.detach()[-1].clone(), x.grad[-1].clone()))
        del x, o, pm, stack
        torch.cuda.empty_cache()
    for name, a, b in zip(("o", "head mean", "dqkv"), outs[0], outs[1]):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), (name, float((a.float() - b.float()).abs().max()))


def test_probs_dprobs_bf16():
    from acr_wsss_amd import ops
    dev = _dev()
    B, T, H = 2, 145, 12
    g = torch.Generator(device="cpu").manual_seed(12)
    qkv = torch.randn(B, T, 3 * H * 64, generator=g).to(dev).bfloat16().requires_grad_(True)
    o, _ = ops.attention_core(qkv, H, None, 0, None)
    d_o = torch.randn(B, T, H * 64, generator=g).to(dev).bfloat16()
    lse2 = o.grad_fn.saved_tensors[2]
    P = ops.attn_probs(qkv.detach(), lse2, H)
    dP = ops.attn_dprobs(qkv.detach(), d_o, H)
    _, P_ref = _ref_attn(qkv.detach(), H)
    v = qkv.detach().double().reshape(B, T, 3, H, 64)[:, :, 2].permute(0, 2, 1, 3)
    dP_ref = d_o.double().reshape(B, T, H, 64).permute(0, 2, 1, 3) @ v.transpose(-2, -1)
    torch.testing.assert_close(P.double(), P_ref, rtol=2e-3, atol=1e-6)
    torch.testing.assert_close(dP.double(), dP_ref, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("B,T,H", [(2, 145, 12), (1, 577, 12), (1, 2305, 4), (1, 3137, 2)])
def test_probs_dprobs_getam_row(B, T, H):
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(11)
    qkv = torch.randn(B, T, 3 * H * 64, generator=g).to(dev).requires_grad_(True)
    o, _ = ops.attention_core(qkv, H, None, 0, None)
    d_o = torch.randn(B, T, H * 64, generator=g).to(dev)
    lse2 = o.grad_fn.saved_tensors[2]
    P = ops.attn_probs(qkv.detach(), lse2, H)
    dP = ops.attn_dprobs(qkv.detach(), d_o, H)
    _, P_ref = _ref_attn(qkv.detach(), H)
    v = qkv.detach().double().reshape(B, T, 3, H, 64)[:, :, 2].permute(0, 2, 1, 3)
    dP_ref = d_o.double().reshape(B, T, H, 64).permute(0, 2, 1, 3) @ v.transpose(-2, -1)
    torch.testing.assert_close(P.double(), P_ref, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(dP.double(), dP_ref, rtol=1e-4, atol=1e-4)
    for func in ("grad", "cam_grad", "grad_s", "cam_grad_s"):
        for batch in range(B):
            row = torch.zeros(T, device=dev)
            ops.getam_row_accum(qkv.detach(), d_o, lse2, H, batch, func, row)
            ops.getam_row_accum(qkv.detach(), d_o, lse2, H, batch, func, row)      # accumulates
            gr, cm = dP_ref[batch], P_ref[batch]
            mg = gr.clamp(min=0).mean(0)
            mcg = (gr * cm).clamp(min=0).mean(0)
            ref = {"grad": mg, "cam_grad": mcg, "grad_s": mg * mg, "cam_grad_s": mcg * mg}[func][0] * 2
            torch.testing.assert_close(row.double(), ref, rtol=1e-4, atol=1e-6 * float(ref.abs().max()))


@pytest.mark.parametrize("C", [20, 80])                     # 80: COCO (train_acr_coco.py:91)
def test_cam_readouts(C):
    from acr_wsss_amd import ops
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(3)
    N, D = 36, 768
    x = torch.randn(1 + N, D, generator=g).to(dev)
    w = (torch.randn(C, D, generator=g) * D ** -0.5).to(dev)
    bias = torch.randn(C, generator=g).to(dev)
    pc = ops.patch_cam(x[1:], w, bias)
    torch.testing.assert_close(pc, F.relu(F.linear(x[1:], w, bias)), rtol=1e-4, atol=1e-5)
    lab = torch.zeros(C, device=dev)
    lab[[3, 11, C - 1]] = 1.0
    for (oh, ow) in ((75, 61), (6, 6), (5, 13), (1, 1)):
        ref = F.interpolate(pc.t().reshape(1, C, 6, 6), (oh, ow), mode="bilinear", align_corners=False)[0]
        got = ops.bilinear_resize(pc, (oh, ow), False, chan_mul=lab, hflip=True, channels_last=False) if False else \
            ops.bilinear_resize(pc.reshape(6, 6, C), (oh, ow), False, chan_mul=lab, hflip=True, channels_last=True)
        torch.testing.assert_close(got, (ref * lab.view(C, 1, 1)).flip(-1), rtol=1e-5, atol=1e-6)
        src = torch.rand(2, 6, 6, generator=g).to(dev)
        ref = F.interpolate(src[None], (oh, ow), mode="bilinear", align_corners=True)[0]
        acc = torch.ones(2, oh, ow, device=dev)
        ops.bilinear_resize(src, (oh, ow), True, out=acc)
        torch.testing.assert_close(acc, ref + 1.0, rtol=1e-5, atol=1e-6)
    for Ly, T, n in ((12, 37, 11), (12, 577, 3), (12, 2305, 2), (2, 3137, 1)):
        a = torch.rand(Ly, T, T, generator=g).to(dev)
        cams = torch.rand(n, T - 1, generator=g).to(dev)
        out = ops.aff_refine(a, cams)
        ref = (a[:, 1:, 1:].sum(0).double() @ cams.double().t()).t()
        torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-6 * T)


def test_errors_are_loud():
    from acr_wsss_amd import _lib, ops
    with pytest.raises(_lib.AcrHipError):
        ops.attention_core(torch.randn(1, 5, 192), 1, None, 0, None)          # CPU tensor: no fallback
    dev = _dev()
    with pytest.raises(_lib.AcrHipError):
        ops.attention_core(torch.randn(1, 5, 192, device=dev).half(), 1, None, 0, None)
    lib = _lib.load()
    rc = lib.acr_consistency_fwd(None, None, 0, 1, 1, 5, 2, None, None, None)
    assert rc < 0 and b"null" in lib.acr_last_error()
    # a direct C-ABI caller handing an offset (4-byte aligned) view to a kernel that moves 16 bytes per lane gets an error,
    # not a GPU memory fault (the fp32 LayerNorm entry points did not check)
    M, C = 8, 768
    buf = torch.randn(M * C + 4, device=dev)
    x, xo = buf[:M * C], buf[1:M * C + 1]
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    y, stats = torch.empty(M * C, device=dev), torch.empty(2 * M, device=dev)
    st = _lib.stream_ptr()
    assert lib.acr_layernorm_fwd_f32(_lib.ptr(x), _lib.ptr(gam), _lib.ptr(bet), _lib.ptr(y), _lib.ptr(stats), M, C, 1e-6, st) == 0
    rc = lib.acr_layernorm_fwd_f32(_lib.ptr(xo), _lib.ptr(gam), _lib.ptr(bet), _lib.ptr(y), _lib.ptr(stats), M, C, 1e-6, st)
    assert rc < 0 and b"aligned" in lib.acr_last_error()
    ws = torch.empty(lib.acr_layernorm_ws_floats(M, C), device=dev)
    dg, db, dx = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(M * C, device=dev)
    rc = lib.acr_layernorm_bwd_f32(_lib.ptr(xo), _lib.ptr(x), _lib.ptr(gam), _lib.ptr(stats), None, _lib.ptr(dx), _lib.ptr(ws),
                                   _lib.ptr(dg), _lib.ptr(db), M, C, st)
    assert rc < 0 and b"aligned" in lib.acr_last_error()
    # the resident-score attention is an fp32 entry point: other dtypes are refused, not reinterpreted
    d = ops._desc(1, 1, 5, torch.bfloat16)
    q = torch.zeros(5 * 192, device=dev)
    rc = lib.acr_attn_fwd_scores(d, _lib.ptr(q), _lib.ptr(q), _lib.ptr(q), _lib.ptr(q), _lib.ptr(q), _lib.ptr(q), None, 0, 0, st)
    assert rc == -3 and b"fp32" in lib.acr_last_error()
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K,bias,resid", [(1, 128, 64, True, False), (130, 200, 128, True, True),
                                              (785, 2304, 768, True, False), (2 * 785, 768, 768, False, True),
                                              (333, 768, 2304, False, False),
                                              # large problems take the eight-wave 320x256 kernel (>= 200 tiles):
                                              # ragged in M and N, with and without bias / residual; K = 64 is one stage
                                              (320 * 100 + 17, 520, 64, True, True), (32 * 785, 768, 768, True, True),
                                              (320 * 70 + 300, 1000, 192, False, False)])
def test_linear_bf16(M, N, K, bias, resid):
    """acr_linear_bf16 (hand-written MFMA GEMM) forward + autograd against fp64; ragged M/N tails included."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(dev).bfloat16().requires_grad_(True)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16().requires_grad_(True)
    b = torch.randn(N, generator=g).to(dev).bfloat16().requires_grad_(True) if bias else None
    r = torch.randn(M, N, generator=g).to(dev).bfloat16().requires_grad_(True) if resid else None
    y = ops.LinearBf16Fn.apply(x, w, b, r)
    dy = torch.randn(M, N, generator=g).to(dev).bfloat16()
    (y.float() * dy.float()).sum().backward()
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    ref = xd @ wd.t()
    if bias:
        ref = ref + b.detach().double()
    if resid:
        ref = ref + r.detach().double()
    (ref * dy.double()).sum().backward()
    assert (y.double() - ref).abs().max() <= 1e-2 * ref.abs().max()
    assert (x.grad.double() - xd.grad).abs().max() <= 1.5e-2 * xd.grad.abs().max()
    assert (w.grad.double() - wd.grad).abs().max() <= 1.5e-2 * wd.grad.abs().max()
    if bias:
        assert (b.grad.double() - dy.double().sum(0)).abs().max() <= 1e-2 * dy.double().sum(0).abs().max() + 1e-2
    if resid:
        torch.testing.assert_close(r.grad, dy)


@pytest.mark.parametrize("N,C,H,W", [(2, 64, 8, 8), (3, 256, 16, 24), (2, 1024, 28, 28), (1, 64, 224, 224)])
@pytest.mark.parametrize("act", ["none", "relu", "add_relu"])
def test_groupnorm_fused(N, C, H, W, act):
    """Fused bf16 GroupNorm(32) [+ residual] [+ ReLU] forward/backward vs torch group_norm + relu in fp64."""
    from acr_wsss_amd import ops
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(C + H)
    x = (torch.randn(N, C, H, W, generator=g) * 1.7 + 0.3).to(dev).bfloat16().requires_grad_(True)
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(dev).bfloat16().requires_grad_(True)
    b = (0.3 * torch.randn(C, generator=g)).to(dev).bfloat16().requires_grad_(True)
    r = torch.randn(N, C, H, W, generator=g).to(dev).bfloat16().requires_grad_(True) if act == "add_relu" else None
    assert ops.groupnorm_fusable(x, r)
    y = ops.groupnorm_act(x, w, b, act, r)
    dy = torch.randn(N, C, H, W, generator=g).to(dev).bfloat16()
    (y.float() * dy.float()).sum().backward()
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    ref = F.group_norm(xd, 32, wd, bd, 1e-5)
    if r is not None:
        rd = r.detach().double().requires_grad_(True)
        ref = ref + rd
    if act != "none":
        # take the ReLU mask from the kernel's own (bf16-rounded) output so elements that round to +-0 agree
        ref = ref * (y.detach() > 0).double()
    (ref * dy.double()).sum().backward()
    assert (y.double() - ref).abs().max() <= 1.2e-2 * ref.abs().max()
    assert (x.grad.double() - xd.grad).abs().max() <= 2e-2 * xd.grad.abs().max()
    assert (w.grad.double() - wd.grad).abs().max() <= 2e-2 * wd.grad.abs().max() + 1e-2
    assert (b.grad.double() - bd.grad).abs().max() <= 2e-2 * bd.grad.abs().max() + 1e-2
    if r is not None:
        assert (r.grad.double() - rd.grad).abs().max() <= 1e-2 * rd.grad.abs().max()


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (100, 256, 128), (785 * 2 + 3, 768, 2304), (25120, 2304, 768),
                                   # 256x256 eight-wave kernel + ragged tail rows through the 128x128 kernel (M % 32 != 0)
                                   (16 * 785, 768, 768), (16 * 1025, 256, 512), (4096 + 31, 512, 256)])
def test_wgrad_bf16(M, N, K):
    """Split-M TN weight-gradient GEMM (acr_wgrad_bf16 / acr_wgrad_bias_bf16) vs fp64, including ragged token counts; the
    fused bias gradient (column sums of dy from the same sweep) against the fp64 column sums."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + N)
    dy = torch.randn(M, N, generator=g).to(dev).bfloat16()
    x = torch.randn(M, K, generator=g).to(dev).bfloat16()
    dw = ops.wgrad_bf16(dy, x)
    ref = dy.double().t() @ x.double()
    assert dw.shape == (N, K)
    assert (dw.double() - ref).abs().max() <= 1e-2 * ref.abs().max()
    # deterministic: same bits on a second run
    assert torch.equal(dw, ops.wgrad_bf16(dy, x))
    dw2, db = ops.wgrad_bias_bf16(dy, x)
    assert (dw2.double() - ref).abs().max() <= 1e-2 * ref.abs().max()
    cs = dy.double().sum(0)
    assert db.shape == (N,) and (db.double() - cs).abs().max() <= 1e-2 * cs.abs().max() + 1e-2
    dw3, db3 = ops.wgrad_bias_bf16(dy, x)
    assert torch.equal(dw2, dw3) and torch.equal(db, db3)


def test_weight_std_all_fused():
    """One-launch weight standardisation of several conv weights vs the torch expression (std_conv.py:56-59)."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(4)
    shapes = [(64, 3, 7, 7), (64, 64, 1, 1), (256, 64, 3, 3), (33, 5, 1, 1)]
    ws = [(torch.randn(s, generator=g) * 0.3 + 0.05).to(dev).bfloat16().requires_grad_(True) for s in shapes]
    outs = ops.weight_std_all(ws)
    # 1x1 convolutions also get their standardised weight transposed, (cin, cout), from the same launch
    tr = ops.WeightStdAllFn.last_transposed
    assert [t is not None for t in tr] == [False, True, False, True]
    assert torch.equal(tr[1], outs[1].reshape(64, 64).t()) and torch.equal(tr[3], outs[3].reshape(33, 5).t())
    gs = [torch.randn(s, generator=g).to(dev).bfloat16() for s in shapes]
    sum((o.float() * gi.float()).sum() for o, gi in zip(outs, gs)).backward()
    for w, o, gi in zip(ws, outs, gs):
        wd = w.detach().double().requires_grad_(True)
        std, mean = torch.std_mean(wd, dim=[1, 2, 3], keepdim=True, unbiased=False)
        ref = (wd - mean) / (std + 1e-5)
        (ref * gi.double()).sum().backward()
        assert (o.double() - ref).abs().max() <= 1e-2 * ref.abs().max()
        assert (w.grad.double() - wd.grad).abs().max() <= 2e-2 * wd.grad.abs().max() + 1e-3


@pytest.mark.parametrize("M,C", [(1, 256), (37, 768), (25120, 768), (130, 1024)])
def test_layernorm_bf16(M, C):
    """HIP LayerNorm forward/backward (one-pass backward with dgamma/dbeta) vs torch layer_norm in fp64."""
    from acr_wsss_amd import ops
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 2 + 0.5).to(dev).bfloat16().requires_grad_(True)
    ln = torch.nn.LayerNorm(C, eps=1e-6).to(dev).bfloat16()
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * torch.randn(C, generator=g))
        ln.bias.copy_(0.3 * torch.randn(C, generator=g))
    y = ops.layer_norm(x, ln)
    assert y.grad_fn is not None and "LayerNormFn" in type(y.grad_fn).__name__
    dy = torch.randn(M, C, generator=g).to(dev).bfloat16()
    (y.float() * dy.float()).sum().backward()
    xd = x.detach().double().requires_grad_(True)
    wd, bd = ln.weight.detach().double().requires_grad_(True), ln.bias.detach().double().requires_grad_(True)
    ref = F.layer_norm(xd, (C,), wd, bd, 1e-6)
    (ref * dy.double()).sum().backward()
    assert (y.double() - ref).abs().max() <= 1.2e-2 * ref.abs().max()
    assert (x.grad.double() - xd.grad).abs().max() <= 2e-2 * xd.grad.abs().max()
    assert (ln.weight.grad.double() - wd.grad).abs().max() <= 1e-2 * wd.grad.abs().max() + 1e-2
    assert (ln.bias.grad.double() - bd.grad).abs().max() <= 1e-2 * bd.grad.abs().max() + 1e-2


@pytest.mark.parametrize("N,cin,cout,H,W", [(2, 64, 64, 8, 8), (3, 64, 256, 16, 24), (2, 256, 64, 28, 28),
                                             (2, 1024, 256, 28, 28), (2, 128, 512, 56, 56), (1, 64, 256, 112, 112)])
def test_conv1x1_bf16(N, cin, cout, H, W):
    """NCHW 1x1 convolution on the GEMM kernels: forward, input gradient, weight gradient vs fp64 conv2d; covers pixel
    tiles that run past H*W (28x28 = 784 = 6.125 tiles) and the ragged 64-pixel chunk of the weight gradient."""
    from acr_wsss_amd import ops
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(cin + cout + H)
    x = torch.randn(N, cin, H, W, generator=g).to(dev).bfloat16().requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 1, generator=g) * cin ** -0.5).to(dev).bfloat16().requires_grad_(True)
    assert ops.conv1x1_fusable(x, w, 1)
    y = ops.conv1x1(x, w)
    dy = torch.randn(N, cout, H, W, generator=g).to(dev).bfloat16()
    (y.float() * dy.float()).sum().backward()
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    ref = F.conv2d(xd, wd)
    (ref * dy.double()).sum().backward()
    assert (y.double() - ref).abs().max() <= 1e-2 * ref.abs().max()
    assert (x.grad.double() - xd.grad).abs().max() <= 1.5e-2 * xd.grad.abs().max()
    assert (w.grad.double() - wd.grad).abs().max() <= 1e-2 * wd.grad.abs().max()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(2, 3, 8, 8), (2, 5, 9, 11), (1, 64, 224, 224), (2, 4, 7, 16), (3, 2, 16, 24)])
def test_maxpool_same_bf16(shape, dtype):
    """SAME-padded 3x3/2 max-pool (forward value, argmax routing of the gradient) vs F.pad(-inf)+max_pool2d, bf16 and fp32
    maps; values are made distinct so the argmax is unique and the comparison is exact."""
    from acr_wsss_amd import ops
    from acr_wsss_amd.backbone import _same_pad, pad_same
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(sum(shape))
    n = math.prod(shape)
    x = (torch.randperm(n, generator=g).float().reshape(shape) % 251 - 125.0) / 4.0      # bf16-exact, few ties
    x = x.to(dev).to(dtype).requires_grad_(True)
    ph, pw = _same_pad(shape[2], 3, 2), _same_pad(shape[3], 3, 2)
    y = ops.maxpool3x3s2_same(x, ph // 2, pw // 2, ph, pw)
    xr = x.detach().clone().requires_grad_(True)
    yr = F.max_pool2d(pad_same(xr, 3, 2, value=-float("inf")), 3, 2)
    assert y.shape == yr.shape and torch.equal(y, yr)
    dy = torch.randn(y.shape, generator=g).to(dev).to(dtype)
    y.backward(dy)
    yr.backward(dy)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-6          # a pixel collects up to 4 window gradients: summation order only
    assert torch.allclose(x.grad.float(), xr.grad.float(), atol=tol, rtol=tol)


@pytest.mark.parametrize("M,D,Hd", [(197 * 2, 192, 768), (785, 768, 3072), (320 * 3 + 17, 128, 256)])
def test_fused_mlp_bf16(M, D, Hd):
    """MlpFn (GELU and GELU' inside the GEMM epilogues) against an fp64 fc2(gelu(fc1(x))) + resid: output and all six
    gradients; tile edges in both dimensions (M not a multiple of 320, hidden/out not multiples of 256)."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + D)
    fc1 = torch.nn.Linear(D, Hd).to(dev).bfloat16()
    fc2 = torch.nn.Linear(Hd, D).to(dev).bfloat16()
    x = (torch.randn(2, M // 2 if M % 2 == 0 else M, D, generator=g)[:1 if M % 2 else 2]).to(dev).bfloat16()
    x = x.reshape(1, -1, D).contiguous().requires_grad_(True)
    r = torch.randn(x.shape, generator=g).to(dev).bfloat16().requires_grad_(True)
    dy = torch.randn(x.shape, generator=g).to(dev).bfloat16()
    assert ops.mlp_fusable(x, fc1, fc2)
    y = ops.mlp(x, fc1, fc2, r)
    (y.float() * dy.float()).sum().backward()
    got = [y, x.grad, r.grad, fc1.weight.grad, fc1.bias.grad, fc2.weight.grad, fc2.bias.grad]
    xd, rd = x.detach().double().requires_grad_(True), r.detach().double().requires_grad_(True)
    p = [t.detach().double().requires_grad_(True) for t in (fc1.weight, fc1.bias, fc2.weight, fc2.bias)]
    ref = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(xd, p[0], p[1])), p[2], p[3]) + rd
    (ref * dy.double()).sum().backward()
    want = [ref, xd.grad, rd.grad, p[0].grad, p[1].grad, p[2].grad, p[3].grad]
    names = ["y", "dx", "dresid", "dW1", "db1", "dW2", "db2"]
    for n, a, b in zip(names, got, want):
        err = (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-9)
        assert err <= 2.5e-2, (n, err)


def test_weight_transposes_cache():
    """WeightTransposes: one launch transposes every registered Linear weight (ragged 64x64 tiles included); the cached
    copy is served only while the weight's version is the recorded one -- an in-place update makes weight_t fall back."""
    from acr_wsss_amd import ops
    dev = _dev()
    torch.manual_seed(3)
    lins = [torch.nn.Linear(i, o).to(dev).bfloat16() for i, o in ((768, 2304), (3072, 768), (72, 200), (64, 64))]
    cache = ops.WeightTransposes(lins)
    cache.refresh()
    for m in lins:
        wt = ops.weight_t(m.weight, m)
        assert wt is m._acr_wt and torch.equal(wt, m.weight.t().contiguous())
    with torch.no_grad():
        lins[0].weight.mul_(2.0)                              # in-place update bumps the version: the copy is stale
    wt0 = ops.weight_t(lins[0].weight, lins[0])
    assert wt0 is not lins[0]._acr_wt and torch.equal(wt0, lins[0].weight.t().contiguous())
    cache.refresh()
    assert ops.weight_t(lins[0].weight, lins[0]) is lins[0]._acr_wt
    assert torch.equal(lins[0]._acr_wt, lins[0].weight.t().contiguous())
    # a re-allocated storage (`p.data = ...`, module.to(other device)) is seen through the recorded address ...
    lins[1].weight.data = lins[1].weight.data.clone() * 3.0
    wt1 = ops.weight_t(lins[1].weight, lins[1])
    assert wt1 is not lins[1]._acr_wt and torch.equal(wt1, lins[1].weight.t().contiguous())
    # ... a write THROUGH .data is not (no version bump, same storage): the documented contract is an explicit refresh
    lins[2].weight.data.mul_(2.0)
    assert ops.weight_t(lins[2].weight, lins[2]) is lins[2]._acr_wt and not torch.equal(lins[2]._acr_wt, lins[2].weight.t())
    cache2 = ops.WeightTransposes(lins)
    cache2.refresh()
    for m in lins:
        assert ops.weight_t(m.weight, m) is m._acr_wt and torch.equal(m._acr_wt, m.weight.t().contiguous())


@pytest.mark.parametrize("M,N,K", [(1, 32, 32), (130, 200, 72), (785, 2304, 768), (2 * 785 + 3, 768, 3072), (333, 576, 192),
                                   (25120, 768, 768), (1025, 3072, 768), (1154, 768, 3072), (290, 768, 768),
                                   (256, 256, 2048), (1280, 2176, 256)])     # 4 tiles split 16 ways; 170 tiles (the limit) 3 ways
@pytest.mark.parametrize("split", [0, 1])
def test_gemm_f32_linear(M, N, K, split):
    """acr_gemm_f32 (exact-fp32 MFMA, reference precision) through LinearF32Fn: forward NT with bias + residual, input
    gradient NN on the weight as stored, weight + bias gradient in one TN sweep -- against fp64; ragged M (not a multiple
    of 128), N with a partial tile (200, 576), K tails (72 = 2.25 chunks) and a token count that is not a multiple of
    the 32-deep chunk in the TN contraction (785, 1573, 1025); products of a few dozen tiles (CAM generation at batch 2:
    1154 = 2 x 577 and 290 = 2 x 145 tokens), which run entirely as K-split parts + the tail epilogue.
    split = 1: the same products as six bf16-MFMA terms of a three-way operand split (math = ACR_MATH_BF16X3, a per-call
    argument) -- held to the SAME 1e-5 against fp64 as the exact-fp32 MFMA, and bit-reproducible."""
    from acr_wsss_amd import ops
    _gemm_f32_linear_case(ops, _dev(), M, N, K, split)


def _gemm_f32_linear_case(ops, dev, M, N, K, math=0):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).requires_grad_(True)
    b = torch.randn(N, generator=g).to(dev).requires_grad_(True)
    r = torch.randn(M, N, generator=g).to(dev).requires_grad_(True)
    assert ops.linear_f32_usable(x, w)
    y = ops.LinearF32Fn.apply(x, w, b, r, None, math)
    dy = torch.randn(M, N, generator=g).to(dev)
    (y * dy).sum().backward()
    xd, wd, bd, rd = (t.detach().double().requires_grad_(True) for t in (x, w, b, r))
    ref = xd @ wd.t() + bd + rd
    (ref * dy.double()).sum().backward()
    for name, got, want in (("y", y, ref), ("dx", x.grad, xd.grad), ("dw", w.grad, wd.grad), ("db", b.grad, bd.grad)):
        err = (got.double() - want).abs().max() / want.abs().max()
        assert err <= 1e-5, (name, float(err))
    torch.testing.assert_close(r.grad, dy)
    # deterministic (split-token slabs are summed in a fixed order)
    w.grad = None
    x.grad = None
    y2 = ops.LinearF32Fn.apply(x, w, b, r, None, math)
    (y2 * dy).sum().backward()
    assert torch.equal(y2, y)
    dw1 = w.grad.clone()
    w.grad = None
    (ops.LinearF32Fn.apply(x, w, b, r, None, math) * dy).sum().backward()
    assert torch.equal(dw1, w.grad)


def _gemm_pair(ops, mode, a, b, shape):
    """(exact-fp32 MFMA result, split-product result, fp64 reference, condition scale sum_k |a||b|) of one acr_gemm_f32 call."""
    outs = []
    for math in (0, 1):
        c = torch.empty(shape, dtype=torch.float32, device=a.device)
        ops.gemm_f32_raw(mode, a, b, c, math=math)
        outs.append(c.double())
    ad, bd = a.double(), b.double()
    if mode == "nt":
        ref, scale = ad @ bd.t(), ad.abs() @ bd.abs().t()
    elif mode == "nn":
        ref, scale = ad @ bd, ad.abs() @ bd.abs()
    else:
        ref, scale = ad.t() @ bd, ad.abs().t() @ bd.abs()
    return outs[0], outs[1], ref, scale


@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
@pytest.mark.parametrize("case", ["range", "cancel", "underflow"])
def test_split_math_adversarial_operands(mode, case):
    """the split-product arithmetic (math = ACR_MATH_BF16X3: six bf16-MFMA terms of a three-way operand
    split) on operands chosen to break a narrower format, held to <= 2x the exact-fp32 MFMA kernel's OWN error against fp64
    on the same inputs (errors normalised by sum_k |a||b|, the scale fp32 rounding errors live on):
      range      every contraction row spans 2^-40 .. 2^+40 (bf16 alone would keep 8 bits of the large terms and nothing of
                 the small ones; the three pieces carry 24 bits of EACH element, whatever its exponent);
      cancel     dot products that cancel to ~1e-6 of their terms (the second half of the contraction repeats the first with
                 the opposite sign and a 1e-6 relative perturbation): the result is made of the low-order bits;
      underflow  operands of magnitude 2^-112: their second / third pieces fall below bf16's normal range.  Measured and
                 documented behaviour: the lost pieces are an ABSOLUTE error of at most 2^-126 per product factor, i.e.
                 <= 2^-126 * (sum_k |a| + sum_k |b|) per output on top of the 2x bound -- operands above ~2^-100 (every
                 activation and gradient of this network by > 60 binades) are unaffected."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(11)
    M, N, K = 384, 256, 768
    am, bm = ((M, K), (N, K)) if mode == "nt" else (((M, K), (K, N)) if mode == "nn" else ((K, M), (K, N)))
    kdim_a = 1 if mode in ("nt", "nn") else 0
    kdim_b = 1 if mode == "nt" else 0
    a, b = torch.randn(am, generator=g), torch.randn(bm, generator=g)
    if case == "range":
        a = a * torch.exp2(torch.randint(-40, 41, am, generator=g).float())
        b = b * torch.exp2(torch.randint(-20, 21, bm, generator=g).float())
    elif case == "cancel":
        h = K // 2
        ia, ib = [slice(None)] * 2, [slice(None)] * 2
        ia2, ib2 = list(ia), list(ib)
        ia[kdim_a], ia2[kdim_a], ib[kdim_b], ib2[kdim_b] = slice(0, h), slice(h, K), slice(0, h), slice(h, K)
        a[tuple(ia2)] = a[tuple(ia)]
        b[tuple(ib2)] = -b[tuple(ib)] * (1 + 1e-6 * torch.randn(b[tuple(ib)].shape, generator=g))
    else:
        a = a * 2.0 ** -112
    a, b = a.to(dev).contiguous(), b.to(dev).contiguous()
    exact, split, ref, scale = _gemm_pair(ops, mode, a, b, (M, N))
    assert torch.isfinite(exact).all() and torch.isfinite(split).all()
    if case == "cancel":                                     # the instance really cancels: |result| ~ 1e-6 .. 1e-5 of its terms
        assert float((ref.abs() / scale).median()) < 2e-5
    e_exact, e_split = (exact - ref).abs() / scale, (split - ref).abs() / scale
    floor = 2.0 ** -126 * (a.double().abs().sum(kdim_a).reshape(-1, 1) + b.double().abs().sum(kdim_b).reshape(1, -1)) / scale \
        if case == "underflow" else 0.0
    print("%s/%s: max normalised error exact %.3e split %.3e; rms exact %.3e split %.3e" % (
        mode, case, float(e_exact.max()), float(e_split.max()), float(e_exact.pow(2).mean().sqrt()), float(e_split.pow(2).mean().sqrt())))
    assert float(e_split.max()) <= 2 * float(e_exact.max()) + float(torch.as_tensor(floor).max())
    assert float(e_split.pow(2).mean().sqrt()) <= 2 * float(e_exact.pow(2).mean().sqrt()) + float(torch.as_tensor(floor).max())
    assert float(e_exact.max()) < 1e-6                       # sanity: the yardstick itself is an fp32-accurate kernel


def test_split_math_propagates_non_finite_values():
    """inf / NaN operands under the split-product arithmetic: a non-finite input never yields a finite output where the exact
    arithmetic has none (a NaN stays a NaN; an inf becomes a NaN because inf - bf16(inf) is one: documented in
    include/acr_hip.h's acr_math), and rows / columns without one are untouched."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(5)
    M, N, K = 256, 128, 256
    a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    a[3, 17], a[40, 200], a[77, 5] = float("inf"), float("nan"), float("-inf")
    b[9, 100] = float("nan")
    a, b = a.to(dev), b.to(dev)
    exact, split, ref, _ = _gemm_pair(ops, "nt", a, b, (M, N))
    bad = ~torch.isfinite(ref)
    assert bad[3].all() and bad[40].all() and bad[77].all() and bad[:, 9].all() and int(bad.sum()) == 3 * N + M - 3
    assert (~torch.isfinite(split))[bad].all() and (~torch.isfinite(exact))[bad].all()
    assert torch.isnan(split[40]).all() and torch.isnan(split[:, 9]).all()
    ok = ~bad
    assert torch.isfinite(split[ok]).all()
    assert float((split[ok] - ref[ok]).abs().max()) <= 2 * float((exact[ok] - ref[ok]).abs().max()) + 1e-6


def test_split3_planes_sum_to_the_operand_exactly():
    """acr_split3_bf16 (the operand form of the split-product arithmetic): x = p0 + p1 + p2 EXACTLY for every fp32 value whose
    pieces stay in bf16's normal range -- 3 x 8 significand bits cover the 24 of an fp32 -- incl. values spanning 2^+-60."""
    from acr_wsss_amd import _lib as L
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(3)
    rows, cols, ld = 300, 192, 200
    buf = (torch.randn(rows, ld, generator=g) * torch.exp2(torch.randint(-60, 61, (rows, ld), generator=g).float())).to(dev)
    x = buf[:, :cols]
    planes = torch.empty((3, rows, cols), dtype=torch.bfloat16, device=dev)
    L.check(L.load().acr_split3_bf16(L.ptr(x), rows, cols, ld, L.ptr(planes), rows * cols, L.stream_ptr()), "acr_split3_bf16")
    p = planes.double()
    assert torch.equal(p[0] + p[1] + p[2], x.double())
    assert torch.equal(planes[0], x.to(torch.bfloat16))
    assert float((p[1].abs() / x.double().abs().clamp_min(1e-300)).max()) <= 2.0 ** -8
    assert float((p[2].abs() / x.double().abs().clamp_min(1e-300)).max()) <= 2.0 ** -16


def test_attention_split_math_sharp_softmax_is_as_accurate_as_exact():
    """The split-product attention on an adversarial instance for a narrower format: logits of +-200 (q, k scaled 6x: one-hot
    softmax rows beside flat ones), v with a 2^+-20 spread inside a row.  Output, head mean and dqkv errors against fp64 are
    held to <= 2x those of the exact-fp32 kernels on the same inputs (+ a floor of 1e-6 of the tensor's max)."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(2)
    B, T, H = 1, 197, 2
    qkv = torch.randn(B, T, 3, H, 64, generator=g)
    qkv[:, :, :2] *= 6.0
    qkv[:, :, 2] *= torch.exp2(torch.randint(-20, 21, (B, T, H, 64), generator=g).float())
    qkv = qkv.reshape(B, T, 3 * H * 64).to(dev)
    d_o = torch.randn(B, T, H * 64, generator=g).to(dev)
    gpm = torch.randn(B, T, T, generator=g).to(dev)
    res = []
    for math in (0, 1):
        q = qkv.clone().requires_grad_(True)
        stack = ops.MeanStack(B, 1, T, dev)
        o, pm = ops.attention_core(q, H, stack, 0, None, math)
        ((o * d_o).sum() + (pm * gpm).sum()).backward()
        res.append((o.detach().double(), pm.detach().double(), q.grad.double()))
    qd = qkv.double().requires_grad_(True)
    o_ref, P = _ref_attn(qd, H)
    pm_ref = P.mean(1)
    ((o_ref * d_o.double()).sum() + (pm_ref * gpm.double()).sum()).backward()
    for name, i, want in (("o", 0, o_ref.detach()), ("pmean", 1, pm_ref.detach()), ("dqkv", 2, qd.grad)):
        e0 = float((res[0][i] - want).abs().max() / want.abs().max())
        e1 = float((res[1][i] - want).abs().max() / want.abs().max())
        print("%s: exact %.3e split %.3e of max" % (name, e0, e1))
        assert e1 <= 2 * e0 + 1e-6, (name, e0, e1)


@pytest.mark.parametrize("math", [0, 1])
@pytest.mark.parametrize("M,D,Hd", [(197 * 2, 192, 768), (785, 768, 3072), (131, 128, 260), (290, 768, 3072)])
def test_fused_mlp_f32(M, D, Hd, math):
    """MlpF32Fn (GELU / GELU' inside the fp32 GEMM epilogues) against fp64 fc2(gelu(fc1(x))) + resid and against the
    stock torch fp32 ops it replaces (same exact-erf GELU): output and all six gradients.  math = 1: split products on the bf16
    MFMA, same tolerance."""
    from acr_wsss_amd import ops
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + D)
    fc1, fc2 = torch.nn.Linear(D, Hd).to(dev), torch.nn.Linear(Hd, D).to(dev)
    with torch.no_grad():
        fc1.bias.copy_(torch.randn(Hd, generator=g) * 0.3)
        fc2.bias.copy_(torch.randn(D, generator=g) * 0.3)
        fc1.weight.mul_(3.0)                                    # pre-activations of O(1): both GELU branches exercised
    x = torch.randn(1, M, D, generator=g).to(dev).requires_grad_(True)
    r = torch.randn(1, M, D, generator=g).to(dev).requires_grad_(True)
    dy = torch.randn(1, M, D, generator=g).to(dev)
    assert ops.mlp_f32_usable(x, fc1, fc2)
    y = ops.mlp_f32(x, fc1, fc2, r, math)
    (y * dy).sum().backward()
    got = [y, x.grad, r.grad, fc1.weight.grad, fc1.bias.grad, fc2.weight.grad, fc2.bias.grad]
    xd, rd = x.detach().double().requires_grad_(True), r.detach().double().requires_grad_(True)
    p = [t.detach().double().requires_grad_(True) for t in (fc1.weight, fc1.bias, fc2.weight, fc2.bias)]
    ref = F.linear(F.gelu(F.linear(xd, p[0], p[1])), p[2], p[3]) + rd
    (ref * dy.double()).sum().backward()
    want = [ref, xd.grad, rd.grad, p[0].grad, p[1].grad, p[2].grad, p[3].grad]
    for n, a, b in zip(["y", "dx", "dresid", "dW1", "db1", "dW2", "db2"], got, want):
        err = (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-9)
        assert err <= 2e-5, (n, err)


@pytest.mark.parametrize("M,C", [(1, 256), (37, 768), (25120, 768), (130, 1024)])
def test_layernorm_f32(M, C):
    """fp32 rows through the HIP LayerNorm (reference precision): forward, dx with the fused skip gradient, dgamma, dbeta
    against fp64."""
    from acr_wsss_amd import ops
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 2 + 0.5).to(dev).requires_grad_(True)
    ln = torch.nn.LayerNorm(C, eps=1e-6).to(dev)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * torch.randn(C, generator=g))
        ln.bias.copy_(0.3 * torch.randn(C, generator=g))
    assert ops.layer_norm_fusable(x, ln)
    y, skip = ops.layer_norm_skip(x, ln)
    dy = torch.randn(M, C, generator=g).to(dev)
    ds = torch.randn(M, C, generator=g).to(dev)
    ((y * dy).sum() + (skip * ds).sum()).backward()
    xd = x.detach().double().requires_grad_(True)
    wd, bd = ln.weight.detach().double().requires_grad_(True), ln.bias.detach().double().requires_grad_(True)
    ref = F.layer_norm(xd, (C,), wd, bd, 1e-6)
    ((ref * dy.double()).sum() + (xd * ds.double()).sum()).backward()
    for n, a, b in (("y", y, ref), ("dx", x.grad, xd.grad), ("dgamma", ln.weight.grad, wd.grad), ("dbeta", ln.bias.grad, bd.grad)):
        err = (a.double() - b).abs().max() / b.abs().max()
        assert err <= 2e-5, (n, float(err))


@pytest.mark.parametrize("M,C,N", [(2 * 785, 768, 2304), (300, 256, 576), (130, 1024, 256), (64, 512, 96)])
def test_layernorm_image_f32(M, C, N):
    """LayerNorm whose output leaves as its consumer's split-product image (acr_layernorm_image_f32 + LinearF32Fn / MlpF32Fn with
    x_image; what a block runs under f32_split): norm -> Linear forward, input gradient through the LayerNorm backward (with the
    fused skip gradient), weight / bias gradients of both, against fp64 and against the two-kernel path (LayerNorm in fp32, then the
    image pass).  Row counts that are not multiples of 32 / 128: the image's padding rows must be ZERO (the weight gradient contracts
    over them) -- the allocator is primed with NaNs so that unwritten rows would show."""
    from acr_wsss_amd import ops
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 2 + 0.5).to(dev).requires_grad_(True)
    ln = torch.nn.LayerNorm(C, eps=1e-6).to(dev)
    lin = torch.nn.Linear(C, N).to(dev)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * torch.randn(C, generator=g))
        ln.bias.copy_(0.3 * torch.randn(C, generator=g))
    assert ops.ln_image_usable(x, ln, lin, 1) and not ops.ln_image_usable(x, ln, lin, 0)
    dy = torch.randn(M, N, generator=g).to(dev)
    ds = torch.randn(M, C, generator=g).to(dev)
    prime = torch.full((int(ops.L.load().acr_x3_image_floats(M, C)),), float("nan"), device=dev)
    del prime
    h, skip, hi = ops.layer_norm_image(x, ln)
    assert h.shape == x.shape and h.untyped_storage().nbytes() == 4
    y = ops.linear_or_hip(h, lin, None, True, math=1, x_image=hi)
    ((y * dy).sum() + (skip * ds).sum()).backward()
    got = [y.detach(), x.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()]
    assert all(torch.isfinite(t).all() for t in got)
    xd = x.detach().double().requires_grad_(True)
    ps = [p.detach().double().requires_grad_(True) for p in (ln.weight, ln.bias, lin.weight, lin.bias)]
    ref = F.linear(F.layer_norm(xd, (C,), ps[0], ps[1], 1e-6), ps[2], ps[3])
    ((ref * dy.double()).sum() + (xd * ds.double()).sum()).backward()
    want = [ref, xd.grad] + [p.grad for p in ps]
    names = ("y", "dx", "dgamma", "dbeta", "dW", "db")
    for n, a, b in zip(names, got, want):
        err = (a.double() - b).abs().max() / b.abs().max()
        assert err <= 2e-5, (n, float(err))
    # the two-kernel path: same numbers up to the rounding of the row statistics (another summation order)
    for p in (x, ln.weight, ln.bias, lin.weight, lin.bias):
        p.grad = None
    h2, skip2 = ops.layer_norm_skip(x, ln)
    y2 = ops.linear_or_hip(h2, lin, None, True, math=1)
    ((y2 * dy).sum() + (skip2 * ds).sum()).backward()
    two = [y2.detach(), x.grad, ln.weight.grad, ln.bias.grad, lin.weight.grad, lin.bias.grad]
    for n, a, b in zip(names, got, two):
        err = (a - b).abs().max() / b.abs().max()
        assert err <= 2e-6, (n, float(err))
    # and the MLP form (norm2 -> fc1 -> GELU -> fc2 + skip)
    fc2 = torch.nn.Linear(N, C).to(dev)
    for p in (x, ln.weight, ln.bias, lin.weight, lin.bias):
        p.grad = None
    h, skip, hi = ops.layer_norm_image(x, ln)
    z = ops.mlp_f32(h, lin, fc2, skip, 1, hi)
    (z * ds).sum().backward()
    ps = [p.detach().double().requires_grad_(True) for p in (x, ln.weight, ln.bias, lin.weight, lin.bias, fc2.weight, fc2.bias)]
    zr = ps[0] + F.linear(F.gelu(F.linear(F.layer_norm(ps[0], (C,), ps[1], ps[2], 1e-6), ps[3], ps[4])), ps[5], ps[6])
    (zr * ds.double()).sum().backward()
    for n, a, b in [("z", z, zr)] + [("g%d" % i, p.grad, q.grad) for i, (p, q) in enumerate(zip((x, ln.weight, ln.bias, lin.weight, lin.bias, fc2.weight, fc2.bias), ps))]:
        err = (a.double() - b).abs().max() / b.abs().max()
        assert err <= 3e-5, (n, float(err))


@pytest.mark.parametrize("B,D,h,w,P", [(4, 768, 28, 28, 1), (3, 192, 14, 14, 2), (2, 100, 5, 7, 1), (32, 768, 28, 28, 1)])
def test_tokens_assembly(B, D, h, w, P):
    """csrc/tokens.hip vs the reference's chain (vision_transformer.py:449-467): projection bias add, flatten(2).transpose(1, 2), cat with
    the class (+ distillation) token, + pos_embed -- forward bit for bit (same order of the two additions), the projection's gradient
    bit for bit (a transposed copy), position / bias / prefix gradients to summation-order tolerance.  Tiles that run past T and D."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(B + D)
    y = torch.randn(B, D, h, w, generator=g).to(dev).requires_grad_(True)
    bias = torch.randn(D, generator=g).to(dev).requires_grad_(True)
    prefix = torch.randn(P, D, generator=g).to(dev).requires_grad_(True)
    pos = torch.randn(1, P + h * w, D, generator=g).to(dev).requires_grad_(True)
    assert ops.tokens_fusable(y, bias, prefix, pos)
    tok = ops.tokens(y, bias, prefix, pos)
    dtok = torch.randn(tok.shape, generator=g).to(dev)
    (tok * dtok).sum().backward()
    got = [tok.detach(), y.grad, bias.grad, prefix.grad, pos.grad]
    y2, b2, p2, pos2 = (t.detach().clone().requires_grad_(True) for t in (y, bias, prefix, pos))
    ref = torch.cat([p2.unsqueeze(0).expand(B, -1, -1), (y2 + b2.view(1, -1, 1, 1)).flatten(2).transpose(1, 2)], dim=1) + pos2
    (ref * dtok).sum().backward()
    assert torch.equal(got[0], ref) and torch.equal(got[1], y2.grad)
    for n, a, b in (("dbias", got[2], b2.grad), ("dprefix", got[3], p2.grad), ("dpos", got[4], pos2.grad)):
        assert (a - b).abs().max() <= 2e-6 * b.abs().max() * B ** 0.5, n


@pytest.mark.parametrize("shape", [(2, 8, 16, 24), (3, 5, 9, 13), (1, 4, 7, 16), (2, 256, 112, 112)])
def test_subsample2(shape):
    """ops.subsample2 (the input of a stride-2 1x1 convolution) and its backward are x[:, :, ::2, ::2] and its autograd backward,
    bit for bit -- vector path (W % 8 == 0) and the guarded scalar one (odd sizes)."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(dev).requires_grad_(True)
    y = ops.subsample2(x)
    dy = torch.randn(y.shape, generator=g).to(dev)
    (y * dy).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    ref = x2[:, :, ::2, ::2].contiguous()
    (ref * dy).sum().backward()
    assert torch.equal(y, ref) and torch.equal(x.grad, x2.grad)


def test_weight_std_all_f32():
    """One-launch weight standardisation on fp32 weights vs the fp64 expression (std_conv.py:56-59), forward + backward."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(4)
    shapes = [(64, 3, 7, 7), (64, 64, 1, 1), (256, 64, 3, 3), (33, 5, 1, 1)]
    ws = [(torch.randn(s, generator=g) * 0.3 + 0.05).to(dev).requires_grad_(True) for s in shapes]
    outs = ops.weight_std_all(ws)
    gs = [torch.randn(s, generator=g).to(dev) for s in shapes]
    sum((o * gi).sum() for o, gi in zip(outs, gs)).backward()
    for w, o, gi in zip(ws, outs, gs):
        wd = w.detach().double().requires_grad_(True)
        std, mean = torch.std_mean(wd, dim=[1, 2, 3], keepdim=True, unbiased=False)
        ref = (wd - mean) / (std + 1e-5)
        (ref * gi.double()).sum().backward()
        assert (o.double() - ref).abs().max() <= 1e-5 * ref.abs().max()
        assert (w.grad.double() - wd.grad).abs().max() <= 5e-5 * wd.grad.abs().max()


@pytest.mark.parametrize("N,C,H,W", [(2, 64, 8, 8), (2, 1024, 28, 28), (2, 512, 56, 56), (2, 128, 112, 112), (2, 256, 112, 112), (1, 256, 128, 128),
                                     (2, 96, 6, 6)])
def test_groupnorm_f32_relu_mask_equals_the_residual_path(N, C, H, W, monkeypatch):
    """Round 6: relu(gn(x) + resid) leaves its ReLU mask as one byte per 16-byte vector (acr_groupnorm_fwd_mask_f32) and the backward
    reads that instead of the residual (acr_groupnorm_bwd_mask_f32).  Same expression decides the mask in both directions, so y, dx,
    d(resid), d(gamma), d(beta) must be BIT-identical to the path that re-derives the mask from the residual -- on every kernel
    family: register-resident groups of 8 / 13 slots, the 25-slot forward with the streamed backward (256 x 112^2), and groups too
    large for either (256 x 128^2: both directions streamed)."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(C + H)
    x0 = (torch.randn(N, C, H, W, generator=g) * 1.7 + 0.3).to(dev)
    r0 = torch.randn(N, C, H, W, generator=g).to(dev)
    w0 = (1 + 0.2 * torch.randn(C, generator=g)).to(dev)
    b0 = (0.3 * torch.randn(C, generator=g)).to(dev)
    dy = torch.randn(N, C, H, W, generator=g).to(dev)
    outs = []
    for use_mask in (True, False):
        monkeypatch.setattr(ops, "GN_RELU_MASK", use_mask)
        x, r, w, b = (t.clone().requires_grad_(True) for t in (x0, r0, w0, b0))
        y = ops.groupnorm_act(x, w, b, "add_relu", r)
        assert (len(y.grad_fn.saved_tensors) == 5) and (y.grad_fn.saved_tensors[4].dtype == (torch.uint8 if use_mask else torch.float32))
        (y * dy).sum().backward()
        outs.append((y.detach(), x.grad, r.grad, w.grad, b.grad))
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    assert float((outs[0][0] > 0).float().mean()) > 0.2 and float((outs[0][0] == 0).float().mean()) > 0.2       # the mask does something


@pytest.mark.parametrize("N,C,H,W", [(2, 64, 8, 8), (3, 256, 16, 24), (2, 1024, 28, 28), (1, 64, 224, 224), (2, 256, 112, 112),
                                     (2, 512, 56, 56), (2, 128, 112, 112), (1, 256, 128, 128), (2, 96, 6, 6)])
@pytest.mark.parametrize("act", ["none", "relu", "add_relu"])
def test_groupnorm_f32(N, C, H, W, act):
    """fp32 GroupNorm(32) [+ residual] [+ ReLU] forward/backward (reference precision) vs fp64; includes the largest groups of the
    448^2 stem (8 x 112^2 and 2 x 224^2 floats = 401 KB: 25 register slots per lane forward, streamed backward), the 13-slot
    groups that are register-resident in both directions (512 x 56^2, 128 x 112^2), a group too large for either (256 x 128^2,
    the COCO 512^2 geometry: streamed) and tiny groups (3 x 36)."""
    from acr_wsss_amd import ops
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(C + H)
    x = (torch.randn(N, C, H, W, generator=g) * 1.7 + 0.3).to(dev).requires_grad_(True)
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(dev).requires_grad_(True)
    b = (0.3 * torch.randn(C, generator=g)).to(dev).requires_grad_(True)
    r = torch.randn(N, C, H, W, generator=g).to(dev).requires_grad_(True) if act == "add_relu" else None
    assert ops.groupnorm_fusable(x, r)
    y = ops.groupnorm_act(x, w, b, act, r)
    dy = torch.randn(N, C, H, W, generator=g).to(dev)
    (y * dy).sum().backward()
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    ref = F.group_norm(xd, 32, wd, bd, 1e-5)
    if r is not None:
        rd = r.detach().double().requires_grad_(True)
        ref = ref + rd
    if act != "none":
        ref = ref * (y.detach() > 0).double()               # the kernel's own mask: elements within 1e-7 of 0 may differ
    (ref * dy.double()).sum().backward()
    assert (y.double() - ref).abs().max() <= 1e-5 * ref.abs().max()
    assert (x.grad.double() - xd.grad).abs().max() <= 2e-5 * xd.grad.abs().max()
    assert (w.grad.double() - wd.grad).abs().max() <= 2e-5 * wd.grad.abs().max() + 1e-6
    assert (b.grad.double() - bd.grad).abs().max() <= 2e-5 * bd.grad.abs().max() + 1e-6
    if r is not None:
        assert (r.grad.double() - rd.grad).abs().max() <= 1e-6 * rd.grad.abs().max()
    # deterministic
    x2 = x.detach().clone().requires_grad_(True)
    y2 = ops.groupnorm_act(x2, w.detach(), b.detach(), act, r.detach() if r is not None else None)
    (y2 * dy).sum().backward()
    assert torch.equal(y2, y) and torch.equal(x2.grad, x.grad)


@pytest.mark.parametrize("N,cin,cout,H,W", [(2, 64, 64, 8, 8), (3, 64, 256, 16, 24), (2, 256, 64, 28, 28), (2, 1024, 256, 28, 28),
                                             (2, 128, 512, 56, 56), (1, 64, 256, 112, 112), (2, 1024, 768, 28, 28), (8, 64, 64, 56, 56)])
@pytest.mark.parametrize("math", [0, 1])
def test_conv1x1_f32(N, cin, cout, H, W, math):
    """fp32 NCHW 1x1 convolution on the fp32 GEMM kernels (one z-slice per sample): forward, input gradient with the shortcut's
    gradient added in the epilogue, weight gradient through per-sample slabs -- vs fp64 conv2d.  28x28 = 784 pixels is not a
    multiple of the 32-deep chunk (register-staged kernel for the weight gradient), cout = 64 fills half a tile.  math = 1: split
    products on the bf16 MFMA (the LDS-DMA shapes; the others stay on the exact kernels), same tolerance."""
    from acr_wsss_amd import ops
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(cin + cout + H)
    x = torch.randn(N, cin, H, W, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 1, generator=g) * cin ** -0.5).to(dev).requires_grad_(True)
    assert ops.conv1x1_fusable(x, w, 1)
    y, skip = ops.conv1x1_skip(x, w, None, math)
    dy = torch.randn(N, cout, H, W, generator=g).to(dev)
    ds = torch.randn(N, cin, H, W, generator=g).to(dev)
    ((y * dy).sum() + (skip * ds).sum()).backward()
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    ref = F.conv2d(xd, wd)
    ((ref * dy.double()).sum() + (xd * ds.double()).sum()).backward()
    for n, a, b in (("y", y, ref), ("dx", x.grad, xd.grad), ("dw", w.grad, wd.grad)):
        err = (a.double() - b).abs().max() / b.abs().max()
        assert err <= 1e-5, (n, float(err))


@pytest.mark.parametrize("N,C,H,W,act", [(2, 1024, 24, 24, "add_relu"), (2, 256, 48, 48, "relu"), (1, 256, 96, 96, "none"), (2, 64, 96, 96, "relu")])
def test_groupnorm_f32_small_launch_parts(N, C, H, W, act):
    """GroupNorm forward of a gradient-free pass over a few samples (CAM generation): every (sample, group) is cut into parts
    (two launches, acr_groupnorm_fwd_f32 with ws) -- same values as the one-workgroup-per-group kernel up to the order of the
    fp32 sums, vs float64."""
    from acr_wsss_amd import ops
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(C + H)
    x = (torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(dev)
    r = torch.randn(N, C, H, W, generator=g).to(dev) if act == "add_relu" else None
    gw, gb = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.3 * torch.randn(C, generator=g)).to(dev)
    from acr_wsss_amd import _lib
    assert _lib.load().acr_groupnorm_fwd_ws_floats(N, C, H * W) > 0
    with torch.no_grad():
        y = ops.groupnorm_act(x, gw, gb, act, r)            # no gradient needed: the parts path
    xg = x.clone().requires_grad_(True)
    y1 = ops.groupnorm_act(xg, gw, gb, act, r)               # gradient needed: one workgroup per (sample, group)
    ref = F.group_norm(x.double(), 32, gw.double(), gb.double(), 1e-5)
    if act == "add_relu":
        ref = F.relu(ref + r.double())
    elif act == "relu":
        ref = F.relu(ref)
    assert (y.double() - ref).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert (y - y1).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("M,N,K", [(300, 200, 100), (128, 128, 16), (1000, 768, 772), (2500, 64, 3072), (37, 260, 40)])
def test_gemm_x3_images(M, N, K):
    """The split-product image API (acr_x3_image / acr_x3_image_t / acr_gemm_x3): NT with bias + residual, the input-gradient form
    through the transposed image of the weight, TN on the SAME images of dy and x the other two products read (transposed LDS
    reads), column sums of dy from the image pass -- vs float64 at the fp32 GEMM tests' tolerance; shapes with partial row
    blocks, K not a multiple of the 16-deep stage, N = 64 (half a tile), tails.  The results are bit-identical to
    acr_gemm_f32(math = split) -- it is the same kernels on images it makes itself."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    dy = torch.randn(M, N, generator=g).to(dev)
    xi, wi = ops.x3_image(x), ops.x3_image(w)
    y = torch.empty(M, N, device=dev)
    ops.gemm_x3("nt", xi, wi, y, K, bias=b, aux=res)
    ref = x.double() @ w.double().t() + b.double() + res.double()
    assert (y.double() - ref).abs().max() <= 1e-5 * ref.abs().max()
    y2 = torch.empty(M, N, device=dev)
    ops.gemm_f32_raw("nt", x, w, y2, bias=b, aux=res, math=1)
    assert torch.equal(y, y2)
    db = torch.empty(N, device=dev)
    dyi = ops.x3_image(dy, colsum=db)
    refb = dy.double().sum(0)
    assert (db.double() - refb).abs().max() <= 1e-5 * max(1.0, float(refb.abs().max())) + 1e-5 * float(dy.abs().sum(0).max())
    dx = torch.empty(M, K, device=dev)
    ops.gemm_x3("nt", dyi, ops.x3_image_t(w), dx, N)
    refx = dy.double() @ w.double()
    assert (dx.double() - refx).abs().max() <= 1e-5 * refx.abs().max()
    if N % 4 == 0 and K % 4 == 0:
        dw = torch.empty(N, K, device=dev)
        ops.gemm_x3("tn", dyi, xi, dw, M)
        refw = dy.double().t() @ x.double()
        assert (dw.double() - refw).abs().max() <= 1e-5 * refw.abs().max()
        dw2 = torch.empty(N, K, device=dev)
        db2 = torch.empty(N, device=dev)
        ops.gemm_f32_raw("tn", dy, x, dw2, colsum=db2, math=1)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
        dw3 = torch.empty(N, K, device=dev)
        ops.gemm_x3("tn", dyi, xi, dw3, M)
        assert torch.equal(dw, dw3)                          # fixed slab order


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (1000, 3072, 768), (25120, 3072, 768), (2100, 520, 96)])
def test_gemm_x3_image_epilogues(M, N, K):
    """acr_gemm_x3 act 3 / 4: the product's output leaves the kernel as the next product's image.  The image equals, bit for bit,
    the image pass run over the fp32 result of act 1 / 2 (same erf form, same split), the fp32 GELU' of act 3 equals act 1's, and
    act 4's column sums equal the image pass's (same summation order); covers the K-split tail tiles (1000 x 3072: 192 tiles, all
    tail; 25120 x 3072: 120 of 4728), partial row blocks and a last column tile that is not full (N = 520)."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    xi, wi = ops.x3_image(x), ops.x3_image(w)
    h1, a1 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.gemm_x3("nt", xi, wi, h1, K, bias=b, act=1, c2=a1)
    h3, ai = torch.empty(M, N, device=dev), ops.x3_image_empty(M, N, dev)
    ai.fill_(float("nan"))
    ops.gemm_x3("nt", xi, wi, h3, K, bias=b, act=3, c2=ai)
    assert torch.equal(h1, h3)
    assert torch.equal(ai.view(torch.int32), ops.x3_image(a1).view(torch.int32))
    aux = torch.randn(M, N, generator=g).to(dev)
    d2 = torch.empty(M, N, device=dev)
    ops.gemm_x3("nt", xi, wi, d2, K, aux=aux, act=2)
    db_ref = torch.empty(N, device=dev)
    ref_img = ops.x3_image(d2, colsum=db_ref)
    di, db = ops.x3_image_empty(M, N, dev), torch.empty(N, device=dev)
    di.fill_(float("nan"))
    ops.gemm_x3("nt", xi, wi, None, K, aux=aux, act=4, c2=di, colsum=db, shape=(M, N))
    assert torch.equal(di.view(torch.int32), ref_img.view(torch.int32))
    assert torch.equal(db, db_ref)


def test_split_linears_take_the_image_path():
    """Linear and MLP under split products keep operand images across forward and backward (the path bench.py measures): the
    autograd nodes say so, no fp32 GELU output is kept, and values / gradients equal the per-call path's bit for bit."""
    from acr_wsss_amd import ops
    dev = _dev()
    torch.manual_seed(2)
    fc1, fc2 = torch.nn.Linear(256, 512).to(dev), torch.nn.Linear(512, 256).to(dev)
    x = torch.randn(3, 100, 256, device=dev, requires_grad=True)
    res = torch.randn(3, 100, 256, device=dev)
    out = {}
    for images in (True, "no epilogue images", False):
        ops.X3_IMAGES = bool(images)
        ops.X3_IMAGE_EPILOGUES = images is True
        try:
            for p_ in (x, *fc1.parameters(), *fc2.parameters()):
                p_.grad = None
            y = ops.linear_or_hip(x, fc1, math=1)
            assert y.grad_fn.images == bool(images)
            z = ops.mlp_f32(x, fc1, fc2, resid=res, math=1)
            assert z.grad_fn.images == bool(images) and (not images or z.grad_fn.epi == (images is True))
            (y.sum() * 0.5 + (z * res).sum()).backward()
            out[images] = [t.detach().clone() for t in (y, z, x.grad, fc1.weight.grad, fc1.bias.grad, fc2.weight.grad, fc2.bias.grad)]
        finally:
            ops.X3_IMAGES = ops.X3_IMAGE_EPILOGUES = True
    for other in ("no epilogue images", False):
        for a, b in zip(out[True], out[other]):
            assert torch.equal(a, b)


def test_x3_image_planes_sum_to_the_operand_exactly():
    """An image is the operand: its three bf16 planes, read back through the documented tiling, sum to the fp32 values bit for bit
    (values spanning 2^-20 .. 2^20), and it is zero outside the matrix."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(5)
    rows, cols = 200, 40
    x = (torch.randn(rows, cols, generator=g) * torch.exp2(torch.randint(-20, 21, (rows, cols), generator=g).float())).to(dev)
    img = ops.x3_image(x).view(torch.bfloat16)               # [row block][stage][plane][128 rows][16: two 8-element halves]
    nrb, nkb = (rows + 127) // 128, (cols + 15) // 16
    t = img.view(nrb, nkb, 3, 128, 2, 8).float()
    rr = torch.arange(128, device=dev)
    swap = ((rr >> 3) & 1).bool()
    t = torch.where(swap.view(1, 1, 1, 128, 1, 1), t.flip(4), t)          # halves of rows 8-15 (mod 16) are stored swapped
    full = t.sum(2).permute(0, 2, 1, 3, 4).reshape(nrb * 128, nkb * 16)
    assert torch.equal(full[:rows, :cols], x)
    assert float(full[rows:].abs().max()) == 0.0 and float(full[:, cols:].abs().max()) == 0.0
    xt = ops.x3_image_t(x).view(torch.bfloat16)
    nrb, nkb = (cols + 127) // 128, (rows + 15) // 16
    t = xt.view(nrb, nkb, 3, 128, 2, 8).float()
    t = torch.where(swap.view(1, 1, 1, 128, 1, 1), t.flip(4), t)
    full = t.sum(2).permute(0, 2, 1, 3, 4).reshape(nrb * 128, nkb * 16)
    assert torch.equal(full[:cols, :rows], x.t())
    assert float(full[cols:].abs().max()) == 0.0 and float(full[:, rows:].abs().max()) == 0.0


@pytest.mark.parametrize("wimg", [True, False])
@pytest.mark.parametrize("N,cin,cout,H,W", [(2, 64, 64, 16, 16), (1, 16, 48, 12, 20), (2, 128, 128, 28, 28), (1, 64, 64, 112, 112),
                                             (3, 32, 160, 20, 36), (2, 256, 256, 28, 28), (2, 256, 256, 24, 24), (2, 256, 256, 48, 48),
                                             (32, 128, 128, 56, 56)])
def test_conv3x3_split(N, cin, cout, H, W, wimg, monkeypatch):
    """3x3 stride-1 SAME convolution with split products as an implicit GEMM (csrc/conv3x3.hip): forward, input gradient (same
    kernel, flipped taps) and weight gradient vs fp64 conv2d at the fp32 GEMM tests' tolerance.  Covers cout = 64 (half a tile
    row), pixel tiles that run past H*W and whose rows straddle image rows (W = 20, 28, 36), H != W, 9 cin not a multiple of the
    tile (cin = 16, 32, 64), several pixel parts in the weight gradient (112 x 112), the K-split small launches of CAM generation
    (two views at 24 x 24 and 48 x 48: 20 / 72 workgroups split 7 ways into slabs), a full-size training launch.  wimg: the
    forward / input-gradient launches take the packed weight as a split-product image (acr_conv3x3_x3, the default) or as fp32
    (acr_conv3x3_f32: both operands split in registers)."""
    from acr_wsss_amd import ops
    import torch.nn.functional as F
    monkeypatch.setattr(ops, "CONV3X3_WIMG", wimg)
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(cin + cout + H)
    x = torch.randn(N, cin, H, W, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).to(dev).requires_grad_(True)
    assert ops.conv3x3_fusable(x, w, 1, 1) and not ops.conv3x3_fusable(x, w, 1, 0) and not ops.conv3x3_fusable(x, w, 2, 1)
    y = ops.conv3x3(x, w)
    dy = torch.randn(N, cout, H, W, generator=g).to(dev)
    (y * dy).sum().backward()
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    ref = F.conv2d(xd, wd, padding=1)
    (ref * dy.double()).sum().backward()
    for n, a, b in (("y", y, ref), ("dx", x.grad, xd.grad), ("dw", w.grad, wd.grad)):
        err = (a.double() - b).abs().max() / b.abs().max()
        assert err <= 1e-5, (n, float(err))
    # the border is exact zero padding, not a neighbouring row / channel / sample: a one-hot input's response is the flipped kernel
    x1 = torch.zeros(1, cin, H, W, device=dev)
    x1[0, 3, 0, 0] = 1.0
    x1[0, 5, H - 1, W - 1] = 1.0
    y1 = ops.conv3x3(x1, w.detach())
    r1 = F.conv2d(x1.double(), w.detach().double(), padding=1)
    assert (y1.double() - r1).abs().max() <= 1e-6 * r1.abs().max()
    assert torch.equal(y1 == 0, r1 == 0)
    # bitwise reproducible (fixed slab order in the weight gradient)
    x2 = x.detach().clone().requires_grad_(True)
    w2 = w.detach().clone().requires_grad_(True)
    y2 = ops.conv3x3(x2, w2)
    (y2 * dy).sum().backward()
    assert torch.equal(y2, y) and torch.equal(x2.grad, x.grad) and torch.equal(w2.grad, w.grad)


def test_x3_image_many_equals_the_single_image_passes():
    """acr_x3_image_many (all weight images of a stem group / of the blocks in one launch, sources described as strided views) must
    write, bit for bit, what the single-image passes write: W and W^T of a Linear / 1x1 convolution (acr_x3_image, acr_x3_image_t),
    the packed 3x3 weight w[co][t * ci + c] and its input-gradient pack w[o][c][8 - t'] as (ci x 9 co) (acr_x3_image of the
    permuted copies ops.Conv3x3Fn used to make) -- shapes with partial row blocks and a contraction that is not a multiple of 64."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(11)
    w1 = torch.randn(200, 72, generator=g).to(dev)                   # (out, in): rows not a multiple of 128, K = 72
    w3 = torch.randn(48, 32, 3, 3, generator=g).to(dev)              # (co, ci, 3, 3)
    co, ci = w3.shape[:2]
    specs = [(w1, 0, 200, 72, 72, 72, 0, 1), (w1, 0, 72, 200, 1, 200, 0, 72),
             (w3, 0, co, 9 * ci, 9 * ci, ci, 1, 9), (w3, 8, ci, 9 * co, 9, co, -1, 9 * ci)]
    got = ops.x3_image_many(specs, dev)
    wp = w3.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous()
    wd = w3.flip(2, 3).permute(1, 2, 3, 0).reshape(ci, 9 * co).contiguous()
    want = [ops.x3_image(w1), ops.x3_image_t(w1), ops.x3_image(wp), ops.x3_image(wd)]
    for i, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape and torch.equal(a.view(torch.int32), b.view(torch.int32)), i
    # and the per-owner caches prebuild_weight_images fills are what weight_image would have built
    lin = torch.nn.Linear(72, 200, bias=False).to(dev)
    with torch.no_grad():
        lin.weight.copy_(w1)
    assert ops.prebuild_weight_images([lin]) == 1
    a, at = ops.weight_image(lin.weight, lin), ops.weight_image(lin.weight, lin, True)
    assert torch.equal(a.view(torch.int32), want[0].view(torch.int32)) and torch.equal(at.view(torch.int32), want[1].view(torch.int32))
    with torch.no_grad():
        lin.weight.mul_(2.0)                                         # version bump: the cache entry is stale, a fresh image is built
    assert not torch.equal(ops.weight_image(lin.weight, lin).view(torch.int32), a.view(torch.int32))


def test_conv3x3_reads_nothing_outside_its_input():
    """acr_conv3x3_f32 / acr_conv3x3_wgrad_f32 used to dereference addresses up to ACR_CONV3X3_PAD floats outside x
    (shifted tap windows, values masked) and asked the caller for readable slack.  Now the workgroups at the tensor's two ends
    clamp their reads into it.  The input here IS a whole device allocation: 16 MiB, a multiple of the caching allocator's 2 MiB
    granule and above its 10 MiB small-block limit, so the tensor starts at the first and ends at the last byte of its own
    hipMalloc block -- called straight through the C ABI on exactly those pointers, forward (as x) and both backward launches (as
    dy / x), against fp64 conv2d.  (A read outside the block is a GPU memory fault, not a wrong value: the masked lanes never
    showed up in results, which is why the old contract could not be tested from the outside.)"""
    from acr_wsss_amd import _lib as L
    import torch.nn.functional as F
    dev = _dev()
    lib = L.load()
    N, C, H, W = 4, 64, 128, 128
    g = torch.Generator(device="cpu").manual_seed(5)
    pool = torch.cuda.MemPool()                             # a private pool: nothing cached in it, so the request below is a fresh hipMalloc
    with torch.cuda.use_mem_pool(pool):
        x = torch.empty(N * C * H * W, dtype=torch.float32, device=dev)              # its own 16 MiB segment
    assert x.untyped_storage().nbytes() == 16 << 20 and x.storage_offset() == 0
    seg = [b for b in torch.cuda.memory_snapshot() if b["address"] <= x.data_ptr() < b["address"] + b["total_size"]]
    assert seg and seg[0]["address"] == x.data_ptr() and seg[0]["total_size"] == 16 << 20, "the tensor must fill its hipMalloc block"
    x.copy_(torch.randn(N * C * H * W, generator=g))
    x = x.view(N, C, H, W)
    w = (torch.randn(C, C, 3, 3, generator=g) * (9 * C) ** -0.5).to(dev)
    wp = w.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous()
    from acr_wsss_amd import ops
    y = torch.empty(N, C, H, W, device=dev)
    L.check(lib.acr_conv3x3_f32(1, L.ptr(wp), L.ptr(x), L.ptr(y), N, C, C, H, W, None, L.stream_ptr()), "acr_conv3x3_f32")
    ref = F.conv2d(x.double(), w.double(), padding=1)
    assert (y.double() - ref).abs().max() <= 1e-5 * ref.abs().max()
    y2 = torch.empty(N, C, H, W, device=dev)                # the weight-image kernel: the same windows, the same careful path
    L.check(lib.acr_conv3x3_x3(L.ptr(ops.x3_image(wp)), L.ptr(x), L.ptr(y2), N, C, C, H, W, None, L.stream_ptr()), "acr_conv3x3_x3")
    assert (y2.double() - ref).abs().max() <= 1e-5 * ref.abs().max()
    # the same block as dy of the input gradient and as x of the weight gradient (dy: another exact-fit block)
    wd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(C, 9 * C).contiguous()
    dx = torch.empty(N, C, H, W, device=dev)
    L.check(lib.acr_conv3x3_f32(1, L.ptr(wd), L.ptr(x), L.ptr(dx), N, C, C, H, W, None, L.stream_ptr()), "acr_conv3x3_f32 (dX)")
    refdx = F.conv_transpose2d(x.double(), w.double(), padding=1)
    assert (dx.double() - refdx).abs().max() <= 1e-5 * refdx.abs().max()
    dy = torch.randn(N, C, H, W, generator=g).to(dev)
    ws = torch.empty(lib.acr_conv3x3_wgrad_ws_floats(N, C, C, H, W), device=dev)
    dwp = torch.empty(C, 3, 3, C, device=dev)
    L.check(lib.acr_conv3x3_wgrad_f32(1, L.ptr(dy), L.ptr(x), N, C, C, H, W, L.ptr(ws), L.ptr(dwp), L.stream_ptr()), "acr_conv3x3_wgrad_f32")
    wdbl = w.double().requires_grad_(True)
    (F.conv2d(x.double(), wdbl, padding=1) * dy.double()).sum().backward()
    assert (dwp.permute(0, 3, 1, 2).double() - wdbl.grad).abs().max() <= 1e-5 * wdbl.grad.abs().max()


def _same_conv_ref(xd, wd, stride):
    """fp64 TF-SAME convolution as the reference computes it: F.pad (odd pixel right / bottom) + F.conv2d (std_conv.py:56-65)."""
    import math
    import torch.nn.functional as F
    k = wd.shape[2]
    pads = []
    for n in (xd.shape[3], xd.shape[2]):
        t = max((math.ceil(n / stride) - 1) * stride + k - n, 0)
        pads += [t // 2, t - t // 2]
    return F.conv2d(F.pad(xd, pads), wd, stride=stride)


@pytest.mark.parametrize("N,cin,cout,k,H,W,grads", [(2, 128, 128, 3, 112, 112, True), (2, 256, 256, 3, 56, 56, True), (3, 32, 48, 3, 40, 64, True),
                                                   (2, 3, 64, 7, 448, 448, True), (2, 3, 64, 7, 64, 96, True), (32, 128, 128, 3, 112, 112, True),
                                                   (2, 64, 64, 3, 57, 79, False), (2, 32, 32, 3, 58, 79, False), (2, 3, 64, 7, 183, 250, False),
                                                   (1, 3, 64, 7, 250, 183, False), (2, 3, 64, 7, 24, 40, False)])
def test_conv_s2_split(N, cin, cout, k, H, W, grads):
    """The strided SAME convolutions of the stem under split products (ops.ConvS2Fn: space-to-depth + the tap-table forms of the 3x3
    kernels): the two stride-2 3x3 convolutions at their training shapes (and at the full batch), the 7x7 stem convolution at
    448 x 448, small and non-square maps; forward, input gradient (four phase launches + depth-to-space) and weight gradient vs
    fp64 pad + conv2d.  Forward-only cases: odd heights / widths (CAM generation's scales; the SAME padding in front is then 1 resp.
    3 and the tap table a different one), launches small enough for the K-split slabs."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(cin + cout + H + k)
    x = torch.randn(N, cin, H, W, generator=g).to(dev).requires_grad_(grads and cin > 3)
    w = (torch.randn(cout, cin, k, k, generator=g) * (k * k * cin) ** -0.5).to(dev).requires_grad_(grads)
    with torch.set_grad_enabled(grads):
        assert ops.conv_s2_fusable(x, w, 2, 1) and not ops.conv_s2_fusable(x, w, 2, 0) and not ops.conv_s2_fusable(x, w, 1, 1)
        y = ops.conv_s2(x, w)
    xd, wd = x.detach().double().requires_grad_(x.requires_grad), w.detach().double().requires_grad_(grads)
    ref = _same_conv_ref(xd, wd, 2)
    assert y.shape == ref.shape
    pairs = [("y", y, ref)]
    if grads:
        dy = torch.randn(y.shape, generator=g).to(dev)
        (y * dy).sum().backward()
        (ref * dy.double()).sum().backward()
        pairs.append(("dw", w.grad, wd.grad))
        if x.requires_grad:
            pairs.append(("dx", x.grad, xd.grad))
    for n, a, b in pairs:
        err = (a.double() - b).abs().max() / b.abs().max()
        assert err <= 1e-5, (n, float(err))
    # exact zero padding: a one-hot input answers with single kernel entries, zero elsewhere
    x1 = torch.zeros(1, cin, H, W, device=dev)
    x1[0, 1, 0, 0] = 1.0
    x1[0, 2, H - 1, W - 1] = 1.0
    with torch.no_grad():
        y1 = ops.conv_s2(x1, w.detach())
    r1 = _same_conv_ref(x1.double(), w.detach().double(), 2)
    assert (y1.double() - r1).abs().max() <= 1e-6 * r1.abs().max() and torch.equal(y1 == 0, r1 == 0)
    if grads:                                               # bitwise reproducible
        x2 = x.detach().clone().requires_grad_(x.requires_grad)
        w2 = w.detach().clone().requires_grad_(True)
        y2 = ops.conv_s2(x2, w2)
        (y2 * dy).sum().backward()
        assert torch.equal(y2, y) and torch.equal(w2.grad, w.grad) and (not x.requires_grad or torch.equal(x2.grad, x.grad))


def test_conv_s2_group_images_and_exact_fit_block():
    """(i) The images StdConv2dSame.image_specs describes for the stride-2 convolutions (made with everybody else's in ONE
    acr_x3_image_many launch) give bit for bit what ops.ConvS2Fn builds for itself.  (ii) The tap-table kernels read nothing outside
    their input: the space-to-depth tensor IS a whole 16 MiB hipMalloc block (as in test_conv3x3_reads_nothing_outside_its_input),
    forward and weight gradient called through the C ABI on exactly that pointer; dy of the input-gradient launches likewise."""
    from acr_wsss_amd import _lib as L, ops
    from acr_wsss_amd.backbone import StdConv2dSame
    dev = _dev()
    lib = L.load()
    g = torch.Generator(device="cpu").manual_seed(9)
    for k, ci, co, hw in ((3, 32, 48, 32), (7, 3, 64, 64)):
        conv = StdConv2dSame(ci, co, k, stride=2).to(dev)
        conv.acr_math = 1
        w = (torch.randn(co, ci, k, k, generator=g) * 0.1).to(dev).requires_grad_(True)
        imgs = tuple(ops.x3_image_many(list(conv.image_specs(w.detach())), dev))
        assert len(imgs) == (5 if k == 3 else 1)
        x = torch.randn(2, ci, hw, hw, generator=g).to(dev).requires_grad_(k == 3)
        outs = []
        for im in (None, imgs):
            x.grad = w.grad = None
            y = ops.conv_s2(x, w, im)
            y.square().sum().backward()
            outs.append((y.detach(), w.grad.clone(), x.grad.clone() if k == 3 else None))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert k == 7 or torch.equal(outs[0][2], outs[1][2])
    # (ii) exact-fit blocks
    N, C, H2, W2 = 2, 64, 128, 64                           # xs: N x 4C x H2 x W2 floats = 16 MiB
    pool = torch.cuda.MemPool()
    with torch.cuda.use_mem_pool(pool):
        xs = torch.empty(N * 4 * C * H2 * W2, dtype=torch.float32, device=dev)
        dyb = torch.empty(16 << 18, dtype=torch.float32, device=dev)           # dy of the input gradient: co = 256 rows
    for t in (xs, dyb):
        seg = [b for b in torch.cuda.memory_snapshot() if b["address"] <= t.data_ptr() < b["address"] + b["total_size"]]
        assert seg and seg[0]["address"] == t.data_ptr() and seg[0]["total_size"] == 16 << 20, "the tensor must fill its hipMalloc block"
    x = torch.randn(N, C, 2 * H2, 2 * W2, generator=g).to(dev)
    L.check(lib.acr_space_to_depth2_f32(L.ptr(x), L.ptr(xs), N, C, 2 * H2, 2 * W2, 4 * C, L.stream_ptr()), "s2d")
    xs4 = xs.view(N, 4 * C, H2, W2)
    for py in range(2):
        for px in range(2):
            assert torch.equal(xs4[:, (py * 2 + px) * C:(py * 2 + px + 1) * C], x[:, :, py::2, px::2])
    back = torch.empty_like(x)
    L.check(lib.acr_depth_to_space2_f32(L.ptr(xs), L.ptr(back), N, C, 2 * H2, 2 * W2, L.stream_ptr()), "d2s")
    assert torch.equal(back, x)
    w = (torch.randn(C, C, 3, 3, generator=g) * (9 * C) ** -0.5).to(dev)
    plan = ops.conv_s2_plan(3, C, 2 * H2, 2 * W2, dev)
    y = torch.empty(N, C, H2, W2, device=dev)
    L.check(lib.acr_conv_taps_x3(L.ptr(ops.x3_image(plan.pack(w))), L.ptr(xs), L.ptr(y), N, C, C, H2, W2, 9, plan.fwd[0], plan.fwd[1], plan.fwd[2],
                                 4 * C, C, None, L.stream_ptr()), "acr_conv_taps_x3")
    ref = _same_conv_ref(x.double(), w.double(), 2)
    assert (y.double() - ref).abs().max() <= 1e-5 * ref.abs().max()
    dy = torch.randn(N, C, H2, W2, generator=g).to(dev)
    ws = torch.empty(lib.acr_conv_taps_wgrad_ws_floats(N, C, C, H2, W2, 9), device=dev)
    dwp = torch.empty(C, 9 * C, device=dev)
    L.check(lib.acr_conv_taps_wgrad_f32(1, L.ptr(dy), L.ptr(xs), N, C, C, H2, W2, 9, plan.fwd[0], plan.fwd[1], plan.fwd[2], 4 * C, L.ptr(ws), L.ptr(dwp),
                                        L.stream_ptr()), "acr_conv_taps_wgrad_f32")
    wdbl = w.double().requires_grad_(True)
    (_same_conv_ref(x.double(), wdbl, 2) * dy.double()).sum().backward()
    assert (plan.unpack_grad(dwp, w.shape).double() - wdbl.grad).abs().max() <= 1e-5 * wdbl.grad.abs().max()
    # input gradient: dy (N x 256 x 64 x 128 floats = 16 MiB exactly) read with negative shifts by the phase launches
    co2, ci2, hh, wh = 256, 32, 64, 128
    dy2 = dyb.view(N, co2, hh, wh)
    dy2.copy_(torch.randn(dy2.shape, generator=g))
    w2 = (torch.randn(co2, ci2, 3, 3, generator=g) * (9 * ci2) ** -0.5).to(dev)
    plan2 = ops.conv_s2_plan(3, ci2, 2 * hh, 2 * wh, dev)
    pim = ops.x3_image_many(plan2.dgrad_specs(plan2.pack_dgrad(w2)), dev)
    dxs = torch.empty(N, 4 * ci2, hh, wh, device=dev)
    for (p, _, n, tab), im in zip(plan2.phases, pim):
        L.check(lib.acr_conv_taps_x3(L.ptr(im), L.ptr(dy2), L.c_void_p(dxs.data_ptr() + 4 * p * ci2 * hh * wh), N, ci2, co2, hh, wh, n, tab[0], tab[1],
                                     tab[2], co2, 4 * ci2, None, L.stream_ptr()), "acr_conv_taps_x3 (dX)")
    dx = torch.empty(N, ci2, 2 * hh, 2 * wh, device=dev)
    L.check(lib.acr_depth_to_space2_f32(L.ptr(dxs), L.ptr(dx), N, ci2, 2 * hh, 2 * wh, L.stream_ptr()), "d2s")
    xin = torch.zeros(N, ci2, 2 * hh, 2 * wh, dtype=torch.float64, device=dev, requires_grad=True)
    (_same_conv_ref(xin, w2.double(), 2) * dy2.double()).sum().backward()
    assert (dx.double() - xin.grad).abs().max() <= 1e-5 * xin.grad.abs().max()


@pytest.mark.parametrize("dtype,math", [(torch.float32, 0), (torch.float32, 1), (torch.bfloat16, 0)])
def test_stem_kernels_full_size_are_per_sample(dtype, math):
    """GroupNorm (+ residual + ReLU) and the NCHW 1x1 convolution at the largest launches of the BASELINE step (32 views, 256 channels
    at 112 x 112: 103 M elements per tensor), through a size-independent property: both are per sample in y and dx, so the last of the
    32 samples must equal the same sample run alone, bit for bit (the weight gradients sum over samples and are left to the fp64
    tests at small N).  math = 1: the split-product 1x1 kernels the bench's headline launches (weight image + in-register split)."""
    from acr_wsss_amd import ops
    dev = _dev()
    N, C, Hh, Ww = 32, 256, 112, 112
    g = torch.Generator(device="cpu").manual_seed(3)
    one = lambda *shape: torch.randn(*shape, generator=g).to(dev).to(dtype)
    x, r, dy = one(N, C, Hh, Ww), one(N, C, Hh, Ww), one(N, C, Hh, Ww)
    gw, gb = (1 + 0.2 * one(C)).contiguous(), (0.3 * one(C)).contiguous()
    cw = (one(64, C, 1, 1) * C ** -0.5).contiguous()
    dyc = one(N, 64, Hh, Ww)
    res = []
    # under split products a launch of fewer than 192 tiles (64 x 256 tiles for <= 64 output channels: 49 per sample here) is cut
    # along the CONTRACTION into slabs (CAM generation's small launches, conv1x1_ksplit) -- another grouping of the same fp32 sum,
    # equal to rounding only (the fp64 value tests cover it); the bit-for-bit comparison therefore runs the sample in the smallest
    # batch on the unsplit path
    sub = 4 if math else 1
    for sl, slc in ((slice(0, N), slice(0, N)), (slice(N - 1, N), slice(N - sub, N))):
        xi = x[sl].clone().requires_grad_(True)
        ri = r[sl].clone().requires_grad_(True)
        assert ops.groupnorm_fusable(xi, ri) and ops.conv1x1_fusable(xi, cw, 1)
        y = ops.groupnorm_act(xi, gw, gb, "add_relu", ri)
        (y.float() * dy[sl].float()).sum().backward()
        gn = (y.detach()[-1].clone(), xi.grad[-1].clone(), ri.grad[-1].clone())
        xc = x[slc].clone().requires_grad_(True)
        yc = ops.conv1x1(xc, cw, None, math)
        (yc.float() * dyc[slc].float()).sum().backward()
        res.append(gn + (yc.detach()[-1].clone(), xc.grad[-1].clone()))
        del xi, ri, y, xc, yc
        torch.cuda.empty_cache()
    for name, a, b in zip(("gn y", "gn dx", "gn dres", "conv y", "conv dx"), res[0], res[1]):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a, b), (name, float((a.float() - b.float()).abs().max()))


@pytest.mark.parametrize("N,C,S", [(32, 64, 112), (32, 256, 28)])
def test_conv3x3_full_size_is_per_sample(N, C, S):
    """The stem's 3x3 convolutions (split products, csrc/conv3x3.hip) at the BASELINE batch -- 32 views x 64 channels x 112^2 (stage 0,
    the largest launch) and 32 x 256 x 28^2 (stage 2) -- through the same per-sample property: y and dx of the last sample equal the
    same sample in a batch of half the size bit for bit, and the weight gradient of the batch (summed over samples in a fixed slab order) agrees with the
    fp64 sum of per-sample gradients at the fp32 GEMM tests' tolerance."""
    from acr_wsss_amd import ops
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(N + C + S)
    x = torch.randn(N, C, S, S, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) * (9 * C) ** -0.5).to(dev)
    dy = torch.randn(N, C, S, S, generator=g).to(dev)
    res = []
    # the sub-batch stays on the unsplit path (>= 192 tiles: launches below that are cut along the contraction into slabs, another
    # grouping of the same fp32 sum -- covered by test_conv3x3_split against fp64)
    for sl in (slice(0, N), slice(N // 2, N)):
        xi = x[sl].clone().requires_grad_(True)
        wi = w.clone().requires_grad_(True)
        assert ops.conv3x3_fusable(xi, wi, 1, 1)
        y = ops.conv3x3(xi, wi)
        (y * dy[sl]).sum().backward()
        res.append((y.detach()[-1].clone(), xi.grad[-1].clone(), wi.grad.clone()))
        del xi, y
    for name, a, b in zip(("y", "dx"), res[0][:2], res[1][:2]):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), (name, float((a - b).abs().max()))
    # last sample against fp64, and the batch weight gradient against fp64 in chunks of 4 samples (memory)
    xd = x[-1:].double().requires_grad_(True)
    ref = F.conv2d(xd, w.double(), padding=1)
    (ref * dy[-1:].double()).sum().backward()
    assert (res[0][0].double() - ref[0]).abs().max() <= 1e-5 * ref.abs().max()
    assert (res[0][1].double() - xd.grad[0]).abs().max() <= 1e-5 * xd.grad.abs().max()
    dw = torch.zeros_like(w, dtype=torch.float64)
    for i in range(0, N, 4):
        wd = w.double().requires_grad_(True)
        (F.conv2d(x[i:i + 4].double(), wd, padding=1) * dy[i:i + 4].double()).sum().backward()
        dw += wd.grad
    assert (res[0][2].double() - dw).abs().max() <= 1e-5 * dw.abs().max()


def test_f32_input_gradients_on_cached_transposes_are_bit_identical():
    """fp32 mode: with a current (in, out) copy of the weight (ops.WeightTransposes, refreshed after every optimizer step by
    train.refresh_weight_transposes) the Linear / fused-MLP input gradients run as NT GEMMs on the copy instead of NN on W as
    stored.  Same k-ordered fp32 fmaf chains -> bit-identical gradients; a weight changed behind the cache's back (version
    mismatch) is never served stale."""
    from acr_wsss_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(5)
    fc1 = torch.nn.Linear(768, 3072).to(dev)
    fc2 = torch.nn.Linear(3072, 768).to(dev)
    qkv = torch.nn.Linear(768, 2304).to(dev)
    head = torch.nn.Linear(768, 20).to(dev)                                  # 20 rows: not cached, stays NN
    x = torch.randn(3, 197, 768, generator=g).to(dev)
    dy = torch.randn(3, 197, 768, generator=g).to(dev)
    dq = torch.randn(3, 197, 2304, generator=g).to(dev)

    def grads():
        xa = x.clone().requires_grad_(True)
        ops.mlp_f32(xa, fc1, fc2, None).backward(dy)
        xb = x.clone().requires_grad_(True)
        ops.linear_or_hip(xb, qkv).backward(dq)
        return xa.grad, xb.grad

    base = grads()                                                           # no copies yet: NN on W as stored
    wt = ops.WeightTransposes([fc1, fc2, qkv, head], dtype=torch.float32)
    assert len(wt.lins) == 3
    wt.refresh()
    assert torch.equal(fc1._acr_wt, fc1.weight.t()) and torch.equal(qkv._acr_wt, qkv.weight.t())
    assert ops.weight_t(fc2.weight, fc2, make=False) is fc2._acr_wt
    cached = grads()
    for a, b in zip(base, cached):
        assert torch.equal(a, b)
    with torch.no_grad():
        fc1.weight.mul_(0.5)                                                 # in-place update without a refresh
    assert ops.weight_t(fc1.weight, fc1, make=False) is None                 # stale copy is not served ...
    xa = x.clone().requires_grad_(True)
    ops.mlp_f32(xa, fc1, fc2, None).backward(dy)                             # ... and the gradient follows the new weight
    ops.F32_WT = False
    try:
        xr = x.clone().requires_grad_(True)
        ops.mlp_f32(xr, fc1, fc2, None).backward(dy)
    finally:
        ops.F32_WT = True
    assert torch.equal(xa.grad, xr.grad) and not torch.equal(xa.grad, base[0])
    wt.refresh()
    assert ops.weight_t(fc1.weight, fc1, make=False) is fc1._acr_wt and torch.equal(fc1._acr_wt, fc1.weight.t())


def test_kernel_timer_brackets_first_launch_per_step():
    """ops.KernelTimer (bench.py's in-step kernel timing): one sample per key and step, on the launch stream, and nothing is
    recorded while it is off."""
    from acr_wsss_amd import ops
    dev = _dev()
    lin = torch.nn.Linear(256, 512).to(dev)
    x = torch.randn(4, 64, 256, device=dev)
    assert ops.KERNEL_TIMER is None
    ops.linear_or_hip(x, lin)
    t = ops.KernelTimer()
    ops.KERNEL_TIMER = t
    try:
        for _ in range(3):
            t.next_step()
            ops.linear_or_hip(x, lin)
            ops.linear_or_hip(x, lin)                                        # second launch of the same shape in a step: not timed
    finally:
        ops.KERNEL_TIMER = None
    s = t.collect()
    assert list(s) == ["gemm_f32_nt 256x512x256"] and len(s["gemm_f32_nt 256x512x256"]) == 3
    assert all(0 < ms < 50 for ms in s["gemm_f32_nt 256x512x256"])


def test_c_abi_launches_capture_into_a_hip_graph():
    """include/acr_hip.h promises: every entry point only enqueues work on the caller's stream (no allocation, no
    synchronisation) -> a sequence of them can be captured into a hipGraph and replayed.  Attention forward + backward (fp32,
    head-mean and G term on) and an fp32 GEMM with epilogue, captured once, replayed on new input values."""
    from acr_wsss_amd import _lib as L, ops
    lib = L.load()
    dev = _dev()
    B, H, T = 2, 12, 197
    g = torch.Generator(device="cpu").manual_seed(3)
    qkv = torch.randn(B, T, 3 * H * 64, generator=g).to(dev)
    d_o = torch.randn(B, T, H * 64, generator=g).to(dev)
    gm = (torch.randn(B, T, ops.pad4(T), generator=g).to(dev) * 1e-2)[:, :, :T]
    o = torch.empty(B, T, H * 64, device=dev)
    lse2 = torch.empty(B, H, T, device=dev)
    pm = torch.empty(B, T, T, device=dev)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, H, T, device=dev)
    x = torch.randn(B * T, 768, generator=g).to(dev)
    w = (torch.randn(768, 768, generator=g) * 0.03).to(dev)
    bias = torch.randn(768, generator=g).to(dev)
    y = torch.empty(B * T, 768, device=dev)
    d = ops._desc(B, H, T, torch.float32)
    qp, kp, vp = ops._qkv_ptrs(qkv, H)
    dqp, dkp, dvp = ops._qkv_ptrs(dqkv, H)

    def launch():
        st = L.stream_ptr()
        L.check(lib.acr_attn_fwd(d, qp, kp, vp, L.ptr(o), L.ptr(lse2), L.ptr(pm), T * T, T, st), "fwd")
        L.check(lib.acr_attn_bwd(d, qp, kp, vp, L.ptr(o), L.ptr(d_o), L.ptr(lse2), L.ptr(gm), gm.stride(0), gm.stride(1), dqp, dkp, dvp,
                                 L.ptr(delta), st), "bwd")
        L.check(lib.acr_gemm_f32(0, 0, 0, L.ptr(o.view(B * T, 768)), 768, L.ptr(w), 768, L.ptr(bias), L.ptr(x), 768, L.ptr(y), 768, None, None,
                                 B * T, 768, 768, None, st), "gemm")

    def eager():
        launch()
        torch.cuda.synchronize()
        return o.clone(), pm.clone(), dqkv.clone(), y.clone()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        launch()                                                             # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        launch()
    for seed in (11, 12):
        g2 = torch.Generator(device="cpu").manual_seed(seed)
        qkv.copy_(torch.randn(B, T, 3 * H * 64, generator=g2))
        d_o.copy_(torch.randn(B, T, H * 64, generator=g2))
        ref = eager()
        for t in (o, pm, dqkv, y):
            t.zero_()
        graph.replay()
        torch.cuda.synchronize()
        for got, want in zip((o, pm, dqkv, y), ref):
            assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,pitch", [(16, 20, 20), (3, 80, 96), (300, 7, 7), (1, 1, 4)])
def test_mlsm_loss(N, C, pitch):
    """ops.mlsm_loss (csrc/mlsm.hip, one launch each way) against F.multilabel_soft_margin_loss in float64: the loss of
    train_acr.py:160-161 and its gradient w.r.t. the logits, with an upstream gradient that is not 1, for logits whose rows sit in a
    wider buffer and logits far in both tails of the log-sigmoid."""
    from acr_wsss_amd import ops
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(N * 131 + C)
    big = (torch.randn(N, pitch, generator=g) * 6.0).to(dev)
    if N > 2:
        big[0, 0], big[1, 0] = 80.0, -80.0
    x = big[:, :C].requires_grad_(True)
    y = (torch.rand(N, C, generator=g) > 0.7).float().to(dev)
    loss = ops.mlsm_loss(x, y)
    (loss * 2.5).backward()
    xd = big[:, :C].double().detach().requires_grad_(True)
    ref = F.multilabel_soft_margin_loss(xd, y.double())
    (ref * 2.5).backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref))
    assert (x.grad.double() - xd.grad).abs().max() <= 2e-6 * xd.grad.abs().max()
    stock = F.multilabel_soft_margin_loss(big[:, :C].detach(), y)
    assert abs(float(loss) - float(stock)) <= 1e-6 * abs(float(stock))


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(3, 100), (2, 785), (1, 32), (5, 197), (1, 128), (4, 768)])
def test_attention_output_image_is_the_pass_image(B, T, monkeypatch):
    """Split products: the attention forward writes its output ALSO as the split-product image the Linear behind it reads
    (acr_attn_fwd_scores_oimg) -- bit for bit what the image pass over o writes (acr_x3_image), padding rows of the last 128-row block
    included (round 6: zeroed by the kernel's epilogue, not by the host -- the image buffer arrives NaN-primed here), and o / the
    head mean themselves are unchanged."""
    from acr_wsss_amd import ops
    dev = _dev()
    H = 12
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + T)
    qkv = (1.5 * torch.randn(B, T, 3 * H * 64, generator=g)).to(dev).requires_grad_(True)
    stack = ops.MeanStack(B, 1, T, dev)
    o1, pm1 = ops.attention_core(qkv, H, stack, 0, None, 1)
    o1, pm1 = o1.detach().clone(), pm1.detach().clone()
    empty = ops.x3_image_empty
    monkeypatch.setattr(ops, "x3_image_empty", lambda rows, cols, device: empty(rows, cols, device).fill_(float("nan")))
    o2, pm2, img = ops.attention_core_oimg(qkv, H, stack, 0, None, 1)
    monkeypatch.setattr(ops, "x3_image_empty", empty)
    assert img is not None and torch.equal(o1, o2) and torch.equal(pm1, pm2)
    ref = ops.x3_image(o2.detach().reshape(B * T, H * 64))
    assert img.shape == ref.shape and torch.equal(img.view(torch.int32), ref.view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("W", [4, 8, 12])
def test_conv3x3_narrow_maps_are_widened_not_handed_to_the_library(W, monkeypatch):
    """Split products: a 3x3 SAME convolution of a map narrower than the kernels' 16-pixel rows (the last stage of CAM generation at
    scale 0.5 is 12 x 12) runs on the HIP kernels over a copy widened with zero columns -- which ARE the SAME padding of the last real
    column -- and is cut back: vs float64 conv2d of the standardised weight, and F.conv2d must not be reached."""
    from acr_wsss_amd import backbone
    import torch.nn.functional as F
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(W)
    conv = backbone.StdConv2dSame(64, 96, 3).to(dev)
    conv.acr_math = 1
    x = torch.randn(2, 64, 12, W, generator=g).to(dev)
    w_hat = conv.standardized_weight().detach()
    ref = F.conv2d(x.double(), w_hat.double(), None, 1, 1)
    def boom(*a, **k):
        raise AssertionError("F.conv2d reached")
    monkeypatch.setattr(backbone.F, "conv2d", boom)
    with torch.no_grad():
        y = conv(x)
    assert y.shape == ref.shape and y.is_contiguous()
    assert (y.double() - ref).abs().max() <= 1e-5 * ref.abs().max()
