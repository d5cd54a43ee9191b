"""GPU lab: accuracy of a THREE-MFMA product -- a two-way fp16 split (11 + 11 significand bits) with exact
power-of-two scales per contraction row, terms a0 b0 + a0 b1 + a1 b0 -- against the shipped six-term bf16 x 3 split and the exact
fp32 MFMA, on the operands of tests/test_kernels_gpu.py::test_split_math_adversarial_operands and on the 448^2 reference fixture.

The candidate is EMULATED, not built: each operand is scaled per row of the contraction (a power of two that puts the row's
largest magnitude in [2^14, 2^15): fp16 tops out at 65504), cut into p0 = fp16(x s), p1 = fp16(x s - p0), and the three products
of fp16-VALUED fp32 matrices are run through the exact-fp32 MFMA kernel (acr_gemm_f32, math = 0): an fp16 x fp16 product has 22
significand bits and is exact in fp32, so what that kernel adds is what an fp16 MFMA with fp32 accumulation adds.  The three partial
results are summed in fp32 and unscaled exactly.  (A real kernel would interleave the three terms in one accumulator: same error
terms, one rounding order instead of another.)

Part 1: NT / NN / TN at 384 x 256 x 768, cases range / cancel / underflow + a plain random case at a block shape: normalised errors
        (|c - c64| / sum_k |a||b|), exact vs bf16x3 vs fp16x2, and the test's acceptance rule (<= 2 x the exact kernel's own error).
Part 2: the reference fixture train_hybrid_448_b1 (BASELINE geometry, batch 1) with every block Linear -- forward, input gradient,
        weight gradient -- on the emulated product (attention and stem stay on the shipped split products): loss terms, maps and
        fixture gradients against the fixture, beside the shipped arithmetics, at the tests' tolerances.
usage: python scripts/lab/split_fp16x2_accuracy.py [--no-model]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from acr_wsss_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


def pow2_scale(x, dim):
    """Exact power of two per slice along `dim` that puts the slice's largest magnitude in [2^14, 2^15)."""
    m = x.abs().amax(dim=dim, keepdim=True)
    e = torch.floor(torch.log2(m.clamp_min(2.0 ** -100))).clamp(-100.0, 100.0)
    return torch.where(m > 0, torch.exp2(14.0 - e), torch.ones_like(m))         # an all-zero slice keeps scale 1


def split16(x, s):
    xs = x * s                                              # exact: s is a power of two (barring underflow of tiny elements)
    p0 = xs.half()
    p1 = (xs - p0.float()).half()
    return p0.float(), p1.float()


def gemm_nt_exact(a, b):
    """c = a b^T on the exact-fp32 MFMA kernel (fp32 accumulation)."""
    k = a.shape[1]
    if k % 32:                                              # zero columns change no sum; the kernel wants 16-byte aligned rows
        pad = 32 - k % 32
        a, b = torch.nn.functional.pad(a, (0, pad)), torch.nn.functional.pad(b, (0, pad))
    c = torch.empty(a.shape[0], b.shape[0], device=a.device)
    return ops.gemm_f32_raw("nt", a.contiguous(), b.contiguous(), c, math=0)


def prod_fp16x2_nt(a, b):
    """a (M, K), b (N, K): three-term two-way fp16 split product with per-row scales."""
    sa, sb = pow2_scale(a, 1), pow2_scale(b, 1)
    a0, a1 = split16(a, sa)
    b0, b1 = split16(b, sb)
    c = gemm_nt_exact(a0, b1) + gemm_nt_exact(a1, b0)       # the small terms first, as the shipped kernel orders its six
    c = c + gemm_nt_exact(a0, b0)
    return c / sa / sb.t()


def as_nt(mode, a, b):
    """Operands of c = op(a) op(b) rewritten as (M, K), (N, K) for the NT form."""
    if mode == "nt":
        return a, b
    if mode == "nn":
        return a, b.t().contiguous()
    return a.t().contiguous(), b.t().contiguous()


def kernel(mode, a, b, shape, math):
    c = torch.empty(shape, device=a.device)
    return ops.gemm_f32_raw(mode, a, b, c, math=math)


def part1():
    g = torch.Generator(device="cpu").manual_seed(11)
    print("== part 1: normalised error |c - c64| / sum_k |a||b| (max / rms); rule: <= 2 x the exact kernel's own error")
    rows = []
    for case in ("plain", "range", "cancel", "underflow"):
        for mode in ("nt", "nn", "tn"):
            M, N, K = (384, 256, 768)
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
            elif case == "underflow":
                a = a * 2.0 ** -112
            a, b = a.to(DEV).contiguous(), b.to(DEV).contiguous()
            an, bn = as_nt(mode, a, b)
            ref = an.double() @ bn.double().t()
            scale = an.double().abs() @ bn.double().abs().t()
            exact = kernel(mode, a, b, (M, N), 0)
            x3 = kernel(mode, a, b, (M, N), 1)
            f16 = prod_fp16x2_nt(an, bn)
            errs = [((c.double() - ref).abs() / scale) for c in (exact, x3, f16)]
            mx = [float(e.max()) for e in errs]
            rms = [float(e.pow(2).mean().sqrt()) for e in errs]
            ok3 = mx[1] <= 2 * mx[0] and rms[1] <= 2 * rms[0]
            ok16 = mx[2] <= 2 * mx[0] and rms[2] <= 2 * rms[0]
            if case == "underflow":
                floor = float((2.0 ** -126 * (an.double().abs().sum(1).reshape(-1, 1) + bn.double().abs().sum(1).reshape(1, -1)) / scale).max())
                ok3 = mx[1] <= 2 * mx[0] + floor
                ok16 = mx[2] <= 2 * mx[0] + floor
            rows.append((case, mode, mx, rms, ok3, ok16))
            print("%-9s %-2s  exact %.2e / %.2e   bf16x3 %.2e / %.2e (%s)   fp16x2 %.2e / %.2e (%s)" % (
                case, mode, mx[0], rms[0], mx[1], rms[1], "pass" if ok3 else "FAIL", mx[2], rms[2], "pass" if ok16 else "FAIL"), flush=True)
    return rows


class EmuLinearFn(torch.autograd.Function):
    """y = x W^T + b with all three products of the Linear on the emulated fp16 x 2 product."""

    @staticmethod
    def forward(ctx, x2, w, b):
        ctx.save_for_backward(x2, w)
        y = prod_fp16x2_nt(x2, w)
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = prod_fp16x2_nt(dy, w.t().contiguous())                          # contraction over the output features
        dw = prod_fp16x2_nt(dy.t().contiguous(), x2.t().contiguous())        # contraction over the tokens: scales per feature column
        return dx, dw, dy.sum(0)


def emu_linear(x, lin, resid=None, *a, **k):
    y = EmuLinearFn.apply(x.reshape(-1, x.shape[-1]).contiguous(), lin.weight, lin.bias).reshape(*x.shape[:-1], lin.weight.shape[0])
    return y if resid is None else resid + y


def part2():
    from conftest import load_golden, recipe_sd
    from recipe import make_inputs
    from acr_wsss_amd import backbone
    from acr_wsss_amd.DPT.ACR import ACR
    from acr_wsss_amd.train import acr_loss
    fx = load_golden("train_hybrid_448_b1")
    size, batch, ncls, alpha, seed = [int(v) for v in fx["meta"]]
    img, label = make_inputs(batch, size, ncls, seed)
    img, label = img.to(DEV), label.to(DEV)
    model = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False)
    model.load_state_dict(recipe_sd("hybrid"), strict=True)
    model = model.to(DEV)

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)

    def run(tag):
        model.train()
        model.zero_grad(set_to_none=True)
        cls_list, attn_list = model.forward_mirror(img, img.flip(-1))
        loss, terms = acr_loss(cls_list, attn_list, label, size // 16, alpha)
        loss.backward()
        lt = {k: abs(float(terms[k]) - float(fx[k])) / abs(float(fx[k])) for k in ("loss", "cls_align", "aff_align", "cls_loss_1", "cls_loss_2")}
        maps = []
        for i, k in enumerate(("attn1", "attn2")):
            a = attn_list[i].detach().cpu().numpy()
            s0, s1 = [int(v) for v in fx["sub"]]
            maps.append(max(rel(a[:, :, ::s0, ::s1], fx[k + "_sub"]), rel(a[:, :, 0, :], fx[k + "_row0"])))
        params = dict(model.named_parameters())
        grads = {k[5:]: rel(params[k[5:]].grad.cpu().numpy(), v) for k, v in fx.items() if k.startswith("grad:")}
        worst_g = max((v, k) for k, v in grads.items() if "stem.norm.bias" not in k)
        print("%-28s loss terms worst %.2e (tol 5e-6)   maps %.2e (tol 2e-4)   fixture grads worst %.2e [%s] (tol 2e-3); stem.norm.bias %.2e (tol 8e-2)" % (
            tag, max(lt.values()), max(maps), worst_g[0], worst_g[1].split("model.")[-1],
            grads.get("pretrained.model.patch_embed.backbone.stem.norm.bias", float("nan"))), flush=True)
        return max(lt.values()), max(maps), worst_g[0]

    print("== part 2: reference fixture train_hybrid_448_b1 (deviations from the reference's own fp32 CPU run)")
    out = {}
    model.set_math("f32")
    out["f32"] = run("exact fp32 MFMA")
    model.set_math("f32_split")
    out["f32_split"] = run("bf16 x 3, six terms (shipped)")
    # every block Linear on the emulated three-term product; attention core and stem stay on the shipped split products
    keep = (ops.linear_or_hip, backbone.Mlp.fused, ops.ATTN_O_IMAGE)
    ops.linear_or_hip, backbone.Mlp.fused, ops.ATTN_O_IMAGE = emu_linear, False, False
    try:
        out["fp16x2"] = run("fp16 x 2, three terms (emu)")
    finally:
        ops.linear_or_hip, backbone.Mlp.fused, ops.ATTN_O_IMAGE = keep
    return out


if __name__ == "__main__":
    torch.backends.cudnn.deterministic = True
    part1()
    if "--no-model" not in sys.argv:
        part2()
