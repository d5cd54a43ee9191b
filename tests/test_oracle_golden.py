"""Pin the CPU oracle (oracle/acr_oracle.py) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  fp32 CPU vs fp32 CPU on the same torch build: tolerances are tight."""
import os

import numpy as np
import pytest
import torch

from conftest import assert_seeds_exact_or_tie, load_golden, recipe_sd
from recipe import make_inputs, weights_checksum
from oracle import acr_oracle as O


def _leafify(sd):
    return {k: v.clone().requires_grad_(True) for k, v in sd.items()}


def _check_train(fx, sd, cfg, rtol=2e-5, atol=2e-6):
    size, batch, ncls, alpha, seed = [int(v) for v in fx["meta"]]
    assert abs(weights_checksum(sd) - float(fx["weights_checksum"])) < 1e-3 * float(fx["weights_checksum"])
    img, label = make_inputs(batch, size, ncls, seed)
    if fx["img"].size:
        np.testing.assert_array_equal(img.numpy(), fx["img"])
    np.testing.assert_array_equal(label.numpy(), fx["label"])
    sdg = _leafify(sd)
    loss, terms = O.train_step(sdg, cfg, img, label, alpha)
    loss.backward()
    for k in ("loss", "cls_align", "aff_align", "cls_loss_1", "cls_loss_2", "x_cls_1", "x_cls_2", "x_p_cls_1", "x_p_cls_2"):
        np.testing.assert_allclose(terms[k].detach().numpy(), fx[k], rtol=rtol, atol=atol, err_msg=k)
    for k in ("attn1", "attn2"):
        a = terms[k].detach().numpy()
        if k in fx:
            np.testing.assert_allclose(a, fx[k], rtol=rtol, atol=1e-7, err_msg=k)
        else:
            s0, s1 = [int(v) for v in fx["sub"]]
            np.testing.assert_allclose(a[:, :, ::s0, ::s1], fx[k + "_sub"], rtol=rtol, atol=1e-7)
            np.testing.assert_allclose(a[:, :, 0, :], fx[k + "_row0"], rtol=rtol, atol=1e-7)
    ngrad = 0
    for k, v in fx.items():
        if k.startswith("grad:"):
            g = sdg[k[5:]].grad.numpy()
            scale = np.abs(v).max() + 1e-12
            assert np.abs(g - v).max() <= 2e-4 * scale, (k, np.abs(g - v).max(), scale)
            ngrad += 1
    assert ngrad >= 5
    # permutation form == in-place flip form (bit-exact)
    p = size // 16
    c, a = O.acr_align_perm(terms["attn1"].detach(), terms["attn2"].detach(), p)
    assert float(c) == float(terms["cls_align"]) and float(a) == float(terms["aff_align"])
    return sdg, terms


def test_train_hybrid_64(hybrid_sd):
    fx = load_golden("train_hybrid_64_b2")
    sdg, _ = _check_train(fx, hybrid_sd, O.HYBRID_BASE)
    # PolyOptimizer quirk: first step is plain SGD at lr0; check the one stored parameter
    name = "cls_head.bias"
    params, grads = [sdg[name]], [sdg[name].grad]
    lr = O.poly_sgd_step(params, grads, [None], step=0, max_step=100, lr0=0.05, wt_dec=5e-4)
    np.testing.assert_allclose(params[0].detach().numpy(), fx["after_step:" + name], rtol=1e-6, atol=1e-7)
    assert abs(lr - float(fx["lr_after_step"])) < 1e-9


def test_train_hybrid_96(hybrid_sd):
    _check_train(load_golden("train_hybrid_96_b1"), hybrid_sd, O.HYBRID_BASE)


def test_train_tiny_224(tiny_sd):
    _check_train(load_golden("train_tiny_224_b2"), tiny_sd, O.VIT_TINY)


@pytest.mark.parametrize("name", ["infer_hybrid_64", "infer_hybrid_96"])
def test_infer(hybrid_sd, name):
    fx = load_golden(name)
    size, W, H, seed = [int(v) for v in fx["meta"]]
    img = torch.from_numpy(fx["img"])
    label = torch.from_numpy(fx["label"])
    keys = sorted({k.split(":")[1] for k in fx if k.startswith("getam_rows:")})
    assert len(keys) == 7
    for key in keys:
        func, s, a = key.rsplit("_", 2)
        cam_dict, patch_dict, rows = O.infer_image(hybrid_sd, O.HYBRID_BASE, img, label, (W, H),
                                                   start_layer=int(s[1:]), func=func, aff=bool(int(a[1:])))
        ref_rows = fx["getam_rows:" + key]
        assert np.abs(rows - ref_rows).max() <= 1e-4 * np.abs(ref_rows).max() + 1e-12, key
        for c, v in cam_dict.items():
            np.testing.assert_allclose(v, fx["cam:%s:%d" % (key, c)], rtol=0, atol=2e-4, err_msg=key)
        for t in (0.2, 0.4):
            got = O.seeds_from_cam_dict(cam_dict, t)
            ref = fx["seed:%s:%.1f" % (key, t)]
            assert (got != ref).mean() <= 1e-3, (key, t, (got != ref).mean())
        if key == "grad_s10_a1":
            for c, v in patch_dict.items():
                np.testing.assert_allclose(v, fx["patch_cam:%d" % c], rtol=0, atol=1e-5)


def test_train_coco_512():
    """BASELINE configs[4]: 80 classes, 512^2 (T = 1025, p = 32), train_acr_coco.py:91,134-165."""
    _check_train(load_golden("train_coco_512_b1"), recipe_sd("coco"), O.HYBRID_BASE)


def _big_cases():
    names = ["infer_ms_hybrid_96", "infer_hybrid_384"]
    if os.environ.get("ACR_SLOW_ORACLE") == "1":          # ~4 min of CPU at T up to 2305: opt-in (the GPU test uses the fixture)
        names.append("infer_ms_hybrid_384")
    return names


@pytest.mark.parametrize("name", _big_cases())
def test_infer_real_geometry_and_multi_scale(hybrid_sd, name):
    """infer_cam.py:141-215 at the shipped inference size (384^2 -> T = 577) and over BASELINE configs[3]'s scale set
    {0.5, 1, 1.5, 2}; seeds bit-exact (or a proven fp tie of the reference's own argmax)."""
    fx = load_golden(name)
    size, W, H, seed = [int(v) for v in fx["meta"]]
    scales = tuple(float(s) for s in fx["scales"])
    img, _ = make_inputs(1, size, 20, seed)
    label = torch.from_numpy(fx["label"])
    keys = sorted({k.split(":")[1] for k in fx if k.startswith("seed:")})
    for key in keys:
        func, s, a = key.rsplit("_", 2)
        cam_dict, patch_dict, rows = O.infer_image(hybrid_sd, O.HYBRID_BASE, img, label, (W, H), start_layer=int(s[1:]),
                                                   func=func, aff=bool(int(a[1:])), scales=scales)
        for i, r in enumerate(rows):
            ref = fx["getam_row:%s:%d" % (key, i)]
            assert r.shape == ref.shape and np.abs(r - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-12, (key, i)
        ref_cams = {c: fx["cam:%s:%d" % (key, c)] for c in cam_dict}
        err = max(float(np.abs(cam_dict[c] - ref_cams[c]).max()) for c in cam_dict)
        assert err <= 2e-4, (key, err)
        for t in (0.2, 0.4):
            assert_seeds_exact_or_tie(O.seeds_from_cam_dict(cam_dict, t), fx["seed:%s:%.1f" % (key, t)], ref_cams, t, err, key)
    for c in [int(k.split(":")[1]) for k in fx if k.startswith("patch_cam:")]:
        np.testing.assert_allclose(patch_dict[c], fx["patch_cam:%d" % c], rtol=0, atol=1e-5)


def test_getam_deit_distilled():
    """The oracle on a non-hybrid backbone with two leading tokens (DPT/ACR.py:155-160 'deit_distilled') against the
    reference's own outputs: forward_cam and getam's `[:, 0, 2:]` branch (DPT/ACR.py:210-211)."""
    fx = load_golden("getam_distil_96")
    sd = {k: v.clone().requires_grad_(True) for k, v in recipe_sd("distil").items()}
    img = torch.from_numpy(fx["img"])
    cls_pred, x_patch_cls, attn, patch_cam, maps = O.forward_cam(img, sd, O.DEIT_DISTILLED)
    for got, key in ((cls_pred, "cls_pred"), (x_patch_cls, "x_patch_cls"), (attn, "attn"), (patch_cam, "patch_cam")):
        ref = fx[key]
        assert np.abs(got.detach().numpy() - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-7, key
    for c in fx["classes"]:
        for P in maps:
            P.grad = None
        cls_pred[0, int(c)].backward(retain_graph=True)
        for func in ("grad", "cam_grad", "grad_s", "cam_grad_s"):
            for start_layer in (0, 10):
                cam = O.getam([P.detach() for P in maps], [P.grad for P in maps], 0, start_layer, func, distilled=True)
                ref = fx["getam:%s_s%d:%d" % (func, start_layer, int(c))]
                assert cam.shape == ref.shape and np.abs(cam.numpy() - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-9


def test_iou_counters():
    rng = np.random.default_rng(0)
    gt = rng.integers(0, 21, (40, 50)).astype(np.uint8)
    gt[rng.random((40, 50)) < 0.1] = 255
    pred = gt.copy()
    flip = rng.random((40, 50)) < 0.3
    pred[flip] = rng.integers(0, 21, flip.sum())
    pred[gt == 255] = 0
    TP, P, T = O.iou_counts(pred, gt)
    assert (TP <= P).all() and (TP <= T).all() and T.sum() == (gt < 255).sum()
    assert 0 < O.miou(TP, P, T) < 100
