"""Dense-CRF stage on the GPU, through the C ABI (acr_lattice_*, acr_crf_*): the HIP permutohedral lattice
against fixtures generated from the reference's own C++ lattice and against the oracle -- tables equal up to the numbering of
lattice points, weights and filter outputs BIT-EXACT (the sums run in the CPU code's order) -- and the mean-field CRF against
the oracle within a stated tolerance; properties at full VOC size."""
import os
import time

import numpy as np
import pytest
import torch

from oracle import crf_oracle as C

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


def _same_partition(a, b):
    """two labelings of the same items describe the same partition (a bijection between the label sets)"""
    a, b = a.reshape(-1).astype(np.int64), b.reshape(-1).astype(np.int64)
    pairs = np.unique(np.stack([a, b], axis=1), axis=0)
    return len(pairs) == len(np.unique(a)) == len(np.unique(b))


def _lattice(img, name):
    from acr_wsss_amd.crf import PermutohedralLattice
    h, w = img.shape[:2]
    return PermutohedralLattice(h, w, 80, rgb=img, srgb=13, device=DEV) if name == "bil" else PermutohedralLattice(h, w, 3, device=DEV)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_lattice_matches_reference_fixtures(tag):
    g = np.load(os.path.join(GOLD, "crf_lattice_%s.npz" % tag))
    img, vals = g["img"], g["vals"]
    x = torch.from_numpy(np.ascontiguousarray(vals.T)).to(DEV)
    for name in ("bil", "spa"):
        lat = _lattice(img, name)
        off, wts, keys = lat.tables()
        assert lat.n_points == int(g[name + "_M"])
        assert _same_partition(off, g[name + "_offsets"])                   # index work: identical structure, other numbering
        assert np.array_equal(wts, g[name + "_weights"])                    # fp32 bit for bit
        assert len(np.unique(keys, axis=0)) == lat.n_points
        out = lat.filter(x).cpu().numpy().T
        assert np.array_equal(out, g[name + "_filter"]), np.abs(out - g[name + "_filter"]).max()


def test_lattice_matches_oracle_larger_and_is_deterministic():
    rng = np.random.default_rng(11)
    h, w, k = 97, 131, 5                                                     # n % 4 = 3
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 2) % 256, (yy * 3) % 256, (xx + yy) % 256], -1).astype(np.uint8)
    img[30:70, 40:100] = (10, 200, 90)
    img[:20] = rng.integers(0, 256, (20, w, 3))
    vals = rng.standard_normal((h * w, k)).astype(np.float32)
    x = torch.from_numpy(np.ascontiguousarray(vals.T)).to(DEV)
    for name, feat in (("bil", C.bilateral_features(img, 80, 13)), ("spa", C.spatial_features(h, w, 3))):
        ref = C.lattice_compute(C.lattice_init(feat), vals)
        lat = _lattice(img, name)
        out = lat.filter(x).cpu().numpy().T
        assert np.array_equal(out, ref), np.abs(out - ref).max()
        again = _lattice(img, name).filter(x).cpu().numpy().T
        assert np.array_equal(out, again)                                    # no float atomics anywhere
        norm = torch.rand(h * w, device=DEV) + 0.5
        fused = lat.filter(x, pre=norm, post=norm, scale=3.0).cpu().numpy().T
        nn = norm.cpu().numpy()
        ref2 = np.float32(3.0) * (C.lattice_compute(C.lattice_init(feat), (vals * nn[:, None]).astype(np.float32)) * nn[:, None])
        assert np.array_equal(fused, ref2.astype(np.float32))


def _scene(h, w, k, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 3) % 256, (yy * 2) % 256, (xx * yy) % 256], -1).astype(np.uint8)
    img[h // 4: 3 * h // 4, w // 3: 2 * w // 3] = (230, 40, 40)
    probs = rng.random((k, h, w)).astype(np.float32) * 0.3
    probs[1, h // 4: 3 * h // 4, w // 3: 2 * w // 3] += 0.6
    probs[0] += 0.3
    return img, probs


def _tolerance(img, probs, k, ref, floor=1e-4):
    """|dQ| allowed between the HIP CRF and the oracle.  The lattice is bit-exact; what differs is logf / expf vs numpy's log /
    exp, a last-bit difference at the input of 10 mean-field iterations with Potts weights 3 and 10.  How far one ulp travels is
    a property of the instance, so it is MEASURED: the oracle is run a second time with its unary's log evaluated in float32
    instead of float64 (<= 1 ulp apart, exactly the freedom pydensecrf's np.log has).  The HIP path has eleven such sources (logf
    once, expf in each of the 10 normalisations, each <= 2 ulp), the probe exercises one: the bound is 16 x the probe, floored at
    1e-4 on smooth scenes.  On NOISE images (no flat regions, scores clipped at both ends) single pixels sit on a knife edge
    between two labels and the worst pixel moves by up to 3e-4 whatever the probe says (seen: 2.9e-5 ... 3.2e-4 with probes of
    6e-6 ... 7.6e-5): there the floor is 2e-3 on the worst pixel and the MEAN deviation is held to 1e-5 instead."""
    sens = np.abs(C.crf_inference(img, probs, labels=k, log_dtype=np.float32) - ref).max()
    return max(floor, 16.0 * float(sens)), float(sens)


def test_crf_inference_matches_oracle():
    """tool/imutils.py:345-362 end to end, smooth scenes and a noise image (the hard case: no flat regions, unaries clipped at
    1e-5).  Tolerance: see _tolerance; labels equal wherever the oracle's top-2 margin exceeds twice that bound."""
    from acr_wsss_amd.crf import crf_inference, crf_with_alpha
    for (h, w, k, seed) in ((48, 64, 3, 0), (61, 47, 5, 1), (40, 52, 3, -1)):
        img, probs = _scene(h, w, k, abs(seed))
        if seed < 0:
            rng = np.random.default_rng(9)
            img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
            probs = (rng.random((k, h, w)) ** 8).astype(np.float32)           # many entries below the 1e-5 clip
        ref = C.crf_inference(img, probs, labels=k)
        got = crf_inference(img, probs, labels=k, device=DEV)
        assert got.shape == (k, h, w) and got.dtype == np.float32
        err = np.abs(got - ref).max()
        tol, sens = _tolerance(img, probs, k, ref, floor=2e-3 if seed < 0 else 1e-4)
        assert np.abs(got - ref).mean() <= 1e-5
        print("\ncrf %dx%dx%d: |dQ| max %.2e (1-ulp sensitivity of the instance %.2e, tolerance %.2e)" % (h, w, k, err, sens, tol))
        assert err <= tol, (err, tol)
        top = np.sort(ref, axis=0)
        decided = (top[-1] - top[-2]) > 2 * tol
        assert decided.mean() > 0.99
        assert np.array_equal(got.argmax(0)[decided], ref.argmax(0)[decided])
        q0 = crf_inference(img, probs, t=0, labels=k, device=DEV)           # no iteration: softmax(-unary) = clipped, normalised probs
        pc = np.clip(probs, 1e-5, 1.0)
        assert np.allclose(q0, pc / pc.sum(0, keepdims=True), atol=1e-5)
    cams = {3: probs[1], 7: probs[2]}
    a = crf_with_alpha(cams, 4, img, device=DEV)
    b = C.crf_with_alpha(cams, 4, img)
    assert sorted(a) == sorted(b) == [0, 4, 8]
    for c in a:
        assert np.abs(a[c] - b[c]).max() <= 1e-4               # smooth scene (see the measured bound above)


def test_crf_full_voc_size_properties():
    """375 x 500, 21 labels (the size infer_cam.py:218-225 runs at): normalisation, run-to-run bit-identity, filter linearity and
    positivity; prints the time per image."""
    from acr_wsss_amd.crf import PermutohedralLattice, crf_inference
    h, w, k = 375, 500, 21
    img, probs = _scene(h, w, k, 3)
    crf_inference(img, probs[:2], labels=2, device=DEV)                      # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    q = crf_inference(img, probs, labels=k, device=DEV)
    dt = time.time() - t0
    print("\ncrf_inference 375x500, 21 labels, 10 iterations: %.1f ms" % (1e3 * dt))
    assert np.isfinite(q).all() and np.allclose(q.sum(0), 1.0, atol=1e-5)
    assert np.array_equal(q, crf_inference(img, probs, labels=k, device=DEV))
    inside = q[1, h // 4 + 8: 3 * h // 4 - 8, w // 3 + 8: 2 * w // 3 - 8]
    assert inside.mean() > 0.9                                               # the flat red box ends up as class 1
    lat = PermutohedralLattice(h, w, 80, rgb=img, srgb=13, device=DEV)
    x = torch.rand(2, h * w, device=DEV)
    y = torch.rand(2, h * w, device=DEV)
    lin = lat.filter((2 * x - 3 * y).contiguous())
    ref = 2 * lat.filter(x) - 3 * lat.filter(y)
    assert (lin - ref).abs().max() <= 1e-3 * ref.abs().max()
    assert (lat.filter(torch.ones(1, h * w, device=DEV)) > 0).all()


def test_errors_are_loud():
    from acr_wsss_amd.crf import PermutohedralLattice
    from acr_wsss_amd._lib import AcrHipError
    with pytest.raises(ValueError):
        PermutohedralLattice(8, 8, 3.0, rgb=np.zeros((8, 9, 3), np.uint8), device=DEV)
    with pytest.raises(AcrHipError):                                         # x / sxy far beyond the 12-bit key range
        PermutohedralLattice(64, 4096, 0.01, device=DEV)
    lat = PermutohedralLattice(8, 8, 3.0, device=DEV)
    with pytest.raises(ValueError):
        lat.filter(torch.zeros(2, 63, device=DEV))


def test_infer_cam_list_writes_crf_outputs(tmp_path):
    """infer_cam.py:218-225: with out_crf set, <out_crf>_<alpha>/<name>.npy holds {0: bg, class + 1: ...} for both alphas."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from __graft_entry__ import _recipe_model
    from recipe import make_inputs
    from acr_wsss_amd.infer_cam import infer_cam_list
    model, _ = _recipe_model(torch.device(DEV))
    img, _ = make_inputs(1, 64, 20, 2)
    label = torch.zeros(1, 20)
    label[0, [2, 9]] = 1
    orig = np.random.default_rng(0).integers(0, 256, (40, 52, 3)).astype(np.uint8)
    items = [("im0", img, label, (40, 52), orig)]
    res = infer_cam_list(model, items, out_cam=str(tmp_path / "cam"), out_crf=str(tmp_path / "crf"), low_alpha=1, high_alpha=12)
    for alpha in (1, 12):
        d = np.load(str(tmp_path / ("crf_%d" % alpha) / "im0.npy"), allow_pickle=True).item()
        assert sorted(d) == [0, 3, 10] and d[0].shape == (40, 52) and d[0].dtype == np.float32
        assert np.allclose(sum(d.values()), 1.0, atol=1e-5)
        ref = C.crf_with_alpha(res["im0"], alpha, orig)
        cams = np.stack([res["im0"][c] for c in res["im0"]])
        scores = np.concatenate((np.power(1 - cams.max(0, keepdims=True), alpha), cams), 0)
        tol, _ = _tolerance(orig, scores, scores.shape[0], np.stack([ref[c] for c in sorted(ref)]), floor=2e-3)   # noise image
        for c in d:
            assert np.abs(d[c] - ref[c]).max() <= tol, (alpha, c, np.abs(d[c] - ref[c]).max(), tol)
            assert np.abs(d[c] - ref[c]).mean() <= 1e-5
    with pytest.raises(ValueError):
        infer_cam_list(model, [items[0][:4]], out_crf=str(tmp_path / "x"))


@pytest.mark.parametrize("hw", [(1, 1), (2, 3), (1, 9), (7, 1)])
def test_lattice_tiny_images(hw):
    """degenerate sizes (fewer pixels than the reference's SSE block of 4, single rows / columns) against the oracle"""
    h, w = hw
    rng = np.random.default_rng(h * 10 + w)
    img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    vals = rng.standard_normal((h * w, 2)).astype(np.float32)
    x = torch.from_numpy(np.ascontiguousarray(vals.T)).to(DEV)
    for name, feat in (("bil", C.bilateral_features(img, 80, 13)), ("spa", C.spatial_features(h, w, 3))):
        ref_lat = C.lattice_init(feat)
        lat = _lattice(img, name)
        assert lat.n_points == ref_lat["M"]
        assert np.array_equal(lat.filter(x).cpu().numpy().T, C.lattice_compute(ref_lat, vals))


def test_lattice_large_image_capacity():
    """768 x 1024 (4.7 M pixel-vertex pairs): builds without key overflow, filters deterministically, stays normalisable"""
    from acr_wsss_amd.crf import PermutohedralLattice
    h, w = 768, 1024
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    lat = PermutohedralLattice(h, w, 80, rgb=img, srgb=13, device=DEV)
    assert 0 < lat.n_points <= 6 * (h * w + 1)
    x = torch.rand(2, h * w, device=DEV)
    a = lat.filter(x)
    assert torch.isfinite(a).all() and torch.equal(a, lat.filter(x))
    assert (lat.filter(torch.ones(1, h * w, device=DEV)) > 0).all()
