"""Dense-CRF stage, CPU side: the numpy restatement of the permutohedral lattice against fixtures generated
from the reference's own C++ (tests/golden/crf_lattice_*.npz, make_crf_golden.py) -- bit for bit -- and, where
oracle/_ref/libpermuto_ref.so was built, against that library live on fresh inputs; sanity of the mean-field restatement (its
parity is unpinned: pydensecrf is not available); the product has no CPU path."""
import os

import numpy as np
import pytest

from oracle import crf_oracle as C

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _features(g, name):
    img = g["img"]
    h, w = img.shape[:2]
    return C.bilateral_features(img, 80, 13) if name == "bil" else C.spatial_features(h, w, 3)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_oracle_lattice_matches_reference_fixtures(tag):
    g = np.load(os.path.join(GOLD, "crf_lattice_%s.npz" % tag))
    for name in ("bil", "spa"):
        lat = C.lattice_init(_features(g, name))
        assert lat["M"] == int(g[name + "_M"])
        assert np.array_equal(lat["offsets"], g[name + "_offsets"])            # same numbering: ids in first-insertion order
        assert np.array_equal(lat["weights"], g[name + "_weights"])            # float32, bit for bit
        assert np.array_equal(C.lattice_compute(lat, g["vals"]), g[name + "_filter"])
    h, w = g["img"].shape[:2]
    k = g["vals"].shape[1]
    lat = C.lattice_init(_features(g, "bil"))
    mine = C.lattice_compute(lat, g["vals"]).T.reshape(k, h, w)
    assert np.array_equal(mine, g["entry_bilateralfilter"])                    # bilateralfilter.cpp:22-41, plane by plane


def test_oracle_lattice_matches_compiled_reference_live():
    lib = C.load_ref()
    if lib is None:
        pytest.skip("oracle/_ref not built here (needs /root/reference: make -C oracle)")
    rng = np.random.default_rng(7)
    for (h, w, k, sxy, srgb) in ((30, 45, 3, 80, 13), (25, 25, 1, 8, 5), (50, 20, 6, 40, 3)):
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        img[h // 2:] = img[h // 2, 0]                                          # flat half: many pixels per lattice point
        vals = rng.standard_normal((h * w, k)).astype(np.float32)
        for feat in (C.bilateral_features(img, sxy, srgb), C.spatial_features(h, w, 3), C.spatial_features(h, w, 0.7)):
            lat = C.lattice_init(feat)
            off, wts, m = C.ref_lattice_tables(lib, feat)
            out, _ = C.ref_lattice_filter(lib, feat, vals)
            assert lat["M"] == m and np.array_equal(lat["offsets"], off) and np.array_equal(lat["weights"], wts)
            assert np.array_equal(C.lattice_compute(lat, vals), out)


def test_lattice_properties():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (20, 28, 3)).astype(np.uint8)
    lat = C.lattice_init(C.bilateral_features(img, 80, 13))
    assert np.allclose(lat["weights"].sum(axis=1), 1.0, atol=1e-5)             # barycentric coordinates
    assert lat["weights"].min() >= -1e-6
    x = rng.random((20 * 28, 2)).astype(np.float32)
    y = rng.random((20 * 28, 2)).astype(np.float32)
    lin = C.lattice_compute(lat, 2 * x - 3 * y)
    assert np.allclose(lin, 2 * C.lattice_compute(lat, x) - 3 * C.lattice_compute(lat, y), atol=2e-4)
    assert (C.lattice_compute(lat, np.ones((20 * 28, 1), np.float32)) > 0).all()


def test_mean_field_restatement_sanity():
    rng = np.random.default_rng(5)
    h, w, k = 24, 30, 3
    img = np.zeros((h, w, 3), np.uint8)
    img[:, w // 2:] = 220                                                       # two flat halves
    probs = np.full((k, h, w), 0.2, np.float32)
    probs[1, :, : w // 2] = 0.6
    probs[2, :, w // 2:] = 0.6
    probs += rng.random((k, h, w)).astype(np.float32) * 0.05
    q = C.crf_inference(img, probs, labels=k)
    assert q.shape == (k, h, w) and q.dtype == np.float32
    assert np.allclose(q.sum(axis=0), 1.0, atol=1e-5)
    lab = q.argmax(axis=0)
    assert (lab[:, : w // 2 - 2] == 1).all() and (lab[:, w // 2 + 2:] == 2).all()      # the CRF sharpens towards the halves
    assert q[1, :, : w // 2 - 2].min() > 0.9
    q0 = C.crf_inference(img, probs, t=0, labels=k)                              # no iteration: softmax of -unary = normalised probs
    assert np.allclose(q0, probs / probs.sum(axis=0, keepdims=True), atol=1e-5)
    cams = {4: probs[1], 9: probs[2]}
    out = C.crf_with_alpha(cams, 4, img)
    assert sorted(out) == [0, 5, 10] and out[5].shape == (h, w)


def test_product_has_no_cpu_path():
    from acr_wsss_amd import crf
    from acr_wsss_amd._lib import AcrHipError
    with pytest.raises(AcrHipError):
        crf.crf_inference(np.zeros((4, 4, 3), np.uint8), np.full((2, 4, 4), 0.5, np.float32), labels=2, device="cpu")
    with pytest.raises(AcrHipError):
        crf.PermutohedralLattice(4, 4, 3.0, device="cpu")
