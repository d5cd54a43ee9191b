"""Input pipeline, CPU side: the oracle's restatement of myTool.py:923-1008,1158-1199,1364-1403 against
independent formulations, the host geometry of acr_wsss_amd/data.py against the oracle (same draws in the same order from
the same seeds), and the no-CPU-fallback rule."""
import random

import numpy as np
import pytest
import torch

from acr_wsss_amd import data
from oracle import data_oracle as DO


def test_cv2_linear_rule_matches_torch_bilinear():
    """OpenCV's float INTER_LINEAR rule == F.interpolate(bilinear, align_corners=False) (no antialias), up- and
    down-scaling, odd sizes; plus hand-checked border behaviour."""
    rng = np.random.default_rng(0)
    for (h, w, nh, nw) in ((50, 70, 64, 64), (120, 200, 77, 129), (33, 17, 100, 9), (5, 4, 5, 4), (1, 7, 3, 20)):
        img = rng.integers(0, 256, (h, w, 3)).astype(np.float64)
        got = DO.cv2_resize_linear(img, nw, nh)
        ref = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(nh, nw), mode="bilinear",
                                              align_corners=False)[0].permute(1, 2, 0).numpy()
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9)
    one = np.arange(4, dtype=np.float64).reshape(1, 4, 1).repeat(3, axis=2)          # [0, 1, 2, 3] upscaled x2
    up = DO.cv2_resize_linear(one, 8, 1)[0, :, 0]
    np.testing.assert_allclose(up, [0, 0.25, 0.75, 1.25, 1.75, 2.25, 2.75, 3.0])


def test_geometry_restatement():
    assert DO.resize_long_shape(300, 500, 400) == (400, 240)          # (width, height) as handed to cv2.resize
    assert DO.resize_long_shape(500, 300, 400) == (240, 400)
    assert data.resize_long_target(300, 500, 400) == (240, 400) and data.resize_long_target(500, 300, 400) == (400, 240)
    r = random.Random(0)
    for (h, w) in ((600, 500), (300, 500), (200, 100), (448, 448)):
        for _ in range(20):
            b = DO.random_crop_boxes(h, w, 448, r)
            assert 0 <= b["cont_top"] and b["cont_top"] + b["ch"] <= 448 and 0 <= b["cont_left"] and b["cont_left"] + b["cw"] <= 448
            assert 0 <= b["img_top"] and b["img_top"] + b["ch"] <= h and 0 <= b["img_left"] and b["img_left"] + b["cw"] <= w
            assert b["ch"] == min(h, 448) and b["cw"] == min(w, 448)


def test_host_draws_follow_the_reference_order():
    """TrainBatcher.draw consumes random.Random / np.random.RandomState exactly like the oracle's restatement of
    get_data_from_chunk_v2 -> identical geometry from identical seeds, image after image."""
    b = data.TrainBatcher(448, device="cpu", seed=11)
    pr, nr = random.Random(11), np.random.RandomState(11)
    for (h, w) in ((375, 500), (500, 333), (120, 90), (800, 1200), (448, 448)):
        rec = b.draw(h, w)
        g = DO.draw_train_geometry(h, w, 448, pr, nr)
        assert rec[3:12] == (g["rh"], g["rw"], g["flip"], g["cont_top"], g["cont_left"], g["img_top"], g["img_left"], g["ch"], g["cw"])


def test_oracle_train_image_contract():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (120, 200, 3), dtype=np.uint8)
    g = DO.draw_train_geometry(120, 200, 128, random.Random(3), np.random.RandomState(3))
    x = DO.train_image(img, 128, g)
    assert x.shape == (3, 128, 128) and x.dtype == np.float32
    assert 115 <= max(g["rh"], g["rw"]) <= 146 and (x == 0).any()          # short side leaves a zero band
    lo = min((0 - m) / s for m, s in zip(DO.MEAN, DO.STD))
    hi = max((1 - m) / s for m, s in zip(DO.MEAN, DO.STD))
    assert x.min() >= lo - 1e-4 and x.max() <= hi + 1e-4
    v = DO.val_image(img, 64)
    ref = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1).double()[None], size=(64, 64), mode="bilinear",
                                          align_corners=False)[0]
    m = torch.tensor(DO.MEAN).view(3, 1, 1)
    s = torch.tensor(DO.STD).view(3, 1, 1)
    np.testing.assert_allclose(v, ((ref / 255 - m) / s).float().numpy(), atol=1e-6)


def test_record_layout_matches_the_c_struct():
    """acr_pre_image has an int64 first member: the C struct is padded to 56 bytes, and so must the numpy record be
    (a 52-byte record made every image after the first read garbage geometry)."""
    import re, os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "acr_hip.h")).read()
    body = re.search(r"typedef struct acr_pre_image \{(.*?)\} acr_pre_image;", hdr, re.S).group(1)
    n64 = len(re.findall(r"int64_t\s+\w+", body))
    n32 = sum(len(m.split(",")) for m in re.findall(r"int32_t\s+([\w,\s]+);", body))
    assert (n64, n32) == (1, 11)
    assert data.PRE_IMAGE.itemsize == (8 * n64 + 4 * n32 + 7) // 8 * 8 == 56
    assert [data.PRE_IMAGE.fields[k][1] for k in ("offset", "h", "flip", "cw")] == [0, 8, 24, 48]


def test_no_cpu_path():
    from acr_wsss_amd._lib import AcrHipError
    with pytest.raises(AcrHipError):
        data.val_batch([np.zeros((8, 8, 3), np.uint8)], 16, device="cpu")


GOLDEN_CHUNKS = ["train_a", "train_b", "train_c", "val_a", "val_b"]


@pytest.mark.parametrize("name", GOLDEN_CHUNKS)
def test_oracle_equals_the_reference_chunk_functions(name):
    """fixtures written by the reference's OWN get_data_from_chunk_v2 / get_data_from_chunk_val (myTool.py:1158-1199,
    :1364-1403, imported unmodified by tests/golden/make_data_golden.py with its real RandomResizeLong / flip / RandomCrop,
    normalisation and chunk assembly; Python's random and np.random seeded).  The oracle, fed the same decoded RGB arrays and
    generators seeded alike, must reproduce the reference's float32 batch EXACTLY: same draws in the same order, same target
    shapes, same crop placement, same float64 arithmetic.  What this does not pin is the resize rule itself: the generator's
    cv2.resize stand-in is the oracle's cv2_resize_linear (cv2 is not installed) -- the one unpinned step of this row."""
    import os
    from conftest import GOLDEN
    fx = dict(np.load(os.path.join(GOLDEN, "data_chunk_%s.npz" % name)))
    crop, seed = int(fx["crop"]), int(fx["seed"])
    decoded = [fx["rgb_%d" % i] for i in range(fx["images"].shape[0])]
    if name.startswith("train"):
        got, geoms = DO.get_data_from_chunk_v2(decoded, crop, random.Random(seed), np.random.RandomState(seed))
        assert any(g["flip"] for g in geoms) or len(geoms) == 1          # both flip branches occur in the multi-image chunks
    else:
        got = DO.get_data_from_chunk_val(decoded, crop, np.random.RandomState(seed))
    assert got.dtype == np.float32 and got.shape == fx["images"].shape
    np.testing.assert_array_equal(got, fx["images"])
    # the reference's de-normalised uint8 copy (myTool.py:1186-1191) follows from the batch: the normalisation constants round-trip
    ori = np.zeros_like(got, dtype=np.float32)
    for c in range(3):
        ori[:, c] = (got[:, c] * np.float32(DO.STD[c]) + np.float32(DO.MEAN[c])) * 255.0
    assert np.abs(ori.astype(np.uint8).astype(np.int32) - fx["ori_images"].astype(np.int32)).max() <= 1
