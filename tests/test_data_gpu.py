"""Input pipeline on the GPU: acr_preprocess_batch through acr_wsss_amd.data against the CPU oracle's
restatement of myTool.py:1158-1199 / :1364-1403 (oracle/data_oracle.py) on the same decoded pixels and the same draws."""
import io
import random

import numpy as np
import pytest
import torch

from acr_wsss_amd import data
from oracle import data_oracle as DO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _images(rng, shapes):
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]


@pytest.mark.parametrize("S,shapes", [
    (448, [(375, 500), (500, 333), (120, 90), (800, 1200), (448, 448), (281, 500), (500, 375), (64, 1000)]),   # VOC-like + extremes
    (128, [(120, 200), (260, 90), (128, 128), (31, 17)]),
    (512, [(480, 640), (640, 427)]),                                                                            # COCO-like
])
def test_train_batch_matches_oracle(S, shapes):
    rng = np.random.default_rng(S)
    imgs = _images(rng, shapes)
    labels = torch.zeros(len(imgs), 20)
    b = data.TrainBatcher(S, device=DEV, seed=5)
    x, y = b(imgs, labels)
    assert x.shape == (len(imgs), 3, S, S) and x.dtype == torch.float32 and x.is_cuda and y.is_cuda
    pr, nr = random.Random(5), np.random.RandomState(5)
    flips = 0
    for i, img in enumerate(imgs):
        g = DO.draw_train_geometry(img.shape[0], img.shape[1], S, pr, nr)
        flips += g["flip"]
        ref = DO.train_image(img, S, g)
        got = x[i].cpu().numpy()
        assert np.array_equal(got == 0, ref == 0) or np.abs(got - ref).max() <= 2e-5     # the zero band is exact
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5, err_msg="image %d %s" % (i, g))
    assert 0 < flips < len(imgs) or len(imgs) < 4
    # bf16 output (the training precision of the bf16 mode): one rounding of the same values
    xb, _ = data.TrainBatcher(S, device=DEV, seed=5, dtype=torch.bfloat16)(imgs, labels)
    assert xb.dtype == torch.bfloat16 and torch.equal(xb, x.bfloat16())


def test_val_batch_and_jpeg_decode():
    """get_data_from_chunk_val on images that went through a real decoder (PIL; the reference uses cv2.imread)."""
    from PIL import Image
    rng = np.random.default_rng(9)
    imgs = []
    for (h, w) in ((375, 500), (333, 500), (200, 150)):
        base = rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8)
        arr = np.asarray(Image.fromarray(base).resize((w, h), Image.BILINEAR))
        buf = io.BytesIO()
        Image.fromarray(arr).save(buf, format="JPEG", quality=90)
        imgs.append(np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")))
    x = data.val_batch(imgs, 384, device=DEV)
    assert x.shape == (3, 3, 384, 384)
    for i, img in enumerate(imgs):
        np.testing.assert_allclose(x[i].cpu().numpy(), DO.val_image(img, 384), rtol=0, atol=2e-5)


def test_pipeline_feeds_the_model_and_is_fast():
    """One chunk through TrainBatcher -> forward_mirror (the contract of train_acr.py:129-138), plus a throughput
    figure for the record: the host packs, ONE copy + ONE launch do the rest."""
    import time
    from acr_wsss_amd.DPT.ACR import ACR
    rng = np.random.default_rng(3)
    imgs = _images(rng, [(375, 500)] * 16)
    labels = torch.zeros(16, 20)
    labels[:, 3] = 1
    b = data.TrainBatcher(448, device=DEV, seed=1)
    x, y = b(imgs, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        x, y = b(imgs, labels)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("input pipeline: %.1f img/s for 16 x 375x500 -> 448^2 (host packing + H2D + kernel)" % (16 / dt))
    model = ACR(num_classes=20, backbone_name="vit_tiny", use_pretrain=False).to(DEV).train()
    cls_list, attn_list = model.forward_mirror(x[:2, :, :224, :224].contiguous(), x[:2, :, :224, :224].flip(-1).contiguous())
    assert cls_list[0].shape == (2, 20) and attn_list[0].shape[-1] == 197
