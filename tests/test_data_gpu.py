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
    nr.uniform(0.7, 1.3)                                    # the per-chunk `scale` draw of myTool.py:1161
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


def _write_jpegs(tmp_path, n, rng):
    from PIL import Image
    names, labels = [], {}
    for i in range(n):
        h, w = [(375, 500), (500, 333), (333, 500), (281, 500)][i % 4]
        base = rng.integers(0, 256, (h // 6 + 1, w // 6 + 1, 3), dtype=np.uint8)
        arr = np.asarray(Image.fromarray(base).resize((w, h), Image.BICUBIC))
        name = "2007_%06d" % i
        Image.fromarray(arr).save(tmp_path / (name + ".jpg"), format="JPEG", quality=92)
        names.append(name)
        lab = np.zeros(20, np.float32)
        lab[i % 20] = 1.0
        labels[name] = lab
    return names, labels


def test_chunk_loader_reads_files_like_the_reference(tmp_path):
    """The reference's contract end to end: names in, (images, ori_images, labels, name_list) out, from real JPEG bytes on
    disk (myTool.py:1158-1199, :1364-1403) -- against the oracle's chunk functions on the same decoded pixels with equally
    seeded generators (including the unused per-chunk `scale` draw, :1161), for synchronous calls and for the prefetching
    iterator; plus the decoded-JPEG -> tensor rate for the record (>= 400 img/s per process)."""
    import time
    rng = np.random.default_rng(21)
    names, labels = _write_jpegs(tmp_path, 32, rng)
    np.save(tmp_path / "cls_labels.npy", labels)
    S = 448
    chunks = list(data.chunker(names, 8))
    assert data.read_file.__doc__ and [len(c) for c in chunks] == [8, 8, 8, 8]
    loader = data.ChunkLoader(str(tmp_path), str(tmp_path / "cls_labels.npy"), S, device=DEV, seed=11, workers=8, with_ori=True)
    pr, nr = random.Random(11), np.random.RandomState(11)
    decoded = {n: data.decode_rgb(str(tmp_path / (n + ".jpg"))) for n in names}
    got = [loader.get_data_from_chunk_v2(chunks[0]), loader.get_data_from_chunk_v2(chunks[1])]
    for chunk, (images, ori, lab, name_list) in zip(chunks[:2], got):
        ref, geoms = DO.get_data_from_chunk_v2([decoded[n] for n in chunk], S, pr, nr)
        assert images.shape == (8, 3, S, S) and images.is_cuda and name_list == list(chunk)
        np.testing.assert_allclose(images.cpu().numpy(), ref, rtol=0, atol=2e-5)
        assert torch.equal(lab.cpu(), torch.from_numpy(np.stack([labels[n] for n in chunk])))
        # ori_images (:1186-1190): de-normalised crop as uint8; +-1 where fp32 and float64 straddle an integer
        ref_ori = ((ref * np.array(DO.STD, np.float32).reshape(1, 3, 1, 1) + np.array(DO.MEAN, np.float32).reshape(1, 3, 1, 1)) * 255.0)
        assert ori.dtype == np.uint8 and ori.shape == (8, 3, S, S)
        assert np.abs(ori.astype(np.int32) - ref_ori.astype(np.uint8).astype(np.int32)).max() <= 1
    # validation contract
    images, _, lab, _ = loader.get_data_from_chunk_val(chunks[2])
    ref = DO.get_data_from_chunk_val([decoded[n] for n in chunks[2]], S, nr)
    np.testing.assert_allclose(images.cpu().numpy(), ref, rtol=0, atol=2e-5)
    # the prefetching iterator yields exactly the synchronous sequence for the same seed
    a = data.ChunkLoader(str(tmp_path), labels, S, device=DEV, seed=3, workers=8)
    b = data.ChunkLoader(str(tmp_path), labels, S, device=DEV, seed=3, workers=2)
    seq = [a.get_data_from_chunk_v2(c) for c in chunks]
    for (xi, _, yi, ni), (xs, _, ys, ns) in zip(b.iterate(chunks, train=True), seq):
        assert ni == ns and torch.equal(xi, xs) and torch.equal(yi, ys)
    # throughput: 16-image chunks, decode of the next chunks overlapped
    big = list(data.chunker(names * 4, 16))
    t = data.ChunkLoader(str(tmp_path), labels, S, device=DEV, seed=0, workers=16)
    for _ in t.iterate(big[:2]):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for images, _, _, _ in t.iterate(big):
        n += images.shape[0]
    torch.cuda.synchronize()
    rate = n / (time.perf_counter() - t0)
    print("ChunkLoader: %.0f img/s JPEG file -> (16,3,448,448) fp32 tensor on the GPU (%d decode threads)" % (rate, 16))
    assert rate > 100                                        # sanity floor; the measured figure goes to DESIGN.md
    for l in (loader, a, b, t):
        l.close()


@pytest.mark.parametrize("name", ["train_a", "train_b", "train_c", "val_a", "val_b"])
def test_hip_batch_matches_the_reference_chunk_functions(name):
    """The HIP input pipeline against fixtures written by the REFERENCE's own get_data_from_chunk_v2 / _val (myTool.py:1158-1199,
    :1364-1403; tests/golden/make_data_golden.py -- real RandomResizeLong / flip / RandomCrop / normalisation / chunk assembly,
    seeded generators; only cv2.resize is a stand-in, the one unpinned step): same decoded RGB arrays, the batcher seeded like
    the reference run -> the reference's float32 batch within the pipeline's fp32 tolerance, zero bands exact."""
    import os
    from conftest import GOLDEN
    fx = dict(np.load(os.path.join(GOLDEN, "data_chunk_%s.npz" % name)))
    crop, seed = int(fx["crop"]), int(fx["seed"])
    decoded = [fx["rgb_%d" % i] for i in range(fx["images"].shape[0])]
    if name.startswith("train"):
        x, _ = data.TrainBatcher(crop, device=DEV, seed=seed)(decoded, torch.from_numpy(fx["labels"]))
    else:
        x = data.val_batch(decoded, crop, device=DEV)
    got = x.cpu().numpy()
    assert got.shape == fx["images"].shape
    assert np.array_equal(got == 0, fx["images"] == 0)
    np.testing.assert_allclose(got, fx["images"], rtol=0, atol=2e-5)
