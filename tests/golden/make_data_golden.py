#!/usr/bin/env python3
"""Pin the input-pipeline oracle (oracle/data_oracle.py) against runs of the REFERENCE's own chunk functions.

Build container only (``/root/reference`` does not travel):   python tests/golden/make_data_golden.py
writes tests/golden/data_chunk_{train,val}_*.npz.

What runs: ``myTool.get_data_from_chunk_v2`` (myTool.py:1158-1199) and ``myTool.get_data_from_chunk_val`` (:1364-1403), imported
UNMODIFIED from /root/reference, with the reference's real ``RandomResizeLong`` (:995-1008), ``flip`` (:895-899), ``RandomCrop``
(:923-955), normalisation, HWC->CHW and chunk assembly, driven by Python's ``random`` and ``np.random`` seeded here (the
reference leaves them unseeded: train_acr.py:23 is commented out).

What is stubbed, and why the stubs pin nothing they should not
---------------------------------------------------------------
``myTool.py`` imports cv2, torchvision and (through tool/imutils.py) pydensecrf at module top; none is installed here.  Throw-away
stub modules are put on sys.modules for the import only:
  * ``cv2.imread``      returns a seeded BGR uint8 array per file name (there are no image files): the DECODE is not under test;
  * ``cv2.cvtColor``    BGR -> RGB channel reversal (the one conversion the chunk functions ask for);
  * ``cv2.resize``      **is oracle/data_oracle.cv2_resize_linear** -- OpenCV's published float INTER_LINEAR rule.  The resize
                        arithmetic therefore stays the ONE UNPINNED step of this row (it is checked against itself); everything
                        around it -- the draw order of the two generators, the target shape (incl. Python's banker's rounding
                        in ``int(round(...))``), the flip, the float64 normalisation, crop placement for images larger AND
                        smaller than the crop, the zero container, float32 conversion, layout -- is the reference's own code;
  * ``torchvision.transforms.Compose`` / ``pydensecrf``   empty shells (never called on this path).
numpy >= 1.24 removed the aliases ``np.float`` / ``np.bool`` the reference (pinned to an older numpy) uses at :929,1177: they are
restored as ``float`` / ``bool`` for the run.  ``voc12/cls_labels.npy`` (myTool.py:916-920; not in the reference tree) is written
to a temp dir with seeded multi-hot vectors and the script runs from there.
"""
import os
import random
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("ACR_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
from oracle import data_oracle as DO  # noqa: E402  (only as the body of the cv2.resize stub)

SIZES = {}          # file stem -> (h, w) of the "decoded" image


def decoded_bgr(stem):
    """The array the cv2.imread stub returns for <stem>.jpg: uint8 BGR, a pure function of the name."""
    h, w = SIZES[stem]
    seed = int.from_bytes(stem.encode(), "little") % (2 ** 31)
    return np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)


def install_stubs():
    cv2 = types.ModuleType("cv2")
    cv2.COLOR_BGR2RGB, cv2.INTER_NEAREST, cv2.INTER_LINEAR = 4, 0, 1
    cv2.imread = lambda path, *a: decoded_bgr(os.path.splitext(os.path.basename(path))[0])

    def cvt(img, code):
        assert code == cv2.COLOR_BGR2RGB
        return np.ascontiguousarray(img[:, :, ::-1])

    def resize(img, dsize, interpolation=None, **kw):
        assert interpolation in (None, cv2.INTER_LINEAR)
        return DO.cv2_resize_linear(img, int(dsize[0]), int(dsize[1]))
    cv2.cvtColor, cv2.resize = cvt, resize
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvt.Compose = type("Compose", (), {})
    tv.transforms = tvt
    crf = types.ModuleType("pydensecrf")
    crfd = types.ModuleType("pydensecrf.densecrf")
    crfu = types.ModuleType("pydensecrf.utils")
    crfu.unary_from_labels = lambda *a, **k: None
    crfu.unary_from_softmax = lambda *a, **k: None
    crf.densecrf, crf.utils = crfd, crfu
    for name, mod in (("cv2", cv2), ("torchvision", tv), ("torchvision.transforms", tvt), ("pydensecrf", crf),
                      ("pydensecrf.densecrf", crfd), ("pydensecrf.utils", crfu)):
        sys.modules[name] = mod
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "bool"):
        np.bool = bool


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import myTool                                                  # the reference, unmodified
    work = tempfile.mkdtemp(prefix="acr_data_golden_")
    os.makedirs(os.path.join(work, "voc12"))
    cases = {
        # crop, [(stem, h, w)], seed   -- images larger than the crop in both, one, or no dimension; odd sizes; a chunk of one
        "train_a": (48, [("2007_000001", 60, 90), ("2007_000002", 90, 60), ("2007_000003", 30, 40), ("2007_000004", 48, 48)], 3),
        "train_b": (64, [("2008_000011", 37, 113), ("2008_000012", 200, 150), ("2008_000013", 64, 80)], 12),
        "train_c": (32, [("2009_000021", 33, 31)], 5),
        "val_a": (48, [("2007_000001", 60, 90), ("2007_000003", 30, 40), ("2007_000004", 48, 48)], 7),
        "val_b": (40, [("2008_000012", 200, 150)], 8),
    }
    labels = {}
    lr = np.random.RandomState(99)
    for _, (_, imgs, _) in cases.items():
        for stem, h, w in imgs:
            SIZES[stem] = (h, w)
            if stem not in labels:
                labels[stem] = (lr.rand(20) > 0.8).astype(np.float32)
    np.save(os.path.join(work, "voc12", "cls_labels.npy"), labels)
    os.chdir(work)
    args = types.SimpleNamespace(IMpath=os.path.join(work, "JPEGImages"), crop_size=0)
    for name, (crop, imgs, seed) in cases.items():
        chunk = [stem for stem, _, _ in imgs]
        args.crop_size = crop
        random.seed(seed)
        np.random.seed(seed)
        fn = myTool.get_data_from_chunk_v2 if name.startswith("train") else myTool.get_data_from_chunk_val
        images, ori_images, lab, name_list = fn(chunk, args)
        assert list(name_list) == chunk and tuple(images.shape) == (len(chunk), 3, crop, crop)
        out = {"crop": np.int64(crop), "seed": np.int64(seed), "images": images.numpy().astype(np.float32),
               "ori_images": np.asarray(ori_images, np.uint8), "labels": lab.numpy().astype(np.float32)}
        for i, stem in enumerate(chunk):
            out["rgb_%d" % i] = np.ascontiguousarray(decoded_bgr(stem)[:, :, ::-1])          # what cvtColor hands on: RGB uint8
        path = os.path.join(HERE, "data_chunk_%s.npz" % name)
        np.savez_compressed(path, **out)
        print("%s: %s crop %d -> images %s, |x| max %.4f, zeros %.1f %%  (%d bytes)" % (
            name, [SIZES[s] for s in chunk], crop, tuple(images.shape), float(images.abs().max()),
            100.0 * float((images == 0).float().mean()), os.path.getsize(path)))


if __name__ == "__main__":
    main()
