"""Input pipeline of the ACR training / CAM steps on the device.

Counterpart of myTool.py:1158-1199 (`get_data_from_chunk_v2`) and :1364-1403 (`get_data_from_chunk_val`): the
reference decodes with cv2 on the training process's CPU and does resize / flip / normalise / crop in numpy float64,
one image at a time, synchronously; at >100 img/s/GPU that starves the device.  Here the host only decodes (any
decoder -- PIL is what this image has) and draws the geometry; ONE H2D copy of the packed uint8 pixels and ONE kernel
launch (`acr_preprocess_batch`, include/acr_hip.h) produce the (B,3,S,S) network input:

  random resize-long to [0.9*S, S/0.875]   (RandomResizeLong :995-1008; cv2.resize float-path bilinear)
  horizontal flip with probability 1/2     (flip :895-899)
  (x/255 - mean) / std                     (:1180-1182)
  zero-padded random crop to S x S         (RandomCrop :923-955)

and returns the same contract as the reference: images (B,3,S,S) + labels (B,C) from the `cls_labels.npy` dict
(voc12/make_cls_labels.py:18-22).  The draws follow the reference's order (np.random.uniform for the flip, then
random.randint / random.randrange) from SEEDABLE generators (the reference's are the unseeded globals, train_acr.py:23).
There is no CPU path: tensors land on the GPU through the HIP kernel or the call raises.
"""
import concurrent.futures
import ctypes
import os
import random as _pyrandom

import numpy as np
import torch

from . import _lib as L

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)

PRE_IMAGE = np.dtype([("offset", "<i8"), ("h", "<i4"), ("w", "<i4"), ("rh", "<i4"), ("rw", "<i4"), ("flip", "<i4"),
                      ("cont_top", "<i4"), ("cont_left", "<i4"), ("img_top", "<i4"), ("img_left", "<i4"), ("ch", "<i4"),
                      ("cw", "<i4"), ("_pad", "<i4")])      # struct acr_pre_image: 8-byte aligned -> 56 bytes
assert PRE_IMAGE.itemsize == 56


def load_cls_labels(path, names, num_classes=20):
    """`cls_labels.npy`: pickled {image name: float32 (C,)} (voc12/make_cls_labels.py)."""
    d = np.load(path, allow_pickle=True).item()
    return torch.from_numpy(np.stack([np.asarray(d[n], dtype=np.float32) for n in names]))


def resize_long_target(h, w, target_long):
    """myTool.py:995-1005: the longer side becomes target_long, the other is rounded.  Returns (new_h, new_w)."""
    if w < h:
        return target_long, int(round(w * target_long / h))
    return int(round(h * target_long / w)), target_long


def random_crop_boxes(h, w, crop, rng):
    """myTool.py:923-948 -> (cont_top, cont_left, img_top, img_left, ch, cw); w is drawn before h, like the reference.
    ``rng``: a ``random.Random``."""
    ch, cw = min(crop, h), min(crop, w)
    w_space, h_space = w - crop, h - crop
    if w_space > 0:
        cont_left, img_left = 0, rng.randrange(w_space + 1)
    else:
        cont_left, img_left = rng.randrange(-w_space + 1), 0
    if h_space > 0:
        cont_top, img_top = 0, rng.randrange(h_space + 1)
    else:
        cont_top, img_top = rng.randrange(-h_space + 1), 0
    return cont_top, cont_left, img_top, img_left, ch, cw


def preprocess_batch(images_uint8, records, S, device, dtype=torch.float32):
    """Run acr_preprocess_batch: ``images_uint8`` = list of (h,w,3) uint8 RGB arrays, ``records`` = PRE_IMAGE array with
    everything but ``offset`` filled in.  One pinned staging buffer, one H2D copy, one launch."""
    device = torch.device(device)
    if device.type != "cuda":
        raise L.AcrHipError("acr_wsss_amd.data runs on the GPU only (acr_preprocess_batch); there is no CPU path")
    sizes = [int(a.shape[0]) * int(a.shape[1]) * 3 for a in images_uint8]
    offs = np.concatenate([[0], np.cumsum([(s + 15) // 16 * 16 for s in sizes])])       # 16-byte aligned images
    stage = torch.empty(int(offs[-1]), dtype=torch.uint8).pin_memory()
    sn = stage.numpy()
    for a, o, s in zip(images_uint8, offs[:-1], sizes):
        assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3, "decoded images must be (h, w, 3) uint8 RGB"
        sn[o:o + s] = np.ascontiguousarray(a).reshape(-1)
    records = records.copy()
    records["offset"] = offs[:-1]
    for rec, a in zip(records, images_uint8):               # the kernel trusts the table: validate it here
        ok = (rec["h"] == a.shape[0] and rec["w"] == a.shape[1] and rec["rh"] > 0 and rec["rw"] > 0
              and 0 <= rec["cont_top"] and rec["cont_top"] + rec["ch"] <= S and 0 <= rec["cont_left"] and rec["cont_left"] + rec["cw"] <= S
              and 0 <= rec["img_top"] and rec["img_top"] + rec["ch"] <= rec["rh"] and 0 <= rec["img_left"]
              and rec["img_left"] + rec["cw"] <= rec["rw"])
        if not ok:
            raise L.AcrHipError("acr_preprocess_batch: inconsistent geometry record %s for a %s image" % (rec, a.shape))
    packed = stage.to(device, non_blocking=True)
    raw = records.view(np.uint8).reshape(-1)
    table_host = torch.empty(raw.size, dtype=torch.uint8, pin_memory=True)    # pinned: a pageable source would make the
    table_host.numpy()[:] = raw                                                # "asynchronous" copy drain the stream on the host
    table = table_host.to(device, non_blocking=True)
    out = torch.empty((len(images_uint8), 3, S, S), dtype=dtype, device=device)
    mean = (ctypes.c_float * 3)(*MEAN)
    std = (ctypes.c_float * 3)(*STD)
    with torch.cuda.device(device):
        L.check(L.load().acr_preprocess_batch(L.ptr(packed), L.ptr(table), len(images_uint8), S, mean, std, L.dtype_code(dtype)
                                              if dtype != torch.bfloat16 else L.ACR_BF16, L.ptr(out), L.stream_ptr()),
                "acr_preprocess_batch")
    # the staging buffer and the table must outlive the asynchronous copies: tie them to the output
    out._acr_keep = (stage, packed, table)
    return out


class TrainBatcher:
    """get_data_from_chunk_v2 for a chunk of decoded images.  ``seed`` seeds both generators the reference draws from."""

    def __init__(self, crop_size, device="cuda", seed=None, dtype=torch.float32):
        self.S = crop_size
        self.device = torch.device(device)
        self.dtype = dtype
        self.pyrandom = _pyrandom.Random(seed)
        self.nprandom = np.random.RandomState(seed)

    def draw(self, h, w):
        """The per-image draws in the reference's order: flip_p (:1175), target_long (:996), crop boxes (:935-945)."""
        S = self.S
        flip_p = self.nprandom.uniform(0, 1)
        target_long = self.pyrandom.randint(int(S * 0.9), int(S / 0.875))
        nh, nw = resize_long_target(h, w, target_long)
        ct, cl, it, il, ch, cw = random_crop_boxes(nh, nw, S, self.pyrandom)
        return (0, h, w, nh, nw, int(flip_p > 0.5), ct, cl, it, il, ch, cw, 0)

    def __call__(self, images_uint8, labels):
        """images_uint8: list of (h,w,3) uint8 RGB arrays; labels: (B,C) tensor.  Returns (img, label) on the device."""
        self.nprandom.uniform(0.7, 1.3)                     # myTool.py:1161: `scale`, drawn once per chunk and never used
        rec = np.zeros(len(images_uint8), PRE_IMAGE)
        for i, a in enumerate(images_uint8):
            rec[i] = self.draw(int(a.shape[0]), int(a.shape[1]))
        self.last_records = rec
        return preprocess_batch(images_uint8, rec, self.S, self.device, self.dtype), labels.to(self.device, non_blocking=True)


def val_batch(images_uint8, crop_size, device="cuda", dtype=torch.float32):
    """myTool.py:1364-1403: plain resize to crop x crop + normalise (no augmentation)."""
    S = crop_size
    rec = np.zeros(len(images_uint8), PRE_IMAGE)
    for i, a in enumerate(images_uint8):
        rec[i] = (0, int(a.shape[0]), int(a.shape[1]), S, S, 0, 0, 0, 0, 0, S, S, 0)
    return preprocess_batch(images_uint8, rec, S, device, dtype)


# ------------------------------------------------------------------------------------------------------------------
# The reference's own call contract: names in, tensors out (myTool.py:1158-1199, :1364-1403)
# ------------------------------------------------------------------------------------------------------------------
def decode_rgb(path):
    """One image file -> (h, w, 3) uint8 RGB (what cv2.imread + cvtColor(BGR2RGB) hand to the reference, :1176-1177).  PIL
    is the decoder this image ships (cv2 is absent); it releases the GIL while it decodes, so a thread pool scales."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


class ChunkLoader:
    """`get_data_from_chunk_v2(chunk, args)` / `get_data_from_chunk_val(chunk, args)` for a training process that must not
    wait for its input: files are read and decoded by a small thread pool, geometry is drawn on the host in the reference's
    RNG order, and ONE pinned H2D copy + ONE `acr_preprocess_batch` launch build the (B,3,S,S) batch on the GPU.

        loader = ChunkLoader(img_dir, cls_labels, crop_size, device="cuda", seed=None, workers=8)
        images, ori_images, labels, names = loader.get_data_from_chunk_v2(chunk)        # one chunk, synchronous decode
        for images, ori_images, labels, names in loader.iterate(chunks, train=True):    # decode of chunk i+1 overlaps step i

    `cls_labels`: the `voc12/cls_labels.npy` dict {name: float32 (C,)} (myTool.py:916-920) or its path.  `ori_images`
    (de-normalised uint8 crops, :1186-1190, which no caller of the training loop reads) is None unless `with_ori=True`."""

    def __init__(self, img_dir, cls_labels, crop_size, device="cuda", seed=None, workers=8, dtype=torch.float32, ext=".jpg",
                 with_ori=False):
        self.img_dir, self.S, self.ext, self.with_ori = img_dir, crop_size, ext, with_ori
        self.device, self.dtype = torch.device(device), dtype
        if isinstance(cls_labels, (str, os.PathLike)):
            cls_labels = np.load(cls_labels, allow_pickle=True).item()
        self.cls_labels = cls_labels
        self.batcher = TrainBatcher(crop_size, device, seed, dtype)
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="acr-decode")

    def close(self):
        self.pool.shutdown(wait=True)

    def _submit(self, chunk):
        return [self.pool.submit(decode_rgb, os.path.join(self.img_dir, name + self.ext)) for name in chunk]

    def _labels(self, chunk):
        return torch.from_numpy(np.stack([np.asarray(self.cls_labels[n], dtype=np.float32) for n in chunk]))

    def _ori(self, images):
        if not self.with_ori:
            return None
        mean = torch.tensor(MEAN, device=images.device).view(1, 3, 1, 1)
        std = torch.tensor(STD, device=images.device).view(1, 3, 1, 1)
        return ((images.float() * std + mean) * 255.0).clamp(0, 255).to(torch.uint8).cpu().numpy()     # :1186-1190 (astype truncates)

    def _finish(self, chunk, futures, train):
        decoded = [f.result() for f in futures]
        labels = self._labels(chunk)
        if train:
            images, labels = self.batcher(decoded, labels)
        else:
            self.batcher.nprandom.uniform(0.7, 1.3)         # :1367: the val function draws `scale` too
            images, labels = val_batch(decoded, self.S, self.device, self.dtype), labels.to(self.device, non_blocking=True)
        return images, self._ori(images), labels, list(chunk)

    def get_data_from_chunk_v2(self, chunk):
        """myTool.py:1158-1199 -> (images (B,3,S,S) on the device, ori_images, labels (B,C), name_list)."""
        return self._finish(chunk, self._submit(chunk), True)

    def get_data_from_chunk_val(self, chunk):
        """myTool.py:1364-1403: plain resize to S x S + normalise."""
        return self._finish(chunk, self._submit(chunk), False)

    def iterate(self, chunks, train=True, depth=2):
        """Yield the batches of `chunks` in order while the pool already decodes the next `depth` chunks (the previous step's
        GPU work and this thread's Python run meanwhile; the RNG draws stay in chunk order because they happen here)."""
        chunks = list(chunks)
        pending = [self._submit(c) for c in chunks[:depth]]
        for i, chunk in enumerate(chunks):
            futures = pending.pop(0)
            if i + depth < len(chunks):
                pending.append(self._submit(chunks[i + depth]))
            yield self._finish(chunk, futures, train)


def chunker(seq, size):
    """myTool.py:882-883."""
    return (seq[pos:pos + size] for pos in range(0, len(seq), size))


def read_file(path_to_file):
    """myTool.py:867-873: one image id per line."""
    with open(path_to_file) as f:
        return [line.rstrip("\n") for line in f]
