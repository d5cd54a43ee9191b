"""Input pipeline of the ACR training / CAM steps on the device ("next" row).

Counterpart of myTool.py:1158-1199 (`get_data_from_chunk_v2`) and :1364-1403 (`get_data_from_chunk_val`): the
reference decodes with cv2 on the training process's CPU and does resize / flip / normalise / crop in numpy, one
image at a time, synchronously; at >100 img/s/GPU that starves the device.  Here the host only hands over decoded
uint8 HWC RGB arrays (any decoder); everything else runs on the GPU on the upload stream:

  random resize-long to [0.9*S, S/0.875]   (RandomResizeLong :995-1008; cv2.resize default = bilinear, half-pixel
                                             centres, no anti-aliasing == F.interpolate(bilinear, align_corners=False))
  horizontal flip with probability 1/2     (flip :895-899)
  (x/255 - mean) / std                     (:1180-1182)
  zero-padded random crop to S x S         (RandomCrop :923-955)

and returns the same contract as the reference: images (B,3,S,S) fp32 + labels (B,C) from the `cls_labels.npy`
dict (voc12/make_cls_labels.py:18-22).  Geometry draws come from a seedable numpy Generator (the reference uses
the unseeded `random` module, train_acr.py:23 commented out).  Parity with cv2's resize is NOT pinned (cv2 is not
installed in the build image): the geometry/normalisation contract is tested instead.
"""
import numpy as np
import torch
import torch.nn.functional as F

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _norm_consts(device):
    m = torch.tensor(MEAN, device=device).view(3, 1, 1)
    s = torch.tensor(STD, device=device).view(3, 1, 1)
    return m, s


def load_cls_labels(path, names, num_classes=20):
    """`cls_labels.npy`: pickled {image name: float32 (C,)} (voc12/make_cls_labels.py)."""
    d = np.load(path, allow_pickle=True).item()
    return torch.from_numpy(np.stack([np.asarray(d[n], dtype=np.float32) for n in names]))


def resize_long_target(h, w, target_long):
    """myTool.py:995-1005: the longer side becomes target_long, the other is rounded."""
    if w < h:
        return target_long, int(round(w * target_long / h))          # (new_h, new_w)
    return int(round(h * target_long / w)), target_long


def random_crop_boxes(h, w, crop, rng):
    """myTool.py:923-948 -> (cont_top, cont_left, img_top, img_left, ch, cw)."""
    ch, cw = min(crop, h), min(crop, w)
    w_space, h_space = w - crop, h - crop
    if w_space > 0:
        cont_left, img_left = 0, int(rng.integers(0, w_space + 1))
    else:
        cont_left, img_left = int(rng.integers(0, -w_space + 1)), 0
    if h_space > 0:
        cont_top, img_top = 0, int(rng.integers(0, h_space + 1))
    else:
        cont_top, img_top = int(rng.integers(0, -h_space + 1)), 0
    return cont_top, cont_left, img_top, img_left, ch, cw


class TrainBatcher:
    def __init__(self, crop_size, device="cuda", seed=None):
        self.S = crop_size
        self.device = torch.device(device)
        self.rng = np.random.default_rng(seed)
        self.mean, self.std = _norm_consts(self.device)

    def __call__(self, images_uint8, labels):
        """images_uint8: list of (h,w,3) uint8 RGB arrays; labels: (B,C) tensor.  Returns (img, label) on device."""
        S = self.S
        out = torch.zeros((len(images_uint8), 3, S, S), dtype=torch.float32, device=self.device)
        for i, arr in enumerate(images_uint8):
            t = torch.from_numpy(np.ascontiguousarray(arr))
            if self.device.type == "cuda":
                t = t.pin_memory().to(self.device, non_blocking=True)
            x = t.permute(2, 0, 1).float().unsqueeze(0)                          # (1,3,h,w)
            h, w = x.shape[-2:]
            target_long = int(self.rng.integers(int(S * 0.9), int(S / 0.875) + 1))
            nh, nw = resize_long_target(h, w, target_long)
            x = F.interpolate(x, size=(nh, nw), mode="bilinear", align_corners=False)[0]
            if self.rng.uniform(0, 1) > 0.5:
                x = x.flip(-1)
            x = (x / 255.0 - self.mean) / self.std
            ct, cl, it, il, ch, cw = random_crop_boxes(nh, nw, S, self.rng)
            out[i, :, ct:ct + ch, cl:cl + cw] = x[:, it:it + ch, il:il + cw]
        return out, labels.to(self.device, non_blocking=True)


def val_batch(images_uint8, crop_size, device="cuda"):
    """myTool.py:1364-1403: plain resize to crop x crop + normalise (no augmentation)."""
    device = torch.device(device)
    mean, std = _norm_consts(device)
    out = []
    for arr in images_uint8:
        x = torch.from_numpy(np.ascontiguousarray(arr)).to(device).permute(2, 0, 1).float().unsqueeze(0)
        x = F.interpolate(x, size=(crop_size, crop_size), mode="bilinear", align_corners=False)[0]
        out.append((x / 255.0 - mean) / std)
    return torch.stack(out)
