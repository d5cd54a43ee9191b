"""CAM evaluation -- single-pass counterpart of the reference's evaluation.py.

The reference sweeps the background threshold t = 0.00 ... 0.99 (`evaluation.py:127-133`) by re-reading every
`<name>.npy` dict and redoing `argmax([t, cam_0, ...])` per threshold (`:19-33`), 100 passes over the dataset.
Because plane 0 holds the constant t and `np.argmax` returns the first maximum, the prediction at threshold t is

    pred(t) = 0            if t >= m        (m = max_c cam_c at the pixel; ties go to index 0 = background)
              1 + argmax_c cam_c   otherwise

so all thresholds follow from (m, argmax) computed ONCE per pixel: a histogram of m over the threshold grid per
(gt class, argmax class) pair gives every TP/P/T counter of `evaluation.py:37-49` for every t.  Same wire format
in (pickled `{class: float32 (h,w)}` dicts, `infer_cam.py:227-228`), same counters and mIoU out
(`evaluation.py:59-85`).  Pure numpy: this stage is I/O-bound once it is a single pass.
"""
import os

import numpy as np

CATEGORIES = ['background', 'aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow',
              'diningtable', 'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']


class SweepCounters:
    """TP/P/T per class for every threshold of ``thresholds`` (ascending)."""

    def __init__(self, thresholds, num_cls=21):
        self.t = np.asarray(thresholds, dtype=np.float32)
        assert np.all(np.diff(self.t) > 0)
        self.num_cls = num_cls
        nt = len(self.t)
        self.TP = np.zeros((nt, num_cls), np.int64)
        self.P = np.zeros((nt, num_cls), np.int64)
        self.T = np.zeros(num_cls, np.int64)

    def add(self, cam_dict, gt):
        """cam_dict: {class index (0-based, without background): float32 (h,w)}, gt: uint8 (h,w), 255 = ignore."""
        num_cls, nt = self.num_cls, len(self.t)
        keys = sorted(cam_dict.keys())
        cams = np.stack([cam_dict[k] for k in keys]).astype(np.float32)           # (n,h,w)
        # argmax over the (21,h,w) tensor of evaluation.py:27-31: absent classes are zero planes
        m_present = cams.max(axis=0)
        a_present = np.asarray(keys)[cams.argmax(axis=0)] + 1                       # first max among present, label space
        # a zero plane of an absent class wins/ties only if every present cam <= 0 there; first index wins ties
        absent = [c for c in range(num_cls - 1) if c not in cam_dict]
        if absent:
            first_absent = absent[0] + 1
            lower = (m_present < 0) | ((m_present == 0) & (first_absent < a_present))
            m = np.where(lower, 0.0, m_present).astype(np.float32)
            a = np.where(lower, first_absent, a_present)
        else:
            m, a = m_present, a_present
        valid = gt < 255
        m, a, g = m[valid], a[valid], gt[valid].astype(np.int64)
        # number of thresholds with t < m  -> for those the pixel is predicted `a`, for the rest background
        kfg = np.searchsorted(self.t, m, side="left")                                # t[k] < m  <=>  k < kfg
        np.add.at(self.T, g, 1)
        # P: foreground prediction `a` for thresholds [0, kfg), background for [kfg, nt)
        fg = np.zeros((nt + 1, num_cls), np.int64)
        np.add.at(fg, (kfg, a), 1)                     # pixels whose foreground range ends at kfg
        fg_cum = fg[::-1].cumsum(axis=0)[::-1]         # fg_cum[k] = #pixels with kfg >= k
        self.P += fg_cum[1:]                           # threshold index k is foreground iff kfg > k  -> kfg >= k+1
        self.P[:, 0] += (len(m) - fg_cum[1:].sum(axis=1))
        hit = a == g
        tp = np.zeros((nt + 1, num_cls), np.int64)
        np.add.at(tp, (kfg[hit], a[hit]), 1)
        self.TP += tp[::-1].cumsum(axis=0)[::-1][1:]
        bg = g == 0                                    # background pixels are TP whenever predicted background
        bgk = np.bincount(kfg[bg], minlength=nt + 1)
        self.TP[:, 0] += np.cumsum(bgk)[:nt]           # kfg <= k  <=> background at threshold k

    def miou(self):
        """(nt,) mIoU in percent and (nt, num_cls) IoU, evaluation.py:59-74."""
        iou = self.TP / (self.T[None, :] + self.P - self.TP + 1e-10)
        return iou.mean(axis=1) * 100.0, iou * 100.0


def evaluate_cam_dir(predict_dir, gt_dir, name_list, thresholds=None, num_cls=21):
    """Single pass over ``<predict_dir>/<name>.npy`` + ``<gt_dir>/<name>.png`` for all thresholds.
    Returns (thresholds, mIoU per threshold, SweepCounters)."""
    from PIL import Image
    if thresholds is None:
        thresholds = np.arange(100, dtype=np.float32) / 100.0                       # evaluation.py:128-130
    sc = SweepCounters(thresholds, num_cls)
    for name in name_list:
        cam_dict = np.load(os.path.join(predict_dir, name + ".npy"), allow_pickle=True).item()
        gt = np.array(Image.open(os.path.join(gt_dir, name + ".png")))
        sc.add(cam_dict, gt)
    return sc.t, sc.miou()[0], sc
