"""Single-pass threshold sweep (acr_wsss_amd/evaluation.py) == the reference's per-threshold loop, restated by the
oracle (oracle.acr_oracle.seeds_from_cam_dict + iou_counts <- evaluation.py:19-49), counter for counter."""
import numpy as np

from oracle import acr_oracle as O
from acr_wsss_amd.evaluation import SweepCounters, evaluate_cam_dir


def _case(rng, h, w, classes, ties=False):
    cams = {c: rng.random((h, w)).astype(np.float32) for c in classes}
    if ties:
        for c in classes:
            cams[c] = np.round(cams[c] * 10) / 10          # many exact ties between classes and with thresholds
        cams[classes[0]][: h // 3] = 0.0                    # all-zero regions: absent classes tie with present ones
    gt = rng.integers(0, 21, (h, w)).astype(np.uint8)
    gt[rng.random((h, w)) < 0.15] = 255
    return cams, gt


def test_sweep_matches_reference_loop():
    rng = np.random.default_rng(0)
    thresholds = np.arange(100, dtype=np.float32) / 100.0
    sc = SweepCounters(thresholds)
    images = [_case(rng, 37, 53, [3, 11]), _case(rng, 20, 31, [0], ties=True), _case(rng, 25, 18, [0, 7, 19], ties=True),
              _case(rng, 16, 16, list(range(20)))]
    for cams, gt in images:
        sc.add(cams, gt)
    for k in (0, 1, 20, 40, 50, 77, 99):
        TP = np.zeros(21, np.int64); P = np.zeros(21, np.int64); T = np.zeros(21, np.int64)
        for cams, gt in images:
            pred = O.seeds_from_cam_dict(cams, thresholds[k])
            tp, p, t = O.iou_counts(pred, gt)
            TP += tp; P += p; T += t
        np.testing.assert_array_equal(sc.TP[k], TP, err_msg="TP t=%.2f" % thresholds[k])
        np.testing.assert_array_equal(sc.P[k], P, err_msg="P t=%.2f" % thresholds[k])
        np.testing.assert_array_equal(sc.T, T)
        assert abs(sc.miou()[0][k] - O.miou(TP, P, T)) < 1e-9


def test_evaluate_dir_roundtrip(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    names = []
    for i in range(3):
        cams, gt = _case(rng, 24, 30, [2, 5])
        np.save(tmp_path / ("im%d.npy" % i), cams)             # infer_cam.py:228 wire format
        Image.fromarray(gt).save(tmp_path / ("im%d.png" % i))
        names.append("im%d" % i)
    t, miou, sc = evaluate_cam_dir(str(tmp_path), str(tmp_path), names)
    assert len(t) == 100 and miou.shape == (100,) and np.all(miou >= 0) and np.all(miou <= 100)
    assert sc.T.sum() > 0
