"""Dense-CRF stage at VOC size (375 x 500, 21 labels, 10 iterations) for rocprofv3 / timing:  python scripts/crf_probe.py [reps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acr_wsss_amd.crf import crf_inference  # noqa: E402


def scene(h, w, k, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 3) % 256, (yy * 2) % 256, (xx * yy) % 256], -1).astype(np.uint8)
    img[h // 4: 3 * h // 4, w // 3: 2 * w // 3] = (230, 40, 40)
    img = (img.astype(np.int64) + rng.integers(0, 12, img.shape)).clip(0, 255).astype(np.uint8)      # sensor-like noise
    probs = rng.random((k, h, w)).astype(np.float32) * 0.3
    probs[1, h // 4: 3 * h // 4, w // 3: 2 * w // 3] += 0.6
    probs[0] += 0.3
    return img, probs


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    for k in (3, 21):
        img, probs = scene(375, 500, k, 0)
        crf_inference(img, probs, labels=k)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(reps):
            crf_inference(img, probs, labels=k)
        torch.cuda.synchronize()
        print("crf_inference 375x500, %2d labels, 10 iterations: %.1f ms / image" % (k, 1e3 * (time.time() - t0) / reps))
