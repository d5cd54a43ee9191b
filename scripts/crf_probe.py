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


def cpu_reference_filter(img, k):
    """the reference's own lattice (oracle/_ref, built from wrapper/bilateralfilter/*.cpp) on one host core: init + K planes"""
    from oracle import crf_oracle as C
    lib = C.load_ref()
    if lib is None:
        return None
    h, w = img.shape[:2]
    planes = np.random.default_rng(1).random((k, h, w)).astype(np.float32)
    C.ref_bilateralfilter(lib, img, planes[:1], 13.0, 80.0)
    t0 = time.time()
    C.ref_bilateralfilter(lib, img, planes, 13.0, 80.0)
    return time.time() - t0


def gpu_filter(img, k, reps):
    from acr_wsss_amd.crf import PermutohedralLattice
    h, w = img.shape[:2]
    x = torch.rand(k, h * w, device="cuda")
    torch.cuda.synchronize()
    t0 = time.time()
    lat = PermutohedralLattice(h, w, 80, rgb=img, srgb=13)
    torch.cuda.synchronize()
    t_cold = time.time() - t0                              # first call of the process: code-object load, hipCUB temp storage, allocator
    t0 = time.time()
    for _ in range(reps):
        lat = PermutohedralLattice(h, w, 80, rgb=img, srgb=13)
    torch.cuda.synchronize()
    t_build = (t_cold, (time.time() - t0) / reps)
    lat.filter(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        lat.filter(x)
    torch.cuda.synchronize()
    return t_build, (time.time() - t0) / reps, lat.n_points


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    img, _ = scene(375, 500, 21, 0)
    tb, tf, m = gpu_filter(img, 21, reps)
    print("bilateral lattice 375x500 (sxy 80, srgb 13): %d points; build (host upload of the image + pair kernel + sort + scan + tables), "
          "steady state %.2f ms (the first build of the process: %.2f ms, code-object load and allocator included); filter of 21 planes "
          "(splat + 6 blurs + slice, lattice already built) %.3f ms" % (m, 1e3 * tb[1], 1e3 * tb[0], 1e3 * tf))
    tc = cpu_reference_filter(img, 21)
    if tc is not None:
        print("reference C++ lattice (bilateralfilter.cpp:22-41, 1 core, init + 21 planes): %.1f ms" % (1e3 * tc))
    for k in (3, 21):
        img, probs = scene(375, 500, k, 0)
        crf_inference(img, probs, labels=k)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(reps):
            crf_inference(img, probs, labels=k)
        torch.cuda.synchronize()
        print("crf_inference 375x500, %2d labels, 10 iterations: %.1f ms / image (everything: upload of image + unary, TWO lattice builds "
              "(spatial and bilateral), 2 normalisation filters, 10 x (2 filters + update), download of Q)" % (k, 1e3 * (time.time() - t0) / reps))
