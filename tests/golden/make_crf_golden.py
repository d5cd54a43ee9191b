"""Generates tests/golden/crf_lattice_*.npz from the REFERENCE's own permutohedral lattice
(/root/reference/wrapper/bilateralfilter/*.cpp compiled by oracle/Makefile into oracle/_ref/libpermuto_ref.so): seeded inputs,
the tables of Permutohedral::init (offsets, barycentric weights, number of lattice points) and the outputs of
Permutohedral::compute for the two kernels tool/imutils.py:356-357 configures (bilateral sxy 80 / srgb 13, spatial sxy 3), plus
the bilateralfilter() entry point (bilateralfilter.cpp:22-41).  Run in the build container:  python tests/golden/make_crf_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import crf_oracle as C      # noqa: E402  (feature builders + ctypes wrappers only; outputs come from the .so)


def make_image(h, w, seed, structured):
    rng = np.random.default_rng(seed)
    if not structured:
        return rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 5) % 256, (yy * 7) % 256, ((xx + yy) * 3) % 256], -1).astype(np.uint8)
    img[h // 3:, w // 2:] = (200, 30, 60)                      # a flat region: long splat runs
    img[: h // 4, : w // 3] += rng.integers(0, 4, (h // 4, w // 3, 3)).astype(np.uint8)
    return img


def main():
    lib = C.load_ref()
    assert lib is not None, "build oracle/_ref first: make -C oracle"
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for tag, (h, w, K, seed, structured) in {"a": (24, 32, 3, 1, False), "b": (40, 52, 2, 2, True), "c": (61, 47, 4, 3, True),
                                                    "d": (25, 25, 2, 4, False)}.items():      # N % 4 = 1: the SSE padding quirk
        img = make_image(h, w, seed, structured)
        rng = np.random.default_rng(100 + seed)
        vals = rng.random((h * w, K)).astype(np.float32)
        rec = {"img": img, "vals": vals}
        for name, feat in (("bil", C.bilateral_features(img, 80, 13)), ("spa", C.spatial_features(h, w, 3))):
            off, wts, m = C.ref_lattice_tables(lib, feat)
            filt, m2 = C.ref_lattice_filter(lib, feat, vals)
            assert m == m2
            rec[name + "_offsets"], rec[name + "_weights"], rec[name + "_M"], rec[name + "_filter"] = off, wts, np.int64(m), filt
        rec["entry_bilateralfilter"] = C.ref_bilateralfilter(lib, img, vals.T.reshape(K, h, w).copy(), 13.0, 80.0)
        path = os.path.join(out_dir, "crf_lattice_%s.npz" % tag)
        np.savez_compressed(path, **rec)
        print(path, os.path.getsize(path), "bytes; lattice points:", int(rec["bil_M"]), int(rec["spa_M"]))


if __name__ == "__main__":
    main()
