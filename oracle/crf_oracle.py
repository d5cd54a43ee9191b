"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the dense-CRF stage of the reference.

The reference refines CAMs with ``imutils.crf_inference`` (tool/imutils.py:345-362, called from infer_cam.py:27-40,218-225), i.e.
``pydensecrf`` -- a third-party dependency that is NOT vendored in /root/reference and not installed here (the reference pins no
version; the wrapper is lucasb-eyer/pydensecrf around Kraehenbuehl & Koltun's densecrf v2, NIPS 2011).  What IS in the reference
tree is the lattice that package is built on: wrapper/bilateralfilter/permutohedral.{hpp,cpp} ("modified from Philipp
Kraehenbuehl's NIPS 2011 code").  So parity is anchored in two steps:

  * the permutohedral lattice (init tables + splat / blur / slice) restated here in numpy follows permutohedral.cpp:112-283
    (init, the SSE branch g++ compiles on x86-64) and :441-520 (compute) operation by operation in float32, and is pinned
    BIT-EXACTLY against that very code compiled from the reference sources (oracle/Makefile -> oracle/_ref/libpermuto_ref.so)
    and against the fixtures generated from it (tests/golden/crf_lattice_*.npz, tests/golden/make_crf_golden.py);
  * the mean-field loop around the filter (unary_from_softmax, NORMALIZE_SYMMETRIC kernels, Potts compatibility, expAndNormalize,
    ``inference(t)``) restates the published densecrf v2 algorithm as pydensecrf's defaults configure it.  pydensecrf itself
    cannot run here, so THIS PART IS "parity unpinned": nothing checks it against an execution of the reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import ctypes
import os

import numpy as np

F = np.float32


# ------------------------------------------------------------------------------------------------------------------
# permutohedral lattice
# ------------------------------------------------------------------------------------------------------------------
def lattice_init(features):
    """permutohedral.cpp:112-283.  features (N, d) float32 -> dict(offsets (N, d+1) int32 numbered by first appearance like the
    reference's hash table, weights (N, d+1) float32, neighbors (d+1, M, 2) int32 with -1 = absent, keys (M, d) int16, M)."""
    f = np.ascontiguousarray(features, dtype=F)
    n_real, d = f.shape
    # The SSE branch walks the pixels in blocks of 4 and fills the lanes past N with ZERO features (:161-163) -- and still
    # inserts their d+1 vertex keys into the hash table (:232-240).  When N % 4 != 0 the lattice therefore owns the vertices of
    # the simplex around the origin even if no pixel touches them; they receive no splat but do take part in the blur, which
    # leaks a little mass through them.  One phantom pixel reproduces that (all padded lanes have the same keys).
    if n_real % 4:
        f = np.concatenate([f, np.zeros((1, d), F)], axis=0)
    N = f.shape[0]
    inv_std_dev = F(np.sqrt(2.0 / 3.0) * (d + 1))                                                  # :146
    scale = np.array([1.0 / np.sqrt(float((i + 2) * (i + 1))) * float(inv_std_dev) for i in range(d)]).astype(F)   # :148-149
    invdplus1 = F(1.0) / F(d + 1)
    dplus1 = F(d + 1)

    elevated = np.zeros((N, d + 1), F)
    sm = np.zeros(N, F)
    for j in range(d, 0, -1):                                                                        # :167-173
        cf = f[:, j - 1] * scale[j - 1]
        elevated[:, j] = sm - F(j) * cf
        sm = sm + cf
    elevated[:, 0] = sm

    v = np.rint(invdplus1 * elevated).astype(F)                 # :176-186 (cvtps_epi32 under ROUND_NEAREST = half to even)
    rem0 = v * dplus1
    total = np.zeros(N, F)
    for i in range(d + 1):
        total = total + v[:, i]

    rank = np.zeros((N, d + 1), F)                                                                    # :189-199
    for i in range(d):
        di = elevated[:, i] - rem0[:, i]
        for j in range(i + 1, d + 1):
            dj = elevated[:, j] - rem0[:, j]
            c = (di < dj).astype(F)
            rank[:, i] += c
            rank[:, j] += F(1) - c
    for i in range(d + 1):                                                                            # :202-208
        rank[:, i] += total
        add = np.where(rank[:, i] < 0, dplus1, F(0))
        sub = np.where(rank[:, i] >= dplus1, dplus1, F(0))
        rank[:, i] += add - sub
        rem0[:, i] += add - sub

    bary = np.zeros((N, d + 2), F)                                                                    # :211-224
    rows = np.arange(N)
    irank = rank.astype(np.int64)
    for i in range(d + 1):
        vv = (elevated[:, i] - rem0[:, i]) * invdplus1
        p = d - irank[:, i]
        bary[rows, p] += vv
        bary[rows, p + 1] -= vv
    bary[:, 0] += F(1) + bary[:, d + 1]                                                               # :229

    canonical = np.zeros((d + 1, d + 1), np.int64)                                                    # :137-142
    for i in range(d + 1):
        canonical[i, :d - i + 1] = i
        canonical[i, d - i + 1:] = i - (d + 1)
    keys = np.empty((N, d + 1, d), np.int16)                                                          # :234-240
    for r in range(d + 1):
        for i in range(d):
            keys[:, r, i] = (rem0[:, i] + canonical[r, irank[:, i]].astype(F)).astype(np.int16)
    flat = keys.reshape(N * (d + 1), d)
    uniq, first, inv = np.unique(flat, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")                    # the hash table hands out ids in order of first insertion
    relabel = np.empty(len(uniq), np.int64)
    relabel[order] = np.arange(len(uniq))
    offsets = relabel[inv.reshape(-1)].reshape(N, d + 1).astype(np.int32)[:n_real]
    bary = bary[:n_real]
    pts = uniq[order]
    M = len(pts)

    index = {tuple(k): i for i, k in enumerate(pts.tolist())}                                          # :268-281
    nb = np.full((d + 1, M, 2), -1, np.int32)
    for j in range(d + 1):
        n1 = pts.astype(np.int64) - 1
        n2 = pts.astype(np.int64) + 1
        if j < d:
            n1[:, j] = pts[:, j].astype(np.int64) + d
            n2[:, j] = pts[:, j].astype(np.int64) - d
        for i in range(M):
            nb[j, i, 0] = index.get(tuple(n1[i].tolist()), -1)
            nb[j, i, 1] = index.get(tuple(n2[i].tolist()), -1)
    return {"offsets": offsets, "weights": np.ascontiguousarray(bary[:, :d + 1]), "neighbors": nb, "keys": pts, "M": M, "d": d}


def lattice_compute(lat, values):
    """permutohedral.cpp:441-520 (SSE branch).  values (N, K) float32 -> filtered (N, K) float32."""
    off, w, nb, M, d = lat["offsets"], lat["weights"], lat["neighbors"], lat["M"], lat["d"]
    x = np.ascontiguousarray(values, dtype=F)
    N, K = x.shape
    vals = np.zeros((M + 2, K), F)
    # splat: values[o] += w * in[i], pixel after pixel, vertex after vertex (np.add.at applies the updates in index order)
    contrib = (w.reshape(N, d + 1, 1) * x.reshape(N, 1, K)).reshape(N * (d + 1), K)
    np.add.at(vals, off.reshape(-1) + 1, contrib)
    half = F(0.5)
    for j in range(d + 1):                                                                            # blur along each axis
        new = np.zeros_like(vals)
        new[1:M + 1] = vals[1:M + 1] + half * (vals[nb[j, :, 0] + 1] + vals[nb[j, :, 1] + 1])
        vals = new
    alpha = F(1.0) / (F(1) + F(np.power(F(2), F(-d))))
    out = np.zeros((N, K), F)
    for j in range(d + 1):                                                                            # slice
        wj = (w[:, j] * alpha).astype(F)
        out += wj[:, None] * vals[off[:, j] + 1]
    return out


def spatial_features(h, w, sxy):
    """densecrf.cpp addPairwiseGaussian: (x / sx, y / sy) per pixel, row-major."""
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    return np.stack([xs.reshape(-1).astype(F) / F(sxy), ys.reshape(-1).astype(F) / F(sxy)], axis=1).astype(F)


def bilateral_features(img, sxy, srgb):
    """densecrf.cpp addPairwiseBilateral / bilateralfilter.cpp:4-20: (x / sxy, y / sxy, r / srgb, g / srgb, b / srgb)."""
    h, w = img.shape[:2]
    rgb = img.reshape(h * w, 3).astype(F) / F(srgb)
    return np.concatenate([spatial_features(h, w, sxy), rgb], axis=1).astype(F)


# ------------------------------------------------------------------------------------------------------------------
# mean field (published densecrf v2 algorithm with pydensecrf's defaults -- parity unpinned, see module docstring)
# ------------------------------------------------------------------------------------------------------------------
def _exp_and_normalize(x):
    e = np.exp(x - x.max(axis=0, keepdims=True))
    return (e / e.sum(axis=0, keepdims=True)).astype(F)


class _Kernel:
    """DenseKernel with DIAG_KERNEL + NORMALIZE_SYMMETRIC (pydensecrf's defaults) and a Potts weight."""

    def __init__(self, features, compat):
        self.lat = lattice_init(features)
        ones = np.ones((features.shape[0], 1), F)
        self.norm = (F(1.0) / np.sqrt(lattice_compute(self.lat, ones)[:, 0] + F(1e-20))).astype(F)
        self.compat = F(compat)

    def apply(self, q):                                           # q (K, N) -> w * norm * filter(norm * q)
        filt = lattice_compute(self.lat, np.ascontiguousarray((q * self.norm[None]).T)).T
        return self.compat * (filt * self.norm[None])


def crf_inference(img, probs, t=10, scale_factor=1, labels=21, log_dtype=np.float64):
    """tool/imutils.py:345-362: unary_from_softmax(probs) (-log, clip 1e-5), Gaussian (sxy 3, compat 3) + bilateral
    (sxy 80, srgb 13, compat 10) Potts kernels, ``t`` mean-field iterations.  img (h, w, 3) uint8, probs (labels, h, w).
    ``log_dtype``: precision the unary's log is evaluated in before rounding to float32 (pydensecrf takes np.log of whatever
    dtype it is handed; float32 vs float64 differ by <= 1 ulp -- tests use the pair to measure how far such a last-bit
    difference travels through the mean-field iterations)."""
    h, w = img.shape[:2]
    unary = (-np.log(np.clip(probs.reshape(labels, -1).astype(log_dtype), 1e-5, 1.0))).astype(F)
    kernels = [_Kernel(spatial_features(h, w, 3 / scale_factor), 3),
               _Kernel(bilateral_features(img, 80 / scale_factor, 13), 10)]
    q = _exp_and_normalize(-unary)
    for _ in range(t):
        tmp = -unary
        for k in kernels:
            tmp = tmp + k.apply(q)                                # tmp1 -= (-w * filtered)
        q = _exp_and_normalize(tmp)
    return q.reshape(labels, h, w)


def crf_with_alpha(cam_dict, alpha, orig_img):
    """infer_cam.py:27-40: background score (1 - max cam)^alpha stacked in front of the CAMs -> CRF -> {0: bg, cls + 1: ...}."""
    v = np.array(list(cam_dict.values()))
    bg = np.power(1 - np.max(v, axis=0, keepdims=True), alpha)
    score = np.concatenate((bg, v), axis=0)
    q = crf_inference(orig_img, score, labels=score.shape[0])
    out = {0: q[0]}
    for i, key in enumerate(cam_dict.keys()):
        out[key + 1] = q[i + 1]
    return out


# ------------------------------------------------------------------------------------------------------------------
# the reference's own lattice, compiled from /root/reference by oracle/Makefile (present only where it was built)
# ------------------------------------------------------------------------------------------------------------------
def load_ref():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libpermuto_ref.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    lib.ref_bilateralfilter.argtypes = [fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float]
    lib.ref_bilateralfilter.restype = None
    lib.ref_lattice_filter.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, fp, ctypes.c_int]
    lib.ref_lattice_filter.restype = ctypes.c_int
    lib.ref_lattice_tables.argtypes = [fp, ctypes.c_int, ctypes.c_int, ip, fp]
    lib.ref_lattice_tables.restype = ctypes.c_int
    return lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def ref_lattice_filter(lib, features, values):
    f = np.ascontiguousarray(features, F)
    x = np.ascontiguousarray(values, F)
    out = np.empty_like(x)
    m = lib.ref_lattice_filter(_fp(f), f.shape[1], f.shape[0], _fp(x), _fp(out), x.shape[1])
    return out, m


def ref_lattice_tables(lib, features):
    f = np.ascontiguousarray(features, F)
    n, d = f.shape
    off = np.empty((n, d + 1), np.int32)
    w = np.empty((n, d + 1), F)
    m = lib.ref_lattice_tables(_fp(f), d, n, off.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _fp(w))
    return off, w, m


def ref_bilateralfilter(lib, img, planes, srgb, sxy):
    """bilateralfilter.cpp:22-41: img (h, w, 3) uint8, planes (K, h, w) -> (K, h, w)."""
    h, w = img.shape[:2]
    image = np.ascontiguousarray(img.transpose(2, 0, 1).astype(F))
    x = np.ascontiguousarray(planes, F)
    out = np.empty_like(x)
    lib.ref_bilateralfilter(_fp(image), _fp(x), _fp(out), x.shape[0], h, w, srgb, sxy)
    return out
