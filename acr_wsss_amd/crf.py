"""Dense-CRF refinement of CAMs on the GPU: the host mirror of ``imutils.crf_inference``
(tool/imutils.py:345-362 -- pydensecrf's ``DenseCRF2D`` with ``addPairwiseGaussian(sxy=3, compat=3)``,
``addPairwiseBilateral(sxy=80, srgb=13, compat=10)`` and ``inference(t)``) and of ``_crf_with_alpha``
(infer_cam.py:27-40), same names, arguments and return layout.  All arithmetic runs in csrc/crf.hip behind the C ABI
(``acr_lattice_*``, ``acr_crf_*``); there is no CPU path -- without the HIP library and a GPU these raise."""
import ctypes

import numpy as np
import torch

from . import _lib as L


class PermutohedralLattice:
    """One pairwise kernel of the CRF: the lattice of an H x W image with spatial (rgb=None, d = 2) or bilateral
    (rgb = (H, W, 3) uint8, d = 5) features; ``filter`` applies Permutohedral::compute
    (wrapper/bilateralfilter/permutohedral.cpp:441-520) to K planes."""

    def __init__(self, h, w, sxy, rgb=None, srgb=1.0, device="cuda"):
        lib = L.load()
        dev = torch.device(device)
        if dev.type != "cuda":
            raise L.AcrHipError("PermutohedralLattice needs a GPU (no CPU path in the product)")
        self.h, self.w, self.n = int(h), int(w), int(h) * int(w)
        self.d = 2 if rgb is None else 5
        self.device = dev
        if rgb is not None:
            rgb = torch.as_tensor(np.ascontiguousarray(rgb)) if not torch.is_tensor(rgb) else rgb
            if tuple(rgb.shape) != (self.h, self.w, 3) or rgb.dtype != torch.uint8:
                raise ValueError("rgb must be (H, W, 3) uint8, got %s %s" % (tuple(rgb.shape), rgb.dtype))
            rgb = rgb.to(dev).contiguous()
        with torch.cuda.device(dev):
            nbytes = lib.acr_lattice_ws_bytes(self.n, self.d)
            if nbytes < 0:
                L.check(-1, "acr_lattice_ws_bytes")
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            L.check(lib.acr_lattice_build(L.ptr(rgb) if rgb is not None else None, self.h, self.w, float(sxy), float(srgb),
                                          L.ptr(self.ws), nbytes, L.stream_ptr()), "acr_lattice_build")
            m, ovf = ctypes.c_int32(0), ctypes.c_int32(0)
            L.check(lib.acr_lattice_info(L.ptr(self.ws), ctypes.byref(m), ctypes.byref(ovf), L.stream_ptr()), "acr_lattice_info")
        self.n_points = int(m.value)
        self._vals = None

    def filter(self, x, pre=None, post=None, scale=None, out=None):
        """x (K, n) float32 on the device -> scale * post * filter(pre * x), (K, n)."""
        lib = L.load()
        L.require_gpu(x)
        if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != self.n or not x.is_contiguous():
            raise ValueError("x must be a contiguous (K, %d) float32 tensor" % self.n)
        k = x.shape[0]
        need = 2 * k * (self.n_points + 2)
        if self._vals is None or self._vals.numel() < need:
            self._vals = torch.empty(need, dtype=torch.float32, device=self.device)
        out = torch.empty_like(x) if out is None else out
        with torch.cuda.device(self.device):
            L.check(lib.acr_lattice_filter(L.ptr(self.ws), self.n, self.d, self.n_points, L.ptr(x), L.ptr(pre) if pre is not None else None,
                                           L.ptr(out), L.ptr(post) if post is not None else None,
                                           float(scale) if scale is not None else 1.0, 0 if scale is None else 1, k,
                                           L.ptr(self._vals), L.stream_ptr()), "acr_lattice_filter")
        return out

    def tables(self):
        """(offsets (n, d+1) int32, weights (n, d+1) float32, point keys (n_points, d) int16) as numpy arrays."""
        lib = L.load()
        po, pw, pk = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        L.check(lib.acr_lattice_tables(L.ptr(self.ws), self.n, self.d, ctypes.byref(po), ctypes.byref(pw), ctypes.byref(pk)), "acr_lattice_tables")
        base = self.ws.data_ptr()
        e = self.n * (self.d + 1)
        off = self.ws[po.value - base: po.value - base + 4 * e].view(torch.int32).reshape(self.n, self.d + 1).cpu().numpy()
        wts = self.ws[pw.value - base: pw.value - base + 4 * e].view(torch.float32).reshape(self.n, self.d + 1).cpu().numpy()
        packed = self.ws[pk.value - base: pk.value - base + 8 * self.n_points].view(torch.int64).cpu().numpy()
        keys = np.stack([((packed >> (12 * (self.d - 1 - c))) & 4095) - 2048 for c in range(self.d)], axis=1).astype(np.int16)
        return off, wts, keys


class _Kernel:
    """DenseKernel (DIAG_KERNEL, NORMALIZE_SYMMETRIC: pydensecrf's defaults) with a Potts weight."""

    def __init__(self, lattice, compat):
        lib = L.load()
        self.lat = lattice
        self.compat = float(compat)
        ones = torch.ones((1, lattice.n), dtype=torch.float32, device=lattice.device)
        self.norm = lattice.filter(ones).reshape(-1).contiguous()
        with torch.cuda.device(lattice.device):
            L.check(lib.acr_crf_norm(L.ptr(self.norm), lattice.n, L.stream_ptr()), "acr_crf_norm")

    def apply(self, q, out=None):
        return self.lat.filter(q, pre=self.norm, post=self.norm, scale=self.compat, out=out)


def crf_inference(img, probs, t=10, scale_factor=1, labels=21, device="cuda"):
    """tool/imutils.py:345-362.  img (h, w, 3) uint8, probs (labels, h, w) -> Q (labels, h, w) float32 numpy array."""
    lib = L.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise L.AcrHipError("crf_inference needs a GPU (no CPU path in the product)")
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    n = h * w
    p = torch.as_tensor(np.ascontiguousarray(probs, dtype=np.float32)).reshape(labels, n).to(dev)
    unary = torch.empty_like(p)
    q = torch.empty_like(p)
    with torch.cuda.device(dev):
        L.check(lib.acr_crf_unary(L.ptr(p), L.ptr(unary), labels * n, 1e-5, L.stream_ptr()), "acr_crf_unary")
        kernels = [_Kernel(PermutohedralLattice(h, w, 3 / scale_factor, device=dev), 3),
                   _Kernel(PermutohedralLattice(h, w, 80 / scale_factor, rgb=img, srgb=13, device=dev), 10)]
        L.check(lib.acr_crf_update(L.ptr(unary), None, None, L.ptr(q), n, labels, L.stream_ptr()), "acr_crf_update")
        msg = [torch.empty_like(p), torch.empty_like(p)]
        for _ in range(t):
            for k, m in zip(kernels, msg):
                k.apply(q, out=m)
            L.check(lib.acr_crf_update(L.ptr(unary), L.ptr(msg[0]), L.ptr(msg[1]), L.ptr(q), n, labels, L.stream_ptr()), "acr_crf_update")
    return q.reshape(labels, h, w).cpu().numpy()


def crf_with_alpha(cam_dict, alpha, orig_img, device="cuda"):
    """infer_cam.py:27-40: {class: cam (h, w)} -> {0: background, class + 1: ...} after the CRF, background score
    (1 - max_c cam)^alpha."""
    classes = list(cam_dict.keys())
    cams = np.stack([cam_dict[c] for c in classes], axis=0)
    background = np.power(1 - cams.max(axis=0, keepdims=True), alpha)
    scores = np.concatenate((background, cams), axis=0)
    refined = crf_inference(orig_img, scores, labels=scores.shape[0], device=device)
    out = {0: refined[0]}
    for i, c in enumerate(classes):
        out[c + 1] = refined[i + 1]
    return out
