"""acr_wsss_amd -- MI355X-native (gfx950) implementation of the ACR_WSSS hot path.

Python host surface mirrors the reference (``DPT.ACR.ACR`` with forward_mirror / forward_cls / forward_cam /
getam; the inline ACR loss of train_acr.py as ``acr_loss``; ``infer_cam``), the arithmetic of the path runs in
hand-written HIP kernels behind the C ABI of include/acr_hip.h.
"""
__version__ = "0.1.0"
