"""GETAM / CAM generation -- the per-image body of the reference's infer_cam.py:128-249 on the HIP path.

Differences from the reference that do not change results:
  * ``scales`` is a real list (the reference's loop is ``for scale in [1]`` with the scale variable already
    threaded through all shape arithmetic, infer_cam.py:145-156,186; BASELINE.json names {0.5,1,1.5,2});
  * the backward per class stops at block ``start_layer`` (nothing below it is read by getam, DPT/ACR.py:207)
    and computes no weight gradients (``torch.autograd.grad`` w.r.t. the tokens entering that block);
  * patch-CAM / GETAM read-outs, the affinity product and both bilinear resizes run as HIP kernels and stay
    on the device; one D2H copy per image instead of one per class (infer_cam.py:188).
Rank r of ``world`` processes handles images r::world (embarrassingly parallel, no collective).
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .backbone import pass_graph


_STREAMS = {}


def _scale_streams(dev, n):
    pool = _STREAMS.setdefault(torch.device(dev), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool


CONCURRENT_SCALES = os.environ.get("ACR_INFER_STREAMS", "1") != "0"        # A/B: one stream per scale


def infer_cam_images(model, imgs, labels, out_hws, **kw):
    """CAMs of a batch of same-sized network inputs: ``launch_cam_images`` (every kernel and the device-to-host copies are
    enqueued) followed at once by its ``collect()``; see there for arguments and results."""
    return launch_cam_images(model, imgs, labels, out_hws, **kw)()


def launch_cam_images(model, imgs, labels, out_hws, start_layer=10, func="grad", aff=True, scales=(1,), truncate=True,
                      batch_flips=True, concurrent_scales=None):
    """Enqueue the CAM generation of a batch of same-sized network inputs and return ``collect``, a callable that waits for
    THIS batch's results only and returns them -- so a caller walking a list (infer_cam_list) can enqueue the next images
    before collecting the previous ones and the GPU never waits for the host between images.

    CAMs of a batch of same-sized network inputs.  imgs (B,3,h,w) normalised, on the GPU; labels (B,C) multi-hot;
    out_hws: B pairs (W,H) = (image height, image width) as infer_cam.py:138 names them (each image keeps its own output
    size).  Returns a list of B (cam_dict, patch_cam_dict): {class index: float32 (W,H) numpy array}, min-max
    normalised over the summed passes (:201-215).

    Samples never interact on this path (GroupNorm / LayerNorm / attention are per sample), so images -- and, with
    ``batch_flips``, the flipped and the plain pass of a scale -- share one forward, and one backward per class rank
    k serves the k-th positive class of every image: d(sum_i logit[i, c_i^k]) / d tokens is block-diagonal in i."""
    dev = imgs.device
    if concurrent_scales is None:
        concurrent_scales = CONCURRENT_SCALES
    B, _, h, w = imgs.shape
    C = labels.shape[1]
    lab_cpu = labels.detach().float().cpu()                # host copy first: no device round trip when the labels arrive on the host
    labels = lab_cpu.to(dev, non_blocking=True) if not labels.is_cuda else labels.float()
    classes = [[c for c in range(C) if float(lab_cpu[i, c]) > 1e-5] for i in range(B)]
    kmax = max((len(c) for c in classes), default=0)
    vit = model.pretrained.model
    cam_acc = [torch.zeros((len(classes[i]), out_hws[i][0], out_hws[i][1]), dtype=torch.float32, device=dev) for i in range(B)]
    patch_acc = [torch.zeros((C, out_hws[i][0], out_hws[i][1]), dtype=torch.float32, device=dev) for i in range(B)]
    old_trunc = model.truncate_at
    model.truncate_at = start_layer if truncate else None
    # the truncated backward differentiates w.r.t. the tokens entering block `start_layer` only: with the parameters
    # frozen for the duration of the call no Function computes (and throws away) weight or bias gradients
    frozen = [p for p in model.parameters() if p.requires_grad] if truncate else []
    for p in frozen:
        p.requires_grad_(False)
    # getam reads the attention state of the forward/backward pair; a train()-mode module drops it by default
    attns = [blk.attn for blk in vit.blocks]
    for a in attns:
        a.keep_state_in_training = True
    # Scales are independent until their maps are summed, and the small ones leave most of the chip idle (a batch-2 pass at
    # scale 0.5 launches 18-72 workgroups per GEMM): each input geometry runs on a stream of its own and only the final
    # resize-and-accumulate steps, a few tiny kernels, are issued on the caller's stream in the reference's order -- so the
    # sums are bit-identical to the one-stream order.
    main = torch.cuda.current_stream(dev)
    side = _scale_streams(dev, len(scales)) if (concurrent_scales and len(scales) > 1) else None
    if side is not None and hasattr(vit.patch_embed, "backbone"):
        vit.patch_embed.backbone.refresh_frozen(imgs.dtype)     # the shared standardised conv weights: before the fork
    by_shape, done = {}, []
    # 0/1 masks of the class logits each backward differentiates: rank k -> (samples, C), for a pass of B samples and for the
    # batched flip pair (2 B samples: sample index = fi * B + i).  tgt = (logits * mask).sum(): its backward is one multiply
    # (an advanced-indexing gather would come back as a sorting index_put)
    m1 = torch.zeros((max(kmax, 1), B, C), dtype=torch.float32)
    for k in range(kmax):
        for i in range(B):
            if k < len(classes[i]):
                m1[k, i, classes[i][k]] = 1.0
    mask1 = m1.to(dev, non_blocking=True)
    mask2 = torch.cat([mask1, mask1], dim=1)
    def flush(rec):
        patch_items, cam_items, ev = rec
        if ev is not None:
            main.wait_event(ev)
        for t, i, flipped, ph, pw in patch_items:
            # patch-token CAM: (1,N,C) -> (C,ph,pw) -> bilinear(align_corners=False) * label, un-flip, sum
            ops.bilinear_resize(t.reshape(ph, pw, C), out_hws[i], False, chan_mul=labels[i], hflip=flipped, out=patch_acc[i],
                                channels_last=True)
            if ev is not None:
                t.record_stream(main)
        for t, i, flipped, ph, pw in cam_items:
            ops.bilinear_resize(t.reshape(len(classes[i]), ph, pw), out_hws[i], True, hflip=flipped, out=cam_acc[i])
            if ev is not None:
                t.record_stream(main)

    try:
        # with side streams the largest geometry is issued first: its long kernels then run while the host is still issuing
        # the small passes, which fill the gaps (the accumulation order below is the list's order either way)
        order = sorted(range(len(scales)), key=lambda si: -scales[si]) if side is not None else range(len(scales))
        for si in order:
            scale = scales[si]
            hs, ws = int(h * scale), int(w * scale)
            st = main
            if side is not None:
                st = side[by_shape.setdefault((hs, ws), len(by_shape))]     # one stream per geometry (a PrefixGraph is per geometry)
                st.wait_stream(main)
            with torch.cuda.stream(st):
                base = F.interpolate(imgs, size=(hs, ws), mode="bilinear", align_corners=False)
                ph, pw = int((h * scale) // 16), int((w * scale) // 16)
                # the two passes of a scale (h-flipped first, then plain: infer_cam.py:147-153) are independent samples: run
                # them as ONE batch -- at small batches the pass is launch-bound, not GPU-bound
                passes = ((True, False),) if batch_flips else ((True,), (False,))
                for flips in passes:
                    inp = torch.cat([base.flip(-1) if f else base for f in flips], dim=0)       # sample index = fi * B + i
                    patch_items, cam_items = [], []
                    nf = len(flips)
                    masks = mask2 if nf == 2 else mask1
                    # the whole pass -- forward_cam + one class-logit backward / GETAM read-out per class rank -- as captured
                    # hipGraphs for this geometry (backbone.PassGraph), else launch by launch
                    pg = pass_graph(model, inp, start_layer, func) if truncate else None
                    if pg is not None:
                        pc, attn, rows = pg.run(model, inp, [masks[k] for k in range(kmax)])
                    else:
                        with torch.enable_grad():
                            cls_pred, _, attn, patch_cam = model.forward_cam(inp)
                            pc = patch_cam.detach().float()
                            # one backward per class rank k serves the k-th positive class of every sample of the pass; the
                            # GETAM rows of ALL samples come from one launch per layer (ACR.getam_all) and the affinity product
                            # of all samples and classes from one more
                            rows = []
                            for k in range(kmax):
                                tgt = (cls_pred.float() * masks[k]).sum()
                                if truncate:
                                    torch.autograd.grad(tgt, vit.trunc_input, retain_graph=True)
                                else:
                                    model.zero_grad()
                                    tgt.backward(retain_graph=True)
                                rows.append(model.getam_all(start_layer=start_layer, func=func))       # (nf * B, N)
                    for fi, flipped in enumerate(flips):
                        for i in range(B):
                            patch_items.append((pc[fi * B + i], i, flipped, ph, pw))
                    if kmax:
                        cams_all = torch.stack(rows, dim=1)                                          # (nf * B, kmax, N)
                        if aff:                                                                      # patch_aff @ cam (:164-165,183-184)
                            cams_all = ops.aff_refine_batch(attn.detach(), cams_all)
                        for fi, flipped in enumerate(flips):
                            for i in range(B):
                                if classes[i]:
                                    cam_items.append((cams_all[fi * B + i, :len(classes[i])], i, flipped, ph, pw))
                    rec = (patch_items, cam_items, st.record_event() if side is not None else None)
                    if side is None:
                        flush(rec)
                    else:
                        done.append((si, rec))
        for _, rec in sorted(done, key=lambda d: d[0]):      # stable: the passes of a scale keep their order
            flush(rec)
    finally:
        model.truncate_at = old_trunc
        for p in frozen:
            p.requires_grad_(True)
        for a in attns:
            del a.keep_state_in_training                  # back to the class default
    # min-max normalisation (:201-215) and the device-to-host copies are enqueued behind this batch's last kernel on the caller's
    # stream -- into pinned buffers, asynchronously -- and an event marks their end: collect() waits for that event only
    host = []
    for i in range(B):
        ca, pa = cam_acc[i], patch_acc[i]
        cmin, cmax = ca.amin((1, 2), keepdim=True), ca.amax((1, 2), keepdim=True)
        norm_cam = (ca - cmin) / (cmax - cmin + 1e-6)
        pa = pa[classes[i]] if classes[i] else pa[:0]      # only the positive classes' planes are returned (and copied)
        pmin, pmax = pa.amin((1, 2), keepdim=True), pa.amax((1, 2), keepdim=True)
        patch_norm = (pa - pmin) / (pmax - pmin + 1e-5)
        hc = torch.empty(norm_cam.shape, dtype=torch.float32, pin_memory=True)
        hp = torch.empty(patch_norm.shape, dtype=torch.float32, pin_memory=True)
        hc.copy_(norm_cam, non_blocking=True)
        hp.copy_(patch_norm, non_blocking=True)
        host.append((hc, hp))
    done_ev = torch.cuda.Event()
    done_ev.record(main)

    def collect():
        done_ev.synchronize()
        out = []
        for i in range(B):
            norm_cam, patch_norm = host[i][0].numpy().copy(), host[i][1].numpy().copy()
            out.append(({c: norm_cam[j] for j, c in enumerate(classes[i])}, {c: patch_norm[j] for j, c in enumerate(classes[i])}))
        return out
    return collect


def infer_cam_image(model, img, label, out_hw, start_layer=10, func="grad", aff=True, scales=(1,), truncate=True,
                    batch_flips=True, concurrent_scales=None):
    """CAMs of one image (the unit of infer_cam.py:123 ``chunker(..., 1)``): img (1,3,h,w), label (1,C), out_hw = (W,H).
    Returns (cam_dict, patch_cam_dict); see infer_cam_images."""
    assert img.shape[0] == 1, "one image; use infer_cam_images for a batch"
    return infer_cam_images(model, img, label, [out_hw], start_layer=start_layer, func=func, aff=aff, scales=scales,
                            truncate=truncate, batch_flips=batch_flips, concurrent_scales=concurrent_scales)[0]


def seeds_from_cam_dict(cam_dict, threshold, num_cls=21):
    """evaluation.py:27-33: background plane = threshold, argmax over (21,h,w) -> uint8 seed map."""
    h, w = next(iter(cam_dict.values())).shape
    tensor = np.zeros((num_cls, h, w), np.float32)
    for k, v in cam_dict.items():
        tensor[k + 1] = v
    tensor[0] = threshold
    return np.argmax(tensor, axis=0).astype(np.uint8)


def shard_indices(n, rank=0, world=1):
    """Indices of an n-item list that rank `rank` of `world` processes handles: rank, rank + world, ... (the image list is
    sharded embarrassingly; the reference iterates the whole list on every rank, infer_cam.py:119-123)."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    return list(range(rank, n, world))


def infer_cam_list(model, items, out_cam=None, rank=0, world=1, batch_size=8, out_crf=None, low_alpha=1, high_alpha=12, **kw):
    """Shard ``items`` -- an indexable of (name, img (1,3,h,w), label (1,C), (W,H)[, orig uint8 (W,H,3)]) -- over ranks and
    write ``<out_cam>/<name>.npy`` in the reference's wire format: a pickled {class: float32 (W,H)} dict
    (infer_cam.py:227-228, read back by evaluation.py:23-25).  Returns {name: cam_dict} of this rank.
    ``batch_size`` (default 8, round 6) groups consecutive images of this rank whose network inputs have the same size -- the
    reference resizes every image to crop x crop first (myTool.py:1364-1403), so batches always form on its pipeline; results per
    image are those of the one-image call (samples never interact: test_infer_cam_images_batch_matches_single_images).  A group
    ends at a shape change; if a batch does not fit in memory the batch size is halved for the rest of the list (down to 1 =
    the reference's ``chunker(img_list, 1)``, infer_cam.py:123).
    ``out_crf`` (infer_cam.py:68,218-225, defaults of --low_alpha / --high_alpha :72-73): also run the dense CRF on every
    cam_dict at both alphas, on the GPU (crf.crf_with_alpha), and write ``<out_crf>_<alpha>/<name>.npy``; needs the
    original image as the fifth item element."""
    dev = next(model.parameters()).device
    model.eval()
    results = {}
    mine = shard_indices(len(items), rank, world)

    def finish(grp, collect):
        for i, (cam_dict, _) in zip(grp, collect()):
            name = items[i][0]
            if out_cam is not None:
                os.makedirs(out_cam, exist_ok=True)
                np.save(os.path.join(out_cam, name + ".npy"), cam_dict)
            if out_crf is not None and cam_dict:
                if len(items[i]) < 5:
                    raise ValueError("out_crf needs the original uint8 image as items[i][4] (infer_cam.py:217)")
                from .crf import crf_with_alpha
                for alpha in (low_alpha, high_alpha):
                    folder = out_crf + ("_%s" % alpha)
                    os.makedirs(folder, exist_ok=True)
                    np.save(os.path.join(folder, name + ".npy"), crf_with_alpha(cam_dict, alpha, np.asarray(items[i][4]), device=dev))
            results[name] = cam_dict

    # one batch in flight behind the one being collected: the kernels of images i+1 are enqueued BEFORE the host blocks on the
    # results of images i (each batch's copies land in its own pinned buffers behind its own event)
    pending, pos = None, 0
    while pos < len(mine):
        grp = [mine[pos]]
        shape = items[mine[pos]][1].shape
        while len(grp) < batch_size and pos + len(grp) < len(mine) and items[mine[pos + len(grp)]][1].shape == shape:
            grp.append(mine[pos + len(grp)])
        pos += len(grp)
        try:
            imgs = torch.cat([items[i][1] for i in grp], dim=0).to(dev)
            labels = torch.cat([items[i][2] for i in grp], dim=0)
            collect = launch_cam_images(model, imgs, labels, [items[i][3] for i in grp], **kw)
        except torch.cuda.OutOfMemoryError:
            if len(grp) == 1:
                raise
            # memory bound: collect what is in flight (its buffers go back to the allocator), halve the batch, take this group again
            imgs = labels = None
            if pending is not None:
                finish(*pending)
                pending = None
            torch.cuda.empty_cache()
            batch_size = max(1, len(grp) // 2)
            pos -= len(grp)
            continue
        if pending is not None:
            finish(*pending)
        pending = (grp, collect)
    if pending is not None:
        finish(*pending)
    return results
