"""Training-step pieces of train_acr.py / train_acr_coco.py on the HIP path.

  acr_loss       <- the inline loss block train_acr.py:140-168 (COCO twin train_acr_coco.py:137-165)
  PolyOptimizer  <- tool/torchutils.py:10-31 (including its positional-argument quirk)
  train_step     <- train_acr.py:135-174 (one iteration, minus data loading and logging)
"""
import os

import torch
import torch.nn.functional as F

from . import ops

_set_versions = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)


def acr_loss(cls_list, attn_list, label, p, alpha):
    """loss = MLSM(x1,y) + MLSM(x2,y) + alpha*cls_align + alpha*aff_align   (train_acr.py:160-168).

    ``attn_list`` is what ``ACR.forward_mirror`` returned (``[attn1, attn2]``; the fused two-view stack is
    taken from its ``.stacked`` attribute when present) or the (2B,L,T,T) stack itself.  The reference's
    3*p in-place block flips + two F.l1_loss are one HIP kernel each way (include/acr_hip.h
    acr_consistency_fwd/bwd); the view-2 maps are NOT mutated.  Returns (loss, dict of the four terms)."""
    stacked = getattr(attn_list, "stacked", None)
    if stacked is None and torch.is_tensor(attn_list):
        stacked = attn_list
    if stacked is not None:
        cls_align, aff_align = ops.consistency(stacked, p)
    else:
        cls_align, aff_align = ops.consistency(attn_list[0], p, attn_list[1])
    x1, x2 = cls_list[0], cls_list[1]
    cls_loss_1 = ops.mlsm_loss(x1.float(), label)
    cls_loss_2 = ops.mlsm_loss(x2.float(), label)
    loss = cls_loss_1 + cls_loss_2 + cls_align * alpha + aff_align * alpha
    return loss, dict(cls_loss_1=cls_loss_1, cls_loss_2=cls_loss_2, cls_align=cls_align, aff_align=aff_align, loss=loss)


class PolyOptimizer(torch.optim.SGD):
    """SGD with lr = lr0 * (1 - step/max_step)^0.9  (tool/torchutils.py:10-31).

    Quirk kept on purpose: the reference calls ``SGD.__init__(params, lr, weight_decay)``, so its
    ``weight_decay`` argument lands in SGD's *momentum* slot -- effective momentum = wt_dec (5e-4), effective
    weight decay = 0; ``momentum=0.9`` is only the exponent of the poly schedule."""

    def __init__(self, params, lr, weight_decay, max_step, momentum=0.9, **sgd_kwargs):
        super().__init__(params, lr, weight_decay, **sgd_kwargs)
        self.global_step = 0
        self.max_step = max_step
        self.momentum = momentum
        self._initial_lr = [group["lr"] for group in self.param_groups]
        self.lr_scale = 1

    def step(self, closure=None):
        if self.global_step < self.max_step:
            lr_mult = (1 - self.global_step / self.max_step) ** self.momentum
            for i in range(len(self.param_groups)):
                self.param_groups[i]["lr"] = self._initial_lr[i] * lr_mult * self.lr_scale
        if closure is None and self._fused_ok():
            self._fused_step()
        else:
            super().step(closure)
        self.global_step += 1

    # ---- all-fp32 CUDA model: momentum + update of every parameter in ONE launch (acr_sgd_step_f32), bit-identical to the
    # stock multi-tensor path (13 launches, 0.95 ms per step at this model's 86 M parameters) ----
    fused = True

    def _fused_ok(self):
        if not self.fused or len(self.param_groups) != 1 or _set_versions is None:
            return False
        g = self.param_groups[0]
        if (g.get("dampening", 0) != 0 or g.get("nesterov", False) or g.get("weight_decay", 0) != 0 or g.get("maximize", False)
                or g.get("momentum", 0) == 0):
            return False
        ps = g["params"]
        return bool(ps) and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
                                and (p.grad is None or (p.grad.dtype == torch.float32 and p.grad.is_contiguous() and not p.grad.is_sparse))
                                for p in ps)

    @torch.no_grad()
    def _fused_step(self):
        import numpy as np
        from . import _lib as L
        lib = L.load()
        ps = self.param_groups[0]["params"]
        dev = ps[0].device
        sizes = tuple(p.numel() for p in ps)                 # the chunk table depends on every tensor's size, not just their count
        if getattr(self, "_f_sizes", None) != sizes:
            chunk = lib.acr_sgd_chunk_elems()
            bt, bc = [], []
            for i, p in enumerate(ps):
                nch = (p.numel() + chunk - 1) // chunk
                bt.append(np.full(nch, i, dtype=np.int32))
                bc.append(np.arange(nch, dtype=np.int32))
            self._f_bt = torch.from_numpy(np.concatenate(bt)).to(dev)
            self._f_bc = torch.from_numpy(np.concatenate(bc)).to(dev)
            self._f_host = torch.zeros((len(ps), 5), dtype=torch.int64).pin_memory()
            self._f_tab = torch.zeros((len(ps), 5), dtype=torch.int64, device=dev)
            self._f_evt = None
            self._f_sizes = sizes
        if self._f_evt is not None:
            self._f_evt.synchronize()                      # the previous step's copy of the pinned table (long done)
        moms = []
        for p in ps:                                       # every column from the live tensors: nothing stale survives a
            if p.grad is None:                             # load_state_dict or a re-allocated gradient
                moms.append(0)                             # torch.optim.SGD skips gradient-less parameters entirely: no state
                continue                                   # entry for them (the kernel skips a zero gradient pointer)
            st = self.state[p]
            buf = st.get("momentum_buffer")
            if buf is None:
                buf = st["momentum_buffer"] = torch.zeros_like(p)      # mu * 0 + g == torch's first-step clone(g)
            moms.append(buf.data_ptr())
        hn = self._f_host.numpy()
        hn[:, 0] = [0 if p.grad is None else p.grad.data_ptr() for p in ps]
        hn[:, 1] = [p.data_ptr() for p in ps]
        hn[:, 2] = moms
        hn[:, 4] = sizes
        grp = self.param_groups[0]
        with torch.cuda.device(dev):
            self._f_tab.copy_(self._f_host, non_blocking=True)
            self._f_evt = torch.cuda.Event()
            self._f_evt.record()
            L.check(lib.acr_sgd_step_f32(L.ptr(self._f_tab), L.ptr(self._f_bt), L.ptr(self._f_bc), self._f_bt.numel(), float(grp["lr"]),
                                         float(grp["momentum"]), L.stream_ptr()), "acr_sgd_step_f32")
        # the kernel changed the parameters behind autograd's back: move their version counters on, as the stock in-place
        # update would have -- caches keyed on them (the weight transposes) must see a stale copy as stale
        _set_versions(ps, [p._version + 1 for p in ps])


TABLE_CHECK = True      # A/B (timing only): per-step refresh of the pointer table


class MasterWeights:
    """bf16 model, fp32 master weights: the MI355X training precision of this build.

    The module's parameters (and therefore every activation, gradient and gradient all-reduce) are bf16; the
    optimizer -- the reference's PolyOptimizer, quirk included -- runs on fp32 master copies, which are cast
    back into the module after every step.  Compared with autocast this removes the fp32<->bf16 cast kernels
    around every Linear / norm and halves the elementwise and RCCL bytes.  ``optimizer_factory(params)`` must
    build the optimizer over the list of fp32 masters it is given."""

    def __init__(self, model, optimizer_factory):
        self.model_params = [p for p in model.parameters() if p.requires_grad]
        self.masters = [p.detach().float().clone() for p in self.model_params]
        for m in self.masters:
            m.requires_grad_(True)
        for p in self.model_params:
            p.data = p.data.to(torch.bfloat16)
        for b in model.buffers():
            if b.is_floating_point():
                b.data = b.data.to(torch.bfloat16)
        self.optimizer = optimizer_factory(self.masters)
        self.param_groups = self.optimizer.param_groups
        # (in, out) copies of the Linear weights for the input-gradient GEMMs: refreshed in one launch after every step
        from . import ops as _ops
        self.weight_t = _ops.WeightTransposes(model.modules()) if (self.masters and self.masters[0].is_cuda) else None
        if self.weight_t is not None:
            self.weight_t.refresh()

    @property
    def global_step(self):
        return self.optimizer.global_step

    def zero_grad(self, set_to_none=True):
        for p in self.model_params:
            p.grad = None

    # ---- checkpoint / resume: model.state_dict() alone holds only the bf16 ROUNDINGS of the weights ----
    def state_dict(self):
        """fp32 masters + optimizer state (momentum buffers, lr schedule position): everything a resume needs on top of
        (or instead of) ``model.state_dict()``."""
        import copy
        # a SNAPSHOT: optimizer.state_dict() hands out the live momentum buffers, which the next step updates in place
        return {"masters": [m.detach().clone() for m in self.masters], "optimizer": copy.deepcopy(self.optimizer.state_dict()),
                "global_step": self.optimizer.global_step}

    @torch.no_grad()
    def load_state_dict(self, sd):
        assert len(sd["masters"]) == len(self.masters), "checkpoint has %d masters, model %d" % (len(sd["masters"]), len(self.masters))
        for m, src in zip(self.masters, sd["masters"]):
            m.copy_(src.to(m.device))
        self.optimizer.load_state_dict(sd["optimizer"])     # replaces the momentum buffers: the pointer table is stale now
        self.optimizer.global_step = int(sd["global_step"])
        self._sgd_tab = None
        torch._foreach_copy_([p for p in self.model_params], self.masters)      # bf16 working copies <- masters
        if self.weight_t is not None:
            self.weight_t.refresh()

    @torch.no_grad()
    def step(self):
        if self._fused_ok():
            return self._fused_step()
        live = [(m, p) for m, p in zip(self.masters, self.model_params) if p.grad is not None]
        for m, _ in live:
            if m.grad is None:
                m.grad = torch.empty_like(m)
        for m, p in zip(self.masters, self.model_params):
            if p.grad is None:
                m.grad = None
        if live:
            torch._foreach_copy_([m.grad for m, _ in live], [p.grad for _, p in live])
        self.optimizer.step()
        torch._foreach_copy_([p for p in self.model_params], self.masters)
        if self.weight_t is not None:
            self.weight_t.refresh()

    # ---- fused path: one HIP launch updates momentum, fp32 master and bf16 working copy of every parameter ----
    fused = True

    def _fused_ok(self):
        if not (self.fused and isinstance(self.optimizer, PolyOptimizer) and self.masters and self.masters[0].is_cuda):
            return False
        g = self.optimizer.param_groups
        return (len(g) == 1 and g[0].get("dampening", 0) == 0 and not g[0].get("nesterov", False)
                and g[0].get("weight_decay", 0) == 0 and not g[0].get("maximize", False)
                and all(p.grad is None or (p.grad.dtype == torch.bfloat16 and p.grad.is_contiguous()) for p in self.model_params))

    def _fused_step(self):
        import numpy as np
        from . import _lib as L
        lib = L.load()
        opt = self.optimizer
        dev = self.masters[0].device
        if getattr(self, "_sgd_tab", None) is None:
            chunk = lib.acr_sgd_chunk_elems()
            bt, bc = [], []
            for i, m in enumerate(self.masters):
                nch = (m.numel() + chunk - 1) // chunk
                bt.append(np.full(nch, i, dtype=np.int32))
                bc.append(np.arange(nch, dtype=np.int32))
            self._sgd_bt = torch.from_numpy(np.concatenate(bt)).to(dev)
            self._sgd_bc = torch.from_numpy(np.concatenate(bc)).to(dev)
            self._sgd_host = torch.zeros((len(self.masters), 5), dtype=torch.int64).pin_memory()
            self._sgd_tab = torch.zeros((len(self.masters), 5), dtype=torch.int64, device=dev)
            for i, (m, p) in enumerate(zip(self.masters, self.model_params)):
                st = opt.state[m]
                if "momentum_buffer" not in st or st["momentum_buffer"] is None:
                    st["momentum_buffer"] = torch.zeros_like(m)        # mu*0 + g == torch's first-step clone(g)
                self._sgd_host[i, 1] = m.data_ptr()
                self._sgd_host[i, 2] = st["momentum_buffer"].data_ptr()
                self._sgd_host[i, 4] = m.numel()
        host = self._sgd_host
        # the pinned table is re-filled every step: the previous step's copy must have been consumed
        if getattr(self, "_sgd_evt", None) is not None:
            self._sgd_evt.synchronize()
        hn = host.numpy()
        hn[:, 0] = [0 if p.grad is None else p.grad.data_ptr() for p in self.model_params]
        hn[:, 3] = [p.data_ptr() for p in self.model_params]
        # masters / momentum buffers may have been replaced since the table was built (optimizer.load_state_dict on
        # resume does exactly that): every column is refreshed from the live tensors, so a stale pointer cannot survive
        if TABLE_CHECK:
            # one pass over the state dict's items (no tensor hashing): ~0.1 ms for 315 tensors
            mom_of = {id(p): st.get("momentum_buffer") for p, st in opt.state.items()}
            moms = []
            for m in self.masters:
                buf = mom_of.get(id(m))
                if buf is None:
                    buf = opt.state[m]["momentum_buffer"] = torch.zeros_like(m)
                moms.append(buf)
            hn[:, 1] = [m.data_ptr() for m in self.masters]
            hn[:, 2] = [b.data_ptr() for b in moms]
        self._sgd_tab.copy_(host, non_blocking=True)
        self._sgd_evt = torch.cuda.Event()
        self._sgd_evt.record()
        # PolyOptimizer.step's schedule (tool/torchutils.py:23-27)
        if opt.global_step < opt.max_step:
            lr_mult = (1 - opt.global_step / opt.max_step) ** opt.momentum
            for i in range(len(opt.param_groups)):
                opt.param_groups[i]["lr"] = opt._initial_lr[i] * lr_mult * opt.lr_scale
        grp = opt.param_groups[0]
        L.check(lib.acr_sgd_step_bf16(L.ptr(self._sgd_tab), L.ptr(self._sgd_bt), L.ptr(self._sgd_bc), self._sgd_bt.numel(),
                                      float(grp["lr"]), float(grp["momentum"]), L.stream_ptr()), "acr_sgd_step_bf16")
        opt.global_step += 1
        if self.weight_t is not None:
            self.weight_t.refresh()


def train_step(model, optimizer, img, label, alpha, grad_sync=None, amp_dtype=None):
    """One iteration of train_acr.py:127-174: view 2 = h-flip, forward_mirror, ACR loss, backward, SGD step.

    ``grad_sync`` (acr_wsss_amd.dp.GradSync) all-reduces gradients over RCCL while backward is still running;
    the reference wraps the model in DDP but calls ``model.module.forward_mirror`` so its reducer never fires
     -- this implements the intended data-parallel semantics.  Returns (loss tensor, terms)."""
    p = img.shape[2] // 16
    img2 = img.flip(-1)                                   # RandomHorizontalFlip(p=1), train_acr.py:135
    optimizer.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=amp_dtype, enabled=amp_dtype is not None):
        cls_list, attn_list = model.forward_mirror(img, img2)
    loss, terms = acr_loss(cls_list, attn_list, label, p, alpha)
    if grad_sync is not None:
        grad_sync.prepare()
    loss.backward()
    if grad_sync is not None:
        grad_sync.finish()
    optimizer.step()
    refresh_weight_transposes(model)
    return loss, terms


def refresh_weight_transposes(model):
    """fp32 mode: the (in, out) copies of the block Linears' weights that the input-gradient GEMMs read (one launch; created on
    first use).  Optional -- a loop that does not call this (the reference's own, train_acr.py:135-174) simply runs those GEMMs
    on W as stored, since a copy is only ever used while its weight's version matches.  The bf16 mode's MasterWeights keeps its
    own set."""
    from . import ops as _ops
    _ops.invalidate_weight_images(model)                  # split-product images are keyed like the transposes: same blind spot for .data writes
    if getattr(model, "math", "f32") == "f32_split" and _ops.X3_IMAGES:
        # split products multiply by the IMAGES of W and W^T (ops.weight_image): no fp32 copies to keep; the images of every block
        # Linear for the coming step are made here in one launch (96 image launches of a few microseconds per step otherwise)
        lins = getattr(model, "_acr_x3_linears", None)
        if lins is None:
            from .backbone import Attention, Mlp
            lins = []
            for m in model.modules():
                if isinstance(m, Attention):
                    lins += [m.qkv, m.proj]
                elif isinstance(m, Mlp):
                    lins += [m.fc1, m.fc2]
            object.__setattr__(model, "_acr_x3_linears", lins)
        if lins and lins[0].weight.is_cuda and lins[0].weight.dtype == torch.float32:
            _ops.prebuild_weight_images(lins)
        return
    wt = getattr(model, "_acr_wt_f32", None)
    if wt is None:
        p0 = next(model.parameters(), None)
        if p0 is None or not p0.is_cuda or p0.dtype != torch.float32 or not _ops.F32_WT or not _ops.F32_HIP_LINEAR:
            return
        wt = _ops.WeightTransposes(model.modules(), dtype=torch.float32)
        object.__setattr__(model, "_acr_wt_f32", wt)       # not a submodule / parameter: plain attribute
    wt.refresh()
