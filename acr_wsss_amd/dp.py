"""Data-parallel gradient exchange for the ACR training step: one process per GPU, RCCL over xGMI.

The reference wraps the model in DistributedDataParallel (train_acr.py:99) but calls
``model.module.forward_mirror`` (:138), which bypasses DDP.forward, so its reducer never arms and every rank
trains an independent replica.  This module implements the intended semantics: gradients are
averaged over ranks every step.

MI355X-first choices: gradients live in a few large flat buckets (after a bucket's last gradient has arrived, all of
them are moved in with one multi-tensor launch and ``.grad`` of every parameter becomes a view into the bucket), so the
all-reduce runs in place and the optimizer reads the reduced values without another copy; buckets are large (64 MB default: xGMI is
point-to-point, ring all-reduce is per-link bound, so few big collectives beat many small ones) and each is
launched from the autograd thread the moment its last gradient lands, overlapping with the rest of backward
(the ResNet stem's gradients arrive last and form the only exposed bucket).  Works unchanged on ``gloo`` (CPU
tests) and ``nccl`` (= RCCL on ROCm).
"""
import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("flat", "params", "views", "pending", "work", "index", "late")


class GradSync:
    def __init__(self, params, process_group=None, bucket_mb=64):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [p for p in params if p.requires_grad]
        cap = int(bucket_mb * (1 << 20))
        groups, cur, cur_bytes = [], [], 0
        for p in reversed(params):                         # ~ order in which backward produces gradients
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        self.buckets, self._of = [], {}
        for grp in groups:
            b = _Bucket()
            b.params = grp
            # every slot starts on a 64-element (>= 128-byte) boundary: gradient views stay 16-byte aligned for the
            # vectorised consumers (fused optimizer step); the padding is zero and rides along in the all-reduce
            pad = lambda n: (n + 63) // 64 * 64
            b.flat = torch.zeros(sum(pad(p.numel()) for p in grp), dtype=grp[0].dtype, device=grp[0].device)
            b.views, off = [], 0
            for p in grp:
                b.views.append(b.flat[off:off + p.numel()].view_as(p))
                off += pad(p.numel())
                self._of[p] = b
                p.register_post_accumulate_grad_hook(self._hook)
            b.pending, b.work, b.late = 0, None, False
            b.index = len(self.buckets)
            self.buckets.append(b)
        self._armed = False
        # Parameters that got no gradient in the previous step (the reference model has 9 such tensors: bkg_token, norm.*,
        # head.*, scratch.*) are not waited for: otherwise the bucket they share -- the FIRST one backward fills,
        # the one with the most overlap to gain -- could only be launched from finish(), after backward.
        self._unused = set()
        self.launch_log = []                              # (bucket index, "backward" | "finish") of the last step
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self._avg = backend == "nccl"                     # RCCL has a native AVG; gloo does not

    def prepare(self):
        """Call after zero_grad and before backward.  Gradients are left to autograd (``.grad = None``: the engine then
        *moves* each gradient into ``.grad`` instead of launching an add kernel per parameter); when the last gradient of
        a bucket has arrived they are copied into the bucket with ONE multi-tensor launch and ``.grad`` is re-pointed at
        the bucket views.  Slots of parameters that never receive a gradient stay zero (zeroed once, at construction)."""
        for b in self.buckets:
            b.pending, b.work, b.late = sum(1 for p in b.params if p not in self._unused), None, False
            for p in b.params:
                p.grad = None
        self.launch_log = []
        self._armed = True

    def _launch(self, b, where="backward"):
        self.launch_log.append((b.index, where))
        live = [(v, p.grad) for p, v in zip(b.params, b.views) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if live:
            torch._foreach_copy_([v for v, _ in live], [g for _, g in live])
        for p, v in zip(b.params, b.views):
            if p.grad is not None:
                p.grad = v
        if self.world > 1:
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            b.work = dist.all_reduce(b.flat, op=op, group=self.pg, async_op=True)
        else:
            b.work = False

    def _hook(self, p):
        if not self._armed:
            return
        b = self._of[p]
        if p in self._unused:                             # it does receive a gradient this step after all
            self._unused.discard(p)
            if b.work is not None:                        # its bucket has already gone out without it: redo it in finish()
                b.late = True
            return
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def finish(self):
        """Call after backward: launch buckets whose parameters got no gradient (the reference's 9 unused
        tensors), wait for every collective, and turn sums into means."""
        for b in self.buckets:
            if b.work is None:
                self._launch(b, "finish")
        for b in self.buckets:
            if b.work:
                b.work.wait()
                if not self._avg:
                    b.flat.div_(self.world)
            if b.late:                                    # a parameter thought unused produced a gradient after the launch:
                b.work = None                             # every rank holds the same averaged values, so averaging the
                self._launch(b, "finish")                 # bucket again only adds the late gradient's exchange
                if b.work:
                    b.work.wait()
                    if not self._avg:
                        b.flat.div_(self.world)
        # re-learned every step: a tensor that stops (or starts) receiving gradients costs one late exchange, once
        self._unused = {p for b in self.buckets for p in b.params if p.grad is None}
        self._armed = False


def broadcast_parameters(module, src=0, process_group=None):
    """One-time parameter/buffer broadcast from rank ``src`` (what DDP's constructor does, train_acr.py:99)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=process_group)
