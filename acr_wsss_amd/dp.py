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
    __slots__ = ("flat", "params", "views", "pending", "work", "index")


class GradSync:
    """Bucketed gradient all-reduce overlapped with backward.

    Collective sequences are identical on every rank BY CONSTRUCTION, whatever each rank's autograd graph does:
      * buckets are exchanged in index order only (a bucket whose gradients are complete waits for its predecessors; backward
        fills them in that order anyway), each exactly once, the stragglers from ``finish()`` in the same order;
      * which parameters are not waited for next step (no gradient this step) and which PARAMETERS produced a gradient after
        their bucket had left are agreed collectively: one tiny host-side all-reduce per step (a gloo side group beside RCCL:
        no device synchronisation), so a parameter that receives a gradient on some ranks only costs overlap, never a hang.
    Values are the mean over ranks of what each rank produced THIS step, a missing gradient counting as zero: the slot of a
    parameter without a gradient is zeroed before its bucket leaves (it would otherwise still hold the previous step's
    average), and late gradients are exchanged on their own and ADDED to the averaged slot (re-averaging the whole bucket
    would average already-averaged values with raw ones when the ranks disagree on who was late).
    ``strict=True`` (or ACR_DP_STRICT=1) turns any disagreement between ranks into an error instead."""

    def __init__(self, params, process_group=None, bucket_mb=64, strict=None, always_reduce=False, record_timeline=False,
                 late_params=None, static_graph=False, recheck_every=0):
        """``late_params``: parameters whose gradients are known to arrive at the very end of backward (the stem convolutions
        below the last ResNet stage: ACR.late_gradient_parameters()); they get the LAST bucket(s) to themselves, so that no
        other gradient waits for them (round 4's timeline: 86 MB became launchable 0.1 ms before backward ended because 6 MB of
        early-stage weights shared their buckets).
        ``static_graph``: the model's autograd graph is the same every step (true for ACR training).  The per-step host-side
        agreement (one gloo MAX all-reduce of 3 flags per parameter: a cross-rank host rendezvous) is then only run until ONE
        step has passed in which every rank agreed, nothing was late and the set of gradient-less parameters did not change;
        from then on each rank checks its OWN step against that learned pattern and raises if it deviates (a collective
        decided by one rank alone would hang the others).  ``recheck_every`` = N > 0 runs the collective agreement every N-th
        step anyway (all ranks count steps alike) as a debug check."""
        import os
        # record_timeline: a device event at prepare() (backward about to start), at every bucket launch and at finish()
        # (backward fully issued), on the launch stream -> timeline() says when each bucket became launchable inside backward
        self._tl = [] if (record_timeline and torch.cuda.is_available()) else None
        self._tl_steps = []
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [p for p in params if p.requires_grad]
        self._params = params
        self._pidx = {p: i for i, p in enumerate(params)}
        cap = int(bucket_mb * (1 << 20))
        late_ids = set(id(p) for p in (late_params or ()))
        groups = []
        # ~ the order in which backward produces gradients: reverse parameter order, the declared-late ones behind everything
        for part in ([p for p in reversed(params) if id(p) not in late_ids], [p for p in reversed(params) if id(p) in late_ids]):
            cur, cur_bytes = [], 0
            for p in part:
                nbytes = p.numel() * p.element_size()
                if cur and (cur_bytes + nbytes > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                    groups.append(cur)
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += nbytes
            if cur:
                groups.append(cur)
        self.static_graph, self.recheck_every = bool(static_graph), int(recheck_every)
        self._static_ok = False                           # True once a fully agreeing, pattern-stable step has been seen
        self._static_nograd = None                        # the learned set of gradient-less parameters (indices)
        self.buckets, self._of, self._view = [], {}, {}
        for grp in groups:
            b = _Bucket()
            b.params = grp
            # every slot starts on a 64-element (>= 128-byte) boundary: gradient views stay 16-byte aligned for the
            # vectorised consumers (fused optimizer step); the padding is zero and rides along in the all-reduce
            pad = lambda n: (n + 63) // 64 * 64
            # (+ one 64-element slot behind the LAST bucket's gradients: the static regime's "this rank deviated" flag rides
            # along in that bucket's all-reduce, see finish())
            extra = 64 if grp is groups[-1] else 0
            b.flat = torch.zeros(sum(pad(p.numel()) for p in grp) + extra, dtype=grp[0].dtype, device=grp[0].device)
            b.views, off = [], 0
            for p in grp:
                b.views.append(b.flat[off:off + p.numel()].view_as(p))
                off += pad(p.numel())
                self._of[p] = b
                self._view[p] = b.views[-1]
                p.register_post_accumulate_grad_hook(self._hook)
            b.pending, b.work = 0, None
            b.index = len(self.buckets)
            self.buckets.append(b)
        self._flag = self.buckets[-1].flat[-64:] if self.buckets else None      # see finish(): deviation flag of the static regime
        self._flag_pending = None                         # (pinned host copy, event) of the previous static step's reduced flag
        self._deviation = None                            # what THIS rank saw when it last deviated (for the message)
        self._prev_unused = set()
        self._armed = False
        self._late = []
        self._next = 0                                    # index of the next bucket to exchange (canonical order)
        # Parameters that got no gradient on ANY rank in the previous step (the reference model has 9 such tensors:
        # bkg_token, norm.*, head.*, scratch.*) are not waited for: otherwise the bucket they share -- the FIRST
        # one backward fills, the one with the most overlap to gain -- could only be launched from finish(), after backward.
        self._unused = set()
        self.launch_log = []                              # (bucket index, "backward" | "finish") of the last step
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self.backend = backend
        self._avg = backend == "nccl"                     # RCCL has a native AVG; gloo does not
        self.strict = (os.environ.get("ACR_DP_STRICT") == "1") if strict is None else bool(strict)
        # host-side agreement channel: the data group itself when it is gloo, else a gloo group over the same ranks
        # always_reduce: issue the collectives even in a one-rank group (RCCL smoke tests: the all-reduce of one rank is the
        # identity, but it goes through the same RCCL launch path)
        self._collective = self.world > 1 or (always_reduce and dist.is_initialized())
        self._agree = self._collective
        self._side = process_group                        # None = the default group
        if self._agree and backend != "gloo":
            self._side = dist.new_group(
                ranks=dist.get_process_group_ranks(process_group) if process_group is not None else None, backend="gloo")
        self.stats = {"steps": 0, "bucket_launches_in_backward": 0, "bucket_launches_in_finish": 0, "late_reexchanges": 0,
                      "rank_disagreements": 0, "agreement_exchanges": 0, "static_deviations": 0}

    def describe(self):
        """What bench.py reports about the exchange (sizes in MB per bucket, launch counters so far)."""
        return {"buckets": len(self.buckets),
                "bucket_mb": [round(b.flat.numel() * b.flat.element_size() / 2 ** 20, 1) for b in self.buckets],
                "unused_parameters": len(self._unused), **self.stats}

    def prepare(self):
        """Call after zero_grad and before backward.  Gradients are left to autograd (``.grad = None``: the engine then
        *moves* each gradient into ``.grad`` instead of launching an add kernel per parameter); when the last gradient of
        a bucket has arrived they are copied into the bucket with ONE multi-tensor launch and ``.grad`` is re-pointed at
        the bucket views.  Slots of parameters without a gradient are zeroed when the bucket leaves."""
        self._late = []
        self._prev_unused = set(self._unused)             # snapshot: the hooks discard entries from _unused during backward
        if self._tl is not None:
            self._tl = [("start", -1, 0, self._event())]
        for b in self.buckets:
            b.pending, b.work = sum(1 for p in b.params if p not in self._unused), None
            for p in b.params:
                p.grad = None
        self.launch_log = []
        self._next = 0
        self._armed = True

    def _event(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def timeline(self):
        """Per recorded step (record_timeline=True): backward's duration on the device and, per bucket, its size and when its
        launch was issued relative to the END of backward (ms, negative = that long before the last gradient kernel finished;
        synchronises).  What one GPU can say about overlap: a bucket's all-reduce can hide under whatever backward work is
        still to come after its launch."""
        if self._tl is None:
            return []
        torch.cuda.synchronize()
        out = []
        for tl in self._tl_steps:
            start = next(e for k, _, _, e in tl if k == "start")
            end = next(e for k, _, _, e in tl if k == "end")
            out.append({"backward_ms": round(start.elapsed_time(end), 3),
                        "buckets": [{"bucket": i, "mb": round(nb / 2 ** 20, 2), "where": k,
                                     "ms_before_backward_end": round(e.elapsed_time(end), 3)} for k, i, nb, e in tl
                                    if k in ("backward", "finish")]})
        return out

    def _launch(self, b, where="backward"):
        self.launch_log.append((b.index, where))
        self.stats["bucket_launches_in_" + where] += 1
        if self._tl is not None:
            self._tl.append((where, b.index, b.flat.numel() * b.flat.element_size(), self._event()))
        live = [(v, p.grad) for p, v in zip(b.params, b.views) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if live:
            torch._foreach_copy_([v for v, _ in live], [g for _, g in live])
        # a slot nobody writes this step still holds the PREVIOUS step's average: it must enter the mean as zero
        dead = [v for p, v in zip(b.params, b.views) if p.grad is None]
        if dead:
            torch._foreach_zero_(dead)
        for p, v in zip(b.params, b.views):
            if p.grad is not None:
                p.grad = v
        if self._collective:
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            b.work = dist.all_reduce(b.flat, op=op, group=self.pg, async_op=True)
        else:
            b.work = False

    def _launch_ready(self):
        # static regime: the last bucket leaves from finish(), after this rank's deviation flag has been written behind its
        # gradients (it is the declared-late bucket -- its gradients land with the end of backward anyway)
        limit = len(self.buckets) - (1 if (self._agree and self.static_graph and self._static_ok) else 0)
        while self._next < limit and self.buckets[self._next].pending == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _hook(self, p):
        if not self._armed:
            return
        b = self._of[p]
        if p in self._unused:                             # it does receive a gradient this step after all
            self._unused.discard(p)
            if b.work is not None:                        # its bucket has already gone out without it (slot zeroed):
                self._late.append(p)                      # exchanged on its own from finish()
            return
        b.pending -= 1
        self._launch_ready()

    def _wait(self, b):
        if b.work:
            b.work.wait()
            if not self._avg:
                b.flat.div_(self.world)

    def exchange_only(self):
        """All buckets all-reduced back to back in index order, nothing else -- what the exchange costs on its own, with no
        backward to hide under and no compute competing for HBM, CUs or watts (bench.py's ``ms_allreduce_only``).  The bucket
        contents are whatever the last step left there: a diagnostic, not part of a training step."""
        if not self._collective:
            return
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        works = [dist.all_reduce(b.flat, op=op, group=self.pg, async_op=True) for b in self.buckets]
        for w in works:
            w.wait()

    def _check_static_flag(self):
        """Look at the deviation flag the previous static step all-reduced (see finish()).  Non-zero on every rank alike: leave
        the static regime (the agreeing protocol takes over from this step on and re-learns the pattern) -- or raise if
        ``strict``."""
        if self._flag_pending is None:
            return
        host, ev = self._flag_pending
        self._flag_pending = None
        if ev is not None:
            ev.synchronize()
        if float(host[0]) == 0.0:
            return
        self.stats["static_deviations"] += 1
        self._static_ok = False
        mine = self._deviation
        self._deviation = None
        msg = ("GradSync(static_graph=True): some rank's previous step deviated from the learned gradient pattern (this rank: %s); "
               "its late gradients of that one step were not exchanged"
               % ("%d late parameter(s), %d presence change(s)" % mine if mine else "no deviation"))
        if self.strict:
            raise RuntimeError(msg + " -- the graph is not static: construct GradSync with static_graph=False")
        import warnings
        warnings.warn(msg + "; back on the per-step agreement until the pattern is stable again")

    def finish(self):
        """Call after backward: exchange the buckets backward could not complete (in index order), agree with the other
        ranks on late parameters and on the parameters nobody produced a gradient for, wait for every collective, turn
        sums into means and exchange the late gradients."""
        if self._tl is not None:
            self._tl.append(("end", -1, 0, self._event()))
            self._tl_steps.append(self._tl)
            self._tl_steps = self._tl_steps[-8:]
        npar = len(self._params)
        mine = set(self._late)
        late = [p in mine for p in self._params]
        nograd = [p.grad is None for p in self._params]
        agree = self._agree
        static_step = False
        if agree and self.static_graph and self._static_ok:
            # The flag the PREVIOUS static step all-reduced (its copy landed in pinned memory a step ago: waiting for that event
            # costs nothing while the host runs less than a step ahead of the GPU).  Every rank reads the same reduced value, so
            # every rank leaves the static regime -- or raises, under ``strict`` -- on the same step, BEFORE this step's remaining
            # collectives are issued: nobody is left blocking in RCCL on a peer that raised alone.
            self._check_static_flag()
        if agree and self.static_graph and self._static_ok:
            # static graph, pattern learned: no host rendezvous.  This rank's step should match the pattern every rank agreed on;
            # whether it does is written behind the last bucket's gradients and all-reduced WITH them.  A deviating rank still
            # issues exactly the static sequence of collectives (all buckets in index order, no late exchange), so no rank can
            # hang; its late gradients of this one step are dropped, and the next step is back on the agreeing protocol.
            if self.recheck_every > 0 and (self.stats["steps"] + 1) % self.recheck_every == 0:
                pass                                      # debug: run the collective agreement on this step anyway
            else:
                mism = sum(1 for i, n in enumerate(nograd) if n != (i in self._static_nograd))
                self._flag.fill_(1.0 if (mine or mism) else 0.0)
                if mine or mism:
                    self._deviation = (len(mine), mism)
                    late = [False] * npar
                agree = False
                static_step = True
        while self._next < len(self.buckets):
            self._launch(self.buckets[self._next], "finish")
            self._next += 1
        prev_unused = self._prev_unused
        if agree:
            self.stats["agreement_exchanges"] += 1
            # one small host-side exchange, per parameter: "late" on ANY rank, "has a gradient" on ANY rank, "has none" on ANY
            flags = torch.tensor([float(x) for x in late] + [float(not x) for x in nograd] + [float(x) for x in nograd],
                                 dtype=torch.float32)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self._side)
            any_late = [bool(v) for v in flags[:npar].tolist()]
            any_grad = [bool(v) for v in flags[npar:2 * npar].tolist()]
            any_nograd = [bool(v) for v in flags[2 * npar:].tolist()]
            disagree = sum(1 for a, n in zip(any_grad, any_nograd) if a and n) + sum(1 for a, l in zip(any_late, late) if a != l)
            if disagree:
                self.stats["rank_disagreements"] += 1
                if self.strict:
                    raise RuntimeError("GradSync: ranks disagree on gradient presence for %d parameter(s) this step "
                                       "(data-dependent graph?); the exchange itself stays consistent -- unset ACR_DP_STRICT "
                                       "to run on" % disagree)
            late = any_late
            nograd = [not a for a in any_grad]
            if self.static_graph:
                now = set(i for i, n in enumerate(nograd) if n)
                # decided from all-reduced values only, so that every rank enters (and leaves) the static regime on the same step
                stable = (not any(any_late) and not any(a and n for a, n in zip(any_grad, any_nograd))
                          and now == set(self._pidx[p] for p in prev_unused))
                self._static_ok, self._static_nograd = stable, now
        for b in self.buckets:
            self._wait(b)
        if static_step:
            # every rank leaves the step with the LEARNED presence pattern, whatever its own graph did: a gradient this rank
            # alone produced (late, never exchanged) is dropped and a gradient it alone lacks is the bucket's average -- the
            # optimizers of all ranks then apply the same update and the replicas stay identical
            for b in self.buckets:
                for p, v in zip(b.params, b.views):
                    p.grad = None if self._pidx[p] in self._static_nograd else v
        if static_step and self._collective:
            # behind the last bucket's all-reduce on the current stream: the reduced flag (> 0 iff some rank deviated) goes to
            # pinned memory asynchronously; it is looked at in the next step's finish()
            host = torch.empty(1, dtype=torch.float32, pin_memory=self._flag.is_cuda)
            host.copy_(self._flag[:1].float(), non_blocking=True)
            ev = None
            if self._flag.is_cuda:
                ev = torch.cuda.Event()
                ev.record()
            self._flag_pending = (host, ev)
        if any(late):
            # Late gradients (a parameter thought unused produced one, on some rank, after its bucket had left with a zeroed
            # slot): ONE extra exchange of just those tensors -- own late gradient, or zero where it was not late here (then it
            # either rode in the bucket already or does not exist) -- and the mean is ADDED to the averaged slot.
            ps = [p for p, l in zip(self._params, late) if l]
            parts = [(p.grad if p in mine else torch.zeros_like(p)).reshape(-1) for p in ps]
            flat = torch.cat(parts)
            self.launch_log.append(("late", "finish"))
            self.stats["late_reexchanges"] += 1
            if self._collective:
                dist.all_reduce(flat, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.pg)
                if not self._avg:
                    flat.div_(self.world)
            off = 0
            for p in ps:
                v = self._view[p]
                v.add_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
                p.grad = v
        if agree:                                         # a gradient that exists on another rank only: its average is ours too
            for b in self.buckets:
                for p, v in zip(b.params, b.views):
                    if p.grad is None and not nograd[self._pidx[p]]:
                        p.grad = v
        # re-learned every step: a tensor that stops (or starts) receiving gradients costs one late exchange, once
        if static_step:                                   # (static regime: the learned pattern stands; a deviation shows next step)
            nograd = [i in self._static_nograd for i in range(npar)]
        self._unused = {p for p, n in zip(self._params, nograd) if n}
        self.stats["steps"] += 1
        self._armed = False


def broadcast_parameters(module, src=0, process_group=None, bucket_mb=64):
    """One-time parameter/buffer broadcast from rank ``src`` (what DDP's constructor does, train_acr.py:99), coalesced like
    DDP's: tensors of one dtype and device are packed into flat buffers of about ``bucket_mb`` and each buffer is ONE
    collective (7 of them for the 417 MB hybrid-base model instead of one per tensor)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return 0
    cap = int(bucket_mb * (1 << 20))
    groups, cur, cur_bytes = [], [], 0
    for t in list(module.parameters()) + list(module.buffers()):
        nbytes = t.numel() * t.element_size()
        if cur and (cur_bytes + nbytes > cap or t.dtype != cur[0].dtype or t.device != cur[0].device):
            groups.append(cur)
            cur, cur_bytes = [], 0
        cur.append(t)
        cur_bytes += nbytes
    if cur:
        groups.append(cur)
    with torch.no_grad():
        for grp in groups:
            flat = torch.cat([t.detach().reshape(-1) for t in grp])
            dist.broadcast(flat, src=src, group=process_group)
            outs, off = [], 0
            for t in grp:
                outs.append(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
            torch._foreach_copy_([t.detach() for t in grp], outs)
    return len(groups)
