#!/usr/bin/env python3
"""bench.py -- ACR training-step throughput (BASELINE.json metric) on N MI355X GPUs of one node.

The HEADLINE (`value`, `dtype: "f32_split"`) is measured on fp32 tensors end to end (train_acr.py:137, autocast(enabled=False))
with the model's matrix products evaluated as six bf16-MFMA terms of a three-way operand split (24 mantissa bits per operand,
fp32 accumulate: ACR(..., math="f32_split"); every fp32 parity test runs under it at the fp32 tolerances).  The
same invocation then runs the exact-fp32-MFMA arithmetic (sub-record `"f32"`, the previous rounds' headline) and the bf16
training mode of this build (bf16 tensors + bf16 MFMA, fp32 master weights / accumulate / softmax / loss; parity pinned by
tests/test_model_gpu.py::test_bf16_train_step_448; sub-record `"bf16"` -- never `value`).  ACR_BENCH_HEADLINE=f32 puts the
exact arithmetic back in front.

    python bench.py [--gpus N] [--steps K] [--warmup W]
        N = 1: runs in this process.  N > 1 without WORLD_SIZE in the environment: this process touches no GPU, starts
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py ...`
        as a CHILD, relays its one JSON line and exits with its return code.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W  (the driver's form: one rank per GPU, RCCL)

A "step" is one iteration of train_acr.py:127-174 on a synthetic batch already resident in HBM: h-flip view,
ViT-hybrid-base forward over both views, ACR loss, full backward, (RCCL gradient all-reduce,) PolyOptimizer
step.  Workload = BASELINE configs[1]: 448x448, batch 16 per GPU (weak scaling: configs[2] is 16 x 8).
Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_IMG_448 = 1.1307e12          # SURVEY 6 [probe]: 2 views, fwd+bwd, hybrid-base @448^2
# MI355X_MICROARCH.md: dense matrix peaks.  f32_split: six bf16 MFMAs per fp32-equivalent product -> bf16 peak / 6
PEAK_MFMA = {"f32": 157.3e12, "f32_split": 2.5e15 / 6, "bf16": 2.5e15}
PRECISION = {"f32": "fp32 end to end (reference precision, exact-fp32 MFMA)",
             "f32_split": "fp32 tensors end to end; every Linear / convolution (1x1, 3x3, the 7x7 stem one) / attention product evaluated on the bf16 MFMA as six exact "
                          "terms of a three-way operand split (24 mantissa bits per operand), fp32 accumulation -- as accurate against fp64 as the "
                          "fp32 FMA chain (every fp32 parity test runs under it at the same tolerances; adversarial-operand test <= 2x the exact "
                          "kernel's error); softmax, norms, loss, optimizer: exact fp32 as in f32",
             "bf16": "bf16 params/activations/grads + bf16 MFMA, fp32 master weights + fp32 accumulate/softmax/loss"}
# rocprofv3 kernel-trace of this bench + separate PMC passes, per dtype, recorded BY THE BUILDER on its own gpurun box and
# committed: the `in_step` blocks and `traffic` figures of the line are CONSTANTS read from this file, not measurements of the
# run that prints the line (only `launch_ms` / `achieved` / `frac` / `value` are measured live) -- the line says so itself
# (`roofline.profile_source`)
IN_STEP = next((p for p in (os.path.join(ROOT, "profiles", "r%02d_in_step_kernels.json" % r) for r in (6, 5, 4, 3)) if os.path.exists(p)),
               os.path.join(ROOT, "profiles", "r06_in_step_kernels.json"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU (BASELINE configs[1]: 16)")
    ap.add_argument("--size", type=int, default=448)
    ap.add_argument("--dtype", choices=["both", "f32", "f32_split", "bf16"], default=os.environ.get("ACR_BENCH_DTYPE", "both"),
                    help="both (default): three runs from one invocation -- f32_split (fp32 tensors, split products on the bf16 MFMA) is the "
                         "headline `value`, the exact-fp32 run rides along as the sub-record 'f32' and the bf16 training mode as 'bf16' "
                         "(ACR_BENCH_HEADLINE=f32 swaps the two fp32 arithmetics); f32 / f32_split / bf16: that run only (profiling)")
    ap.add_argument("--classes", type=int, default=20)
    ap.add_argument("--alpha", type=int, default=125)
    ap.add_argument("--amp", choices=["master", "autocast"], default="master",
                    help="bf16 mode: bf16 model + fp32 master weights (default) or torch.autocast")
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--stock-linear", action="store_true", help="block Linears on torch F.linear instead of the HIP GEMMs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--probe-only", action="store_true", help="run only the kernel roofline probe (for rocprofv3 --pmc passes)")
    ap.add_argument("--no-infer", action="store_true", help="skip the CAM-generation record (BASELINE configs[3])")
    ap.add_argument("--infer-only", action="store_true", help="run only the CAM-generation record (profiling)")
    return ap.parse_args()


def make_batch(batch, size, ncls, seed, dev):
    g = torch.Generator(device="cpu").manual_seed(1000 + seed)
    img = torch.randn(batch, 3, size, size, generator=g)
    label = (torch.rand(batch, ncls, generator=g) > 0.85).float()
    label[:, 0] = 1.0
    return img.to(dev), label.to(dev)


def time_kernel(fn, iters=10, warm=2):
    """Average duration (s) of fn()'s launches, HIP events on the stream the kernels are launched on."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def _mfma_rec(name, alg_flops, exec_flops, t, peak, extra=None):
    rec = {"bound": "mfma", "kernel": name, "achieved": round(alg_flops / t / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
           "frac": round(alg_flops / t / peak, 4), "launch_ms": round(t * 1e3, 4), "flops_per_launch": alg_flops,
           "executed_flops_per_launch": exec_flops, "executed_frac": round(exec_flops / t / peak, 4)}
    if extra:
        rec.update(extra)
    return rec


def roofline_probe(args, dev, dtype, live=None):
    """The hand-written kernels of the step, launched one by one through the C ABI at the bench geometry and timed with
    HIP events on the launch stream.  ALGORITHMIC work per launch:
      attention forward  : S = q k^T and O = P v             -> 2 products of 2*T*T*64 FLOP per (sample, head)
      attention backward : dP = dO v^T, dV, dK, dQ           -> 4 products (what autograd's bmm backward executes)
      Linear / dX / dW   : 2*M*N*K
    `executed_*` additionally counts what the flash-style kernels recompute (S in both backward sweeps, the head-mean
    pass of the forward).  When `live` is given (average ms of the same launches measured with HIP events INSIDE the timed
    steps, ops.KernelTimer) `achieved` / `frac` / `launch_ms` are the in-step values and the isolated replay is kept as
    `isolated_launch_ms`.  `frac` is ALGORITHMIC work / time / dense matrix peak of the dtype.  Returns the record of the
    kernel that is on top of this round's in-step rocprof profile (IN_STEP: the newest profiles/rNN_in_step_kernels.json) with the others
    under `kernels`."""
    from acr_wsss_amd import _lib as L, ops
    lib = L.load()
    B, H, T = 2 * args.batch, 12, (args.size // 16) ** 2 + 1
    dt = torch.float32 if dtype in ("f32", "f32_split") else torch.bfloat16
    math = 1 if dtype == "f32_split" else 0              # acr_math: the per-call arithmetic of the fp32 entry points
    peak = PEAK_MFMA[dtype]
    g = torch.Generator(device="cpu").manual_seed(0)
    qkv = torch.randn(B, T, 3 * H * 64, generator=g).to(dev).to(dt)
    d_o = torch.randn(B, T, H * 64, generator=g).to(dev).to(dt)
    gm = (torch.randn(B, T, ops.pad4(T), generator=g).to(dev) * 1e-3)[:, :, :T]
    o = torch.empty(B, T, H * 64, dtype=dt, device=dev)
    lse2 = torch.empty(B, H, T, dtype=torch.float32, device=dev)
    pm = torch.empty(B, T, T, dtype=torch.float32, device=dev)
    dqkv = torch.empty_like(qkv)
    d = ops._desc(B, H, T, dt, math=math)
    delta = torch.empty(lib.acr_attn_bwd_ws_floats(d) if dt == torch.float32 else B * H * T, dtype=torch.float32, device=dev)
    qp, kp, vp = ops._qkv_ptrs(qkv, H)
    dqp, dkp, dvp = ops._qkv_ptrs(dqkv, H)
    st = L.stream_ptr()

    scores_path = dt == torch.float32 and ops.ATTN_F32_SCORES   # what ops.AttnCoreFn launches in fp32: logits resident in HBM
    if scores_path:
        sres = torch.empty(lib.acr_attn_scores_floats(d), dtype=torch.float32, device=dev)

        def fwd():
            L.check(lib.acr_attn_fwd_scores(d, qp, kp, vp, L.ptr(o), L.ptr(lse2), L.ptr(sres), L.ptr(pm), T * T, T, st), "fwd")

        def bwd():
            L.check(lib.acr_attn_bwd_scores(d, qp, kp, vp, L.ptr(o), L.ptr(d_o), L.ptr(lse2), L.ptr(sres), L.ptr(gm), gm.stride(0),
                                            gm.stride(1), dqp, dkp, dvp, L.ptr(delta), st), "bwd")
    else:
        def fwd():
            L.check(lib.acr_attn_fwd(d, qp, kp, vp, L.ptr(o), L.ptr(lse2), L.ptr(pm), T * T, T, st), "fwd")

        def bwd():
            L.check(lib.acr_attn_bwd(d, qp, kp, vp, L.ptr(o), L.ptr(d_o), L.ptr(lse2), L.ptr(gm), gm.stride(0), gm.stride(1), dqp, dkp,
                                     dvp, L.ptr(delta), st), "bwd")

    t_fwd = time_kernel(fwd)
    t_bwd = time_kernel(bwd)
    unit = 2.0 * T * T * 64 * B * H                      # one T x T x 64 product over all (b, h)
    if scores_path:                                      # executed products: 2 forward, 5 backward (dP dQ | dP dV dK)
        sbytes = 4.0 * B * H * ((T + 31) // 32) ** 2 * 1024
        kernels = {
            "acr_attn_bwd": _mfma_rec("acr_attn_bwd_scores (row term stream + dq + dkdv, logits read from HBM)", 4 * unit, 5 * unit, t_bwd,
                                      peak, {"scores_bytes_read_per_launch": 3 * sbytes}),
            "acr_attn_fwd": _mfma_rec("acr_attn_fwd_scores (+ head-mean stream, logits written once)", 2 * unit, 2 * unit, t_fwd, peak,
                                      {"scores_bytes_written_per_launch": sbytes, "scores_bytes_read_per_launch": sbytes}),
        }
        del sres
    else:
        kernels = {
            "acr_attn_bwd": _mfma_rec("acr_attn_bwd (delta + dq + dkdv)", 4 * unit, 8 * unit, t_bwd, peak),
            "acr_attn_fwd": _mfma_rec("acr_attn_fwd (+ head-mean)", 2 * unit, 3 * unit, t_fwd, peak),
        }
    # K3 (HBM-bound): forward reads 2 stacks
    Lyr = 12
    a = torch.rand(B, Lyr, T, T, device=dev)
    ws = torch.empty(lib.acr_consistency_ws_floats(B // 2, Lyr, T), device=dev)
    out2 = torch.empty(2, device=dev)

    def k3():
        L.check(lib.acr_consistency_fwd(L.ptr(a[:B // 2]), L.ptr(a[B // 2:]), a.stride(0), B // 2, Lyr, T,
                                        args.size // 16, L.ptr(ws), L.ptr(out2), st), "k3")
    t_k3 = time_kernel(k3)
    k3_bytes = 4.0 * B * Lyr * T * T
    kernels["acr_consistency_fwd"] = {"bound": "hbm", "achieved": round(k3_bytes / t_k3 / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                      "frac": round(k3_bytes / t_k3 / 8e12, 4), "launch_ms": round(t_k3 * 1e3, 4),
                                      "bytes_per_launch": k3_bytes}
    del a, ws
    # the block Linears at the bench's token count: fc1's three GEMMs (forward, input gradient, weight gradient)
    Mtok, Nw, Kw = B * T, 3072, 768
    x = torch.randn(Mtok, Kw, generator=g).to(dev).to(dt)
    w = (torch.randn(Nw, Kw, generator=g) * Kw ** -0.5).to(dev).to(dt)
    bias = torch.zeros(Nw, device=dev, dtype=dt)
    dy = torch.randn(Mtok, Nw, generator=g).to(dev).to(dt)
    fl = 2.0 * Mtok * Nw * Kw
    if dtype == "bf16":
        y = torch.empty(Mtok, Nw, dtype=dt, device=dev)
        wt = w.t().contiguous()
        dx = torch.empty(Mtok, Kw, dtype=dt, device=dev)
        wsw = torch.empty(lib.acr_wgrad_ws_floats(Mtok, Nw, Kw), dtype=torch.float32, device=dev)
        dww = torch.empty(Nw, Kw, dtype=dt, device=dev)
        t = time_kernel(lambda: L.check(lib.acr_linear_bf16(L.ptr(x), Kw, L.ptr(w), Kw, L.ptr(bias), None, 0, L.ptr(y), Nw, Mtok, Nw, Kw, st), "lin"))
        kernels["acr_linear_bf16"] = _mfma_rec("gemm_nt_bf16_wide (fc1 forward, %dx%dx%d)" % (Mtok, Nw, Kw), fl, fl, t, peak)
        t = time_kernel(lambda: L.check(lib.acr_linear_bf16(L.ptr(dy), Nw, L.ptr(wt), Nw, None, None, 0, L.ptr(dx), Kw, Mtok, Kw, Nw, st), "dx"))
        kernels["acr_linear_bf16_dx"] = _mfma_rec("gemm_nt_bf16_wide (fc1 input gradient)", fl, fl, t, peak)
        t = time_kernel(lambda: L.check(lib.acr_wgrad_bf16(L.ptr(dy), Nw, L.ptr(x), Kw, Mtok, Nw, Kw, L.ptr(wsw), L.ptr(dww), st), "wgrad"))
        kernels["acr_wgrad_bf16"] = _mfma_rec("gemm_tn_bf16_wide + slab reduction (fc1 weight gradient)", fl, fl, t, peak)
    elif hasattr(ops, "gemm_f32_raw"):
        y = torch.empty(Mtok, Nw, dtype=dt, device=dev)
        dx = torch.empty(Mtok, Kw, dtype=dt, device=dev)
        dww = torch.empty(Nw, Kw, dtype=dt, device=dev)
        wt32 = w.t().contiguous()                          # what train.refresh_weight_transposes keeps per Linear
        x3 = bool(math) and getattr(ops, "X3_IMAGES", False)
        if x3:                                              # split products: what the step's Linears launch -- products on images made once
            xi, wi, dyi, wti = ops.x3_image(x), ops.x3_image(w), ops.x3_image(dy), ops.x3_image_t(w)
            probes = (
                ("acr_gemm_f32_nt", "acr_gemm_x3 NT on images (fc1 forward, %dx%dx%d)" % (Mtok, Nw, Kw), lambda: ops.gemm_x3("nt", xi, wi, y, Kw, bias=bias)),
                ("acr_gemm_f32_dx", "acr_gemm_x3 NT on the image of W^T (fc1 input gradient, %dx%dx%d; same shape as fc2 forward)" % (Mtok, Kw, Nw),
                 lambda: ops.gemm_x3("nt", dyi, wti, dx, Nw)),
                ("acr_gemm_f32_tn", "acr_gemm_x3 TN on the same images of dy and x (fc1 weight gradient)", lambda: ops.gemm_x3("tn", dyi, xi, dww, Mtok)))
        else:
            probes = (
                ("acr_gemm_f32_nt", "gemm_f32 NT (fc1 forward, %dx%dx%d)" % (Mtok, Nw, Kw), lambda: ops.gemm_f32_raw("nt", x, w, y, bias=bias, math=math)),
                ("acr_gemm_f32_dx", "gemm_f32 NT on the cached W^T (fc1 input gradient, %dx%dx%d; same shape as fc2 forward)" % (Mtok, Kw, Nw),
                 lambda: ops.gemm_f32_raw("nt", dy, wt32, dx, math=math)),
                ("acr_gemm_f32_tn", "gemm_f32 TN (fc1 weight gradient)", lambda: ops.gemm_f32_raw("tn", dy, x, dww, math=math)))
        for key, name, call in probes:
            t = time_kernel(call, iters=5)
            kernels[key] = _mfma_rec(name + (" [split products: 6 bf16 MFMAs per fp32 product]" if math else ""), fl, fl, t, peak)
        if x3:                                              # the streaming pass that makes an image: 4 bytes read + 6 written per element
            dbv = torch.empty(Nw, device=dev)
            t = time_kernel(lambda: ops.x3_image(dy, colsum=dbv), iters=5)
            ib = 10.0 * Mtok * Nw
            kernels["acr_x3_image"] = {"bound": "hbm", "kernel": "planes_tile (image of dy %dx%d + its column sums = the bias gradient)" % (Mtok, Nw),
                                       "achieved": round(ib / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ib / t / 8e12, 4),
                                       "launch_ms": round(t * 1e3, 4), "bytes_per_launch": ib}
            del xi, wi, dyi, wti
    if math:
        for r in kernels.values():
            if r.get("bound") == "mfma":
                r["peak_note"] = "fp32-equivalent FLOP against the dense bf16 MFMA peak / 6 (2.5 PF / 6 = 416.7 TF)"
    # durations measured inside the timed steps take precedence over the isolated replays above
    Mt = B * T
    live_keys = {"acr_attn_fwd": "attn_fwd %dx%dx%d" % (B, H, T), "acr_attn_bwd": "attn_bwd %dx%dx%d" % (B, H, T),
                 "acr_gemm_f32_nt": "gemm_f32_nt %dx%dx%d" % (Mt, 3072, 768), "acr_gemm_f32_dx": "gemm_f32_nt %dx%dx%d" % (Mt, 768, 3072),
                 "acr_gemm_f32_tn": "gemm_f32_tn %dx%dx%d" % (3072, 768, Mt), "acr_linear_bf16": "linear_bf16 %dx%dx%d" % (Mt, 3072, 768),
                 "acr_linear_bf16_dx": "linear_bf16 %dx%dx%d" % (Mt, 768, 3072), "acr_wgrad_bf16": "wgrad_bf16 %dx%dx%d" % (Mt, 3072, 768)}
    if math and getattr(ops, "X3_IMAGES", False):
        live_keys.update({"acr_gemm_f32_nt": "gemm_x3_nt %dx%dx%d" % (Mt, 3072, 768), "acr_gemm_f32_dx": "gemm_x3_nt %dx%dx%d" % (Mt, 768, 3072),
                          "acr_gemm_f32_tn": "gemm_x3_tn %dx%dx%d" % (3072, 768, Mt)})
    for k, lk in live_keys.items():
        if live and k in kernels and lk in live:
            r, t = kernels[k], live[lk] * 1e-3
            r["isolated_launch_ms"] = r["launch_ms"]
            r["launch_ms"] = round(t * 1e3, 4)
            r["achieved"] = round(r["flops_per_launch"] / t / 1e12, 2)
            r["frac"] = round(r["flops_per_launch"] / t / peak, 4)
            r["executed_frac"] = round(r["executed_flops_per_launch"] / t / peak, 4)
            r["timed"] = "HIP events around the launch inside the timed steps (average over the steps)"
    # in-step rocprofv3 durations of this round (scripts/profile_round.sh + scripts/make_in_step.py: kernel trace of this very
    # command) and the HBM traffic of the probe's launches from the two separate PMC passes; the record's head is the
    # kernel group on TOP of the in-step profile, not the one that looks worst
    in_step, traffic, top = {}, {}, None
    try:
        with open(IN_STEP) as f:
            rec = json.load(f).get(dtype, {})
        if (args.batch, args.size) == (16, 448):
            in_step, traffic, top = rec.get("kernels", {}), rec.get("traffic", {}), rec.get("top")
    except OSError:
        pass
    for k, v in in_step.items():
        if k in kernels:
            kernels[k]["in_step"] = {"kernel_launches_per_step": v["kernel_launches_per_step"], "ms_per_step": v["ms_per_step"]}
    for k, v in traffic.items():
        if k in kernels:
            kernels[k]["traffic"] = v.get("bytes_per_launch")
    if top not in kernels:                               # no profile committed yet for this dtype: slowest MFMA group of the probe
        top = max((k for k in kernels if kernels[k]["bound"] == "mfma"), key=lambda k: kernels[k]["launch_ms"])
    head = dict(kernels[top])
    head.setdefault("traffic", None)
    head["kernels"] = {k: v for k, v in kernels.items() if k != top}
    head["profile_source"] = ("`traffic` (HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes) and every `in_step` block are "
                              "constants from %s (builder-run rocprofv3 on its own box, committed); `launch_ms`, `achieved`, `frac` are "
                              "measured live in this run with HIP events" % os.path.relpath(IN_STEP, ROOT)) if (in_step or traffic) else None
    return head


def cpu_baseline(args):
    """The CPU oracle (oracle/acr_oracle.py, a parity-pinned restatement of the reference) timed on this host's cores on
    a bounded sample of the same workload: train steps at the bench resolution with batch 1 -- one warm-up at the SAME
    shape, then 3 timed steps, median reported."""
    from oracle import acr_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from recipe import recipe_state_dict
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_layout.json")) as f:
        layout = json.load(f)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))                       # the GPU box's CPU share is 16 per GPU
    torch.set_num_threads(cores)
    log("cpu_baseline: oracle on %d threads (affinity %d)" % (cores, avail))
    sd = {k: v.requires_grad_(True) for k, v in recipe_state_dict(layout, 0).items()}
    bsz = 1
    g = torch.Generator().manual_seed(1)
    img = torch.randn(bsz, 3, args.size, args.size, generator=g)
    lab = torch.zeros(bsz, 20)
    lab[:, 0] = 1
    times = []
    for it in range(4):
        for v in sd.values():
            v.grad = None
        t0 = time.time()
        loss, _ = O.train_step(sd, O.HYBRID_BASE, img, lab, args.alpha)
        loss.backward()
        times.append(time.time() - t0)
        log("cpu_baseline: step %d (%s) %.2f s" % (it, "warm-up" if it == 0 else "timed", times[-1]))
    timed = sorted(times[1:])
    med = timed[len(timed) // 2]
    return {"value": round(bsz / med, 4), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle train step (fwd mirror + ACR loss + bwd), hybrid-base %dx%d, batch %d, fp32: 1 warm-up at the same shape "
                      "+ 3 timed, median %.2f s (min %.2f, max %.2f)" % (args.size, args.size, bsz, med, timed[0], timed[-1])}


def infer_record(args, dev, with_cpu):
    """The CAM half of the path (infer_cam.py:141-215) as BASELINE configs[3] names it: one 384x384 network input with 2
    positive classes, flipped + plain pass at scales {0.5, 1, 1.5, 2} (T = 145 ... 2305), GETAM `grad` from layer 10 with
    affinity refinement, CAMs resized to 375x500; fp32 (the precision the argmax seeds are pinned in).  `value` = images/s
    of acr_wsss_amd.infer_cam.infer_cam_list walking a list of 16 such images with its defaults (round 6: batches of up to 8
    same-sized images, one batch in flight behind the one being collected, host loop included -- the reference resizes every
    image to crop x crop, so batches always form); `one_at_a_time_img_s` = the same walk with batch_size=1 (rounds 3-5's
    `value`), `single_image_latency_ms` = one call of infer_cam_image alone;
    `batch8_scale1` = 8 images per call at scale 1.  Kernel records: the attention pair at the largest scale through the
    C ABI (MFMA-bound), the GETAM row accumulation and the affinity product (HBM-bound, SURVEY 8d bytes).  CPU baseline:
    the oracle's infer_image on the same image at scale 1 only (bounded sample)."""
    from acr_wsss_amd import _lib as L, ops
    from acr_wsss_amd.DPT.ACR import ACR
    from acr_wsss_amd.infer_cam import infer_cam_image, infer_cam_images, infer_cam_list
    lib = L.load()
    torch.manual_seed(0)
    model = ACR(num_classes=20, backbone_name="vitb_hybrid", use_pretrain=False).to(dev).eval()
    g = torch.Generator(device="cpu").manual_seed(0)
    img = torch.randn(1, 3, 384, 384, generator=g).to(dev)
    lab = torch.zeros(1, 20)
    lab[0, 3] = lab[0, 11] = 1
    scales, out_hw = (0.5, 1.0, 1.5, 2.0), (375, 500)

    def timed(fn, n, warm=1):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    # a LIST of images walked one image at a time, as infer_cam.py:119-123 does: infer_cam_list keeps one image in flight behind
    # the one whose results it collects (launch_cam_images / collect), so the host's launch work overlaps the GPU's
    from acr_wsss_amd.infer_cam import launch_cam_images

    def walk(n, **kw):
        pending = None
        for _ in range(n):
            c = launch_cam_images(model, img, lab, [out_hw], **kw)
            if pending is not None:
                pending()
            pending = c
        pending()
    imgs8, labs8 = img.repeat(8, 1, 1, 1), lab.repeat(8, 1)
    items16 = [("img%02d" % i, img, lab, out_hw) for i in range(16)]
    # both arithmetics of the model's fp32 products: the split-product one carries `value` (every CAM / seed fixture of the
    # reference passes under it at unchanged tolerances: tests/test_model_gpu.py::test_infer_cam_*[f32_split]); the exact-fp32
    # numbers ride along as `f32_exact`
    per_math = {}
    for math in ("f32_split", "f32"):
        model.set_math(math)
        # (two warm-up calls each: a geometry is captured as a hipGraph the SECOND time it is seen, backbone.pass_graph)
        t_one = timed(lambda: infer_cam_image(model, img, lab, out_hw, scales=scales), 3, warm=2)
        t_ms = timed(lambda: walk(6, scales=scales), 2) / 6
        t_s1 = timed(lambda: infer_cam_image(model, img, lab, out_hw), 5, warm=2)
        t_b8 = timed(lambda: infer_cam_images(model, imgs8, labs8, [out_hw] * 8), 2, warm=2)
        # a list of same-sized images (what infer_cam.py walks: every VOC image is resized to the crop first) goes through
        # in batches -- results per image are those of the one-image call (test_infer_cam_images_batch_matches_single_images)
        t_b8ms = timed(lambda: infer_cam_images(model, imgs8, labs8, [out_hw] * 8, scales=scales), 2, warm=2)
        # THE list driver with its defaults: what a caller of infer_cam_list gets (VERDICT r5 #4a)
        t_list = timed(lambda: infer_cam_list(model, items16, scales=scales), 2, warm=1) / len(items16)
        log("infer[%s]: 4 scales %.1f ms/image through infer_cam_list (batches of 8), %.1f ms/image walking one image at a time (%.1f ms "
            "for one image alone); scale 1 %.1f ms/image, batch 8 %.1f ms; 4 scales, one batch of 8: %.1f ms/batch"
            % (math, t_list * 1e3, t_ms * 1e3, t_one * 1e3, t_s1 * 1e3, t_b8 * 1e3, t_b8ms * 1e3))
        per_math[math] = {"value": round(1.0 / t_list, 3), "ms_per_image": round(t_list * 1e3, 2),
                          "one_at_a_time_img_s": round(1.0 / t_ms, 3), "one_at_a_time_ms_per_image": round(t_ms * 1e3, 2),
                          "single_image_latency_ms": round(t_one * 1e3, 2), "scale1_img_s": round(1.0 / t_s1, 2),
                          "batch8_scale1": round(8.0 / t_b8, 2), "batch8_4scales": round(8.0 / t_b8ms, 2)}
    rec = {"workload": "BASELINE configs[3]: a list of 16 images, 384x384 base, scales {0.5,1,1.5,2}, 2 classes, flipped + plain pass, GETAM grad "
                       "start_layer 10 + affinity, CAMs at 375x500 (infer_cam.py:141-215), walked by infer_cam_list with its defaults "
                       "(batches of 8); synthetic image, seeded init",
           "metric": "img/s CAM generation, 1 GPU", "unit": "img/s", "dtype": "f32_split", "precision": PRECISION["f32_split"]}
    rec.update(per_math["f32_split"])
    rec["f32_exact"] = per_math["f32"]
    rec["peak_mem_gb"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
    del model, imgs8
    gc.collect()
    torch.cuda.empty_cache()
    # ---- kernel records at the largest scale: T = (768/16)^2 + 1 = 2305, the flip pair as a batch of 2, 12 heads
    B, H, T, Lyr = 2, 12, 2305, 12
    peak = PEAK_MFMA["f32"]
    gq = torch.Generator(device="cpu").manual_seed(1)
    qkv = torch.randn(B, T, 3 * H * 64, generator=gq).to(dev)
    d_o = torch.randn(B, T, H * 64, generator=gq).to(dev)
    o = torch.empty(B, T, H * 64, device=dev)
    lse2 = torch.empty(B, H, T, device=dev)
    pm = torch.empty(B, T, T, device=dev)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, H, T, device=dev)
    d = ops._desc(B, H, T, torch.float32)
    qp, kp, vp = ops._qkv_ptrs(qkv, H)
    dqp, dkp, dvp = ops._qkv_ptrs(dqkv, H)
    st = L.stream_ptr()
    sres = torch.empty(lib.acr_attn_scores_floats(d), device=dev)
    unit = 2.0 * T * T * 64 * B * H
    t = time_kernel(lambda: L.check(lib.acr_attn_fwd_scores(d, qp, kp, vp, L.ptr(o), L.ptr(lse2), L.ptr(sres), L.ptr(pm), T * T, T, st), "fwd"))
    kernels = {"acr_attn_fwd": _mfma_rec("acr_attn_fwd_scores (+ head-mean stream), B=2 H=12 T=2305", 2 * unit, 2 * unit, t, peak)}
    t = time_kernel(lambda: L.check(lib.acr_attn_bwd_scores(d, qp, kp, vp, L.ptr(o), L.ptr(d_o), L.ptr(lse2), L.ptr(sres), None, 0, 0, dqp, dkp,
                                                            dvp, L.ptr(delta), st), "bwd"))
    kernels["acr_attn_bwd"] = _mfma_rec("acr_attn_bwd_scores (no head-mean gradient: the class logit's backward), B=2 H=12 T=2305",
                                        4 * unit, 5 * unit, t, peak)
    # the same pair under the split-product arithmetic (what `value` runs): fp32-equivalent FLOP against the bf16 peak / 6
    dx = ops._desc(B, H, T, torch.float32, math=1)
    sres3 = torch.empty(lib.acr_attn_scores_floats(dx), device=dev)
    delta3 = torch.empty(lib.acr_attn_bwd_ws_floats(dx), device=dev)
    t = time_kernel(lambda: L.check(lib.acr_attn_fwd_scores(dx, qp, kp, vp, L.ptr(o), L.ptr(lse2), L.ptr(sres3), L.ptr(pm), T * T, T, st), "fwd"))
    kernels["acr_attn_fwd_x3"] = _mfma_rec("acr_attn_fwd_scores [ACR_F32_BF16X3] (+ split + head-mean stream), B=2 H=12 T=2305", 2 * unit, 2 * unit, t,
                                           PEAK_MFMA["f32_split"])
    t = time_kernel(lambda: L.check(lib.acr_attn_bwd_scores(dx, qp, kp, vp, L.ptr(o), L.ptr(d_o), L.ptr(lse2), L.ptr(sres3), None, 0, 0, dqp, dkp,
                                                            dvp, L.ptr(delta3), st), "bwd"))
    kernels["acr_attn_bwd_x3"] = _mfma_rec("acr_attn_bwd_scores [ACR_F32_BF16X3] (no head-mean gradient), B=2 H=12 T=2305", 4 * unit, 5 * unit, t,
                                           PEAK_MFMA["f32_split"])
    del sres3, delta3
    cam_row = torch.zeros(T - 1, device=dev)
    t = time_kernel(lambda: ops.getam_row_accum(qkv, d_o, lse2, H, 0, "grad", cam_row))
    row_bytes = 4.0 * (3 * H * 64 * T + H * 64 + H * T + (T - 1))     # k, v of one sample once, q/dO row 0, lse2 row, the output row
    kernels["acr_getam_row_accum"] = {"bound": "hbm", "kernel": "getam_row (row 0 of relu(dP) head mean, one layer, T=2305)",
                                      "achieved": round(row_bytes / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                      "frac": round(row_bytes / t / 8e12, 5), "launch_ms": round(t * 1e3, 4), "bytes_per_launch": row_bytes,
                                      "note": "latency-bound: 7 MB per launch"}
    attn = torch.rand(Lyr, T, T, device=dev)
    cams = torch.rand(2, T - 1, device=dev)
    t = time_kernel(lambda: ops.aff_refine(attn, cams))
    aff_bytes = 4.0 * Lyr * (T - 1) * (T - 1)
    kernels["acr_aff_refine"] = {"bound": "hbm", "kernel": "aff_refine (sum over 12 layers of the head-mean map, times 2 class rows, T=2305)",
                                 "achieved": round(aff_bytes / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                 "frac": round(aff_bytes / t / 8e12, 4), "launch_ms": round(t * 1e3, 4), "bytes_per_launch": aff_bytes}
    head = dict(kernels.pop("acr_attn_bwd"))
    head["traffic"] = None
    head["kernels"] = kernels
    rec["roofline"] = head
    del sres, qkv, d_o, o, pm, dqkv, attn
    torch.cuda.empty_cache()
    if with_cpu:
        from oracle import acr_oracle as O
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        from recipe import recipe_state_dict
        with open(os.path.join(ROOT, "tests", "golden", "state_dict_layout.json")) as f:
            layout = json.load(f)
        sd = recipe_state_dict(layout, 0)
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        torch.set_num_threads(max(1, min(16, avail)))          # the GPU box's CPU share is 16 per GPU
        t0 = time.time()
        O.infer_image(sd, O.HYBRID_BASE, img.cpu(), lab, out_hw, start_layer=10, func="grad", aff=True, scales=(1,))
        tc = time.time() - t0
        log("infer cpu_baseline: oracle infer_image at scale 1: %.1f s" % tc)
        rec["cpu_baseline"] = {"value": round(1.0 / tc, 4), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "oracle infer_image, the same 384x384 image, scale 1 ONLY (flipped + plain pass, 2 classes x 2 full "
                                         "backwards): %.1f s; compare with scale1_img_s, not with value" % tc}
    return rec


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: start the N ranks as a child torch.distributed.run job.  Decided
    before this process has made any GPU call (it never does); the child's stderr passes through, its one JSON line is
    relayed on stdout, and its return code becomes ours."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    print("[bench] self-launch: %s" % " ".join(cmd), file=sys.stderr, flush=True)
    child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
    for ln in child.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return child.returncode if (child.returncode != 0 or lines) else 1


def dist_record(args, world, rank, local, dev, sync_info):
    """What the process group looked like from inside (rank 0 reports): backend, world size, every rank's device, the RCCL
    version and GradSync's bucket / launch counters -- so that an N > 1 line shows what actually ran."""
    mine = {"rank": rank, "local_rank": local, "device": "cuda:%d" % dev.index, "name": torch.cuda.get_device_name(dev),
            "pid": os.getpid()}
    try:
        mine["uuid"] = str(torch.cuda.get_device_properties(dev).uuid)
    except Exception:
        pass
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    rec = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_devices": allr,
           "distinct_devices": len({r.get("uuid", r["device"]) for r in allr}), "launched_by": os.environ.get("TORCHELASTIC_RUN_ID", "env")}
    try:
        rec["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        rec["nccl_version"] = None
    if sync_info:
        rec.update(sync_info)
    return rec


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.time()


def run_mode(args, dtype, world, rank, dev):
    """Build the model in `dtype`, do W untimed + K timed steps (barrier + synchronize on both sides, max over ranks)."""
    from acr_wsss_amd.DPT.ACR import ACR
    from acr_wsss_amd.train import MasterWeights, PolyOptimizer, train_step
    from acr_wsss_amd.dp import GradSync, broadcast_parameters

    torch.manual_seed(0)
    # the arithmetic is a property of the model (a per-call argument of the C ABI), not a process-wide switch
    model = ACR(num_classes=args.classes, backbone_name="vitb_hybrid", use_pretrain=False, channels_last=args.channels_last,
                math="f32_split" if dtype == "f32_split" else "f32").to(dev)
    if args.channels_last:
        model = model.to(memory_format=torch.channels_last)
    model.train()
    broadcast_parameters(model)
    img, label = make_batch(args.batch, args.size, args.classes, rank, dev)
    amp = None
    if dtype == "bf16" and args.amp == "master":
        opt = MasterWeights(model, lambda ps: PolyOptimizer(ps, lr=0.05, weight_decay=5e-4, max_step=100000))
        img = img.to(torch.bfloat16)
    else:
        opt = PolyOptimizer(model.parameters(), lr=0.05, weight_decay=5e-4, max_step=100000)
        amp = torch.bfloat16 if dtype == "bf16" else None          # f32_split: fp32 tensors, no autocast
    # the training graph is static: after one agreeing step the per-step host-side flag exchange is skipped (dp.GradSync)
    sync = (GradSync(model.parameters(), late_params=model.late_gradient_parameters(), static_graph=os.environ.get("ACR_DP_STATIC", "1") != "0")
            if (world > 1 or os.environ.get("ACR_FORCE_GRADSYNC") == "1") else None)

    def step():
        return train_step(model, opt, img, label, args.alpha, grad_sync=sync, amp_dtype=amp)

    torch.cuda.reset_peak_memory_stats()
    log("%s: model built; warmup" % dtype)
    loss0 = None                                         # loss of the FIRST step: same weights, same batch in every mode, no update yet
    for i in range(args.warmup):
        loss, terms = step()
        torch.cuda.synchronize()
        if loss0 is None:
            loss0 = {k: float(v.detach()) for k, v in terms.items()}
        log("%s: warmup step %d done" % (dtype, i))
    # per-kernel durations are taken INSIDE the timed steps: HIP events around the first launch of every hooked kernel shape
    # in every step, on the launch stream (ops.KernelTimer; ~20 event pairs per step, < 0.1 % of a step)
    from acr_wsss_amd import ops
    timer = ops.KernelTimer() if (rank == 0 and not args.no_roofline) else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.KERNEL_TIMER = timer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if timer is not None:
            timer.next_step()
        loss, terms = step()
        if loss0 is None:                                # --warmup 0: one host read-back inside the timed region, first step only
            loss0 = {k: float(v.detach()) for k, v in terms.items()}
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.KERNEL_TIMER = None
    live = {k: sum(v) / len(v) for k, v in timer.collect().items()} if timer is not None else {}
    log("%s: timed %d steps in %.3f s" % (dtype, args.steps, elapsed))
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    value = args.batch * world * args.steps / elapsed
    # Diagnostics for an N > 1 run, taken AFTER the timed region in the same process: the same step with the
    # exchange disarmed (every rank steps alone: compute only) and the buckets all-reduced back to back with no compute beside
    # them, so that the line itself says how much of a step the exchange exposed (`exposed_ms` = ms_step - ms_compute_only) and
    # what the all-reduce costs when nothing hides it -- the first real 8-GPU run is then a diagnosis, not a single number.
    diag = None
    if sync is not None:
        def span(fn, n):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                fn()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dt = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            return float(dt) / n
        nd = max(2, min(args.steps, 6))
        span(lambda: train_step(model, opt, img, label, args.alpha, grad_sync=None, amp_dtype=amp), 1)      # (no-sync warm-up: .grad storage changes hands)
        t_comp = span(lambda: train_step(model, opt, img, label, args.alpha, grad_sync=None, amp_dtype=amp), nd)
        span(sync.exchange_only, 1)
        t_xchg = span(sync.exchange_only, nd)
        ms_step = elapsed / args.steps * 1e3
        diag = {"ms_step": round(ms_step, 3), "ms_compute_only": round(t_comp * 1e3, 3), "ms_allreduce_only": round(t_xchg * 1e3, 3),
                "exposed_ms": round(ms_step - t_comp * 1e3, 3), "steps_each": nd,
                "allreduce_mb": round(sum(b.flat.numel() * b.flat.element_size() for b in sync.buckets) / 2 ** 20, 1),
                "allreduce_busbw_gbs": round(2.0 * (world - 1) / max(world, 1) * sum(b.flat.numel() * b.flat.element_size() for b in sync.buckets)
                                             / max(t_xchg, 1e-9) / 1e9, 1),
                "note": "max over ranks, barrier + synchronize on both sides; compute-only = the same train_step with GradSync disarmed; "
                        "allreduce-only = every bucket all-reduced back to back, nothing else running; busbw = 2(N-1)/N x bytes / time"}
        log("%s: diagnostics: step %.2f ms, compute only %.2f ms, all-reduce only %.2f ms" % (dtype, ms_step, t_comp * 1e3, t_xchg * 1e3))
    rec = {"value": round(value, 3), "unit": "img/s", "ms_per_step": round(elapsed / args.steps * 1e3, 3), "dtype": dtype,
           "precision": PRECISION[dtype] if not (dtype == "bf16" and args.amp == "autocast") else "torch.autocast(bf16)",
           "loss": round(float(loss.detach()), 5), "loss_step0": loss0,
           "step_mfma_frac": round(value * FLOP_PER_IMG_448 * (args.size / 448.0) ** 2 / (world * PEAK_MFMA[dtype]), 4),
           "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    rec["_live_ms"] = live                               # popped by main(): feeds the roofline records
    if sync is not None:
        info = sync.describe()
        per_step = max(1, info["steps"])
        rec["_sync"] = {"buckets": info["buckets"], "bucket_mb": info["bucket_mb"], "unused_parameters": info["unused_parameters"],
                        "bucket_launches_in_backward": info["bucket_launches_in_backward"],
                        "bucket_launches_in_finish": info["bucket_launches_in_finish"],
                        "bucket_launches_in_backward_per_step": round(info["bucket_launches_in_backward"] / per_step, 2),
                        "late_reexchanges": info["late_reexchanges"], "rank_disagreements": info["rank_disagreements"],
                        "host_agreement_exchanges": info["agreement_exchanges"], "static_graph": sync.static_graph,
                        "static_deviations": info["static_deviations"], "steps_counted": info["steps"]}
        if diag is not None:
            rec["_sync"].update(diag)
    del model, opt, sync, step, img, label, loss
    gc.collect()
    torch.cuda.empty_cache()
    return rec


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # the driver's `python bench.py --gpus N`: no GPU call has happened yet
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            assert ndev >= world, "need one GPU per rank for RCCL (%d GPUs, %d ranks)" % (ndev, world)
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:                                              # rehearsal: ranks may share a GPU, collectives via gloo
            local = local % ndev
            dist.init_process_group(args.backend)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from acr_wsss_amd.tuning import enable_tuned_gemms, use_shipped_miopen_db
    use_shipped_miopen_db()                                # before the first convolution
    tuned = enable_tuned_gemms()                           # shipped hipBLASLt selections for the three library GEMMs
    if os.environ.get("ACR_MIOPEN_FIND", "0") == "1":      # experiment: let MIOpen benchmark its solvers per conv shape
        torch.backends.cudnn.benchmark = True
    if args.stock_linear:
        from acr_wsss_amd.backbone import Attention
        Attention.hip_linear = False
    if args.dtype == "both":
        modes = ["f32", "f32_split", "bf16"] if os.environ.get("ACR_BENCH_HEADLINE", "f32_split") == "f32" else ["f32_split", "f32", "bf16"]
    else:
        modes = [args.dtype]
    if args.probe_only:
        print(json.dumps({m: roofline_probe(args, dev, m) for m in modes}), flush=True)
        return
    if args.infer_only:
        print(json.dumps({"infer": infer_record(args, dev, not args.no_cpu_baseline)}), flush=True)
        return

    runs = {m: run_mode(args, m, world, rank, dev) for m in modes}
    lives = {m: runs[m].pop("_live_ms", {}) for m in modes}
    syncs = {m: runs[m].pop("_sync", None) for m in modes}
    dist_info = dist_record(args, world, rank, local, dev, {"sync": {m: syncs[m] for m in modes}}) if world > 1 else None
    if rank == 0:
        head = runs[modes[0]]                              # fp32 unless a single dtype was asked for
        head_live = lives[modes[0]]
        std = (args.size, args.batch, args.classes) == (448, 16, 20)
        out = {
            "metric": "img/s ACR-ViT-hybrid-base %dx%d train step" % (args.size, args.size), "value": head["value"], "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": head["dtype"], "data": "synthetic",
            "precision": head["precision"],
            "config": {"workload": "%sViT-hybrid-base (DPT) %dx%d, batch %d per GPU, ACR train step (2 views, fwd+bwd+SGD)"
                                   % ("BASELINE configs[1]: " if std else "", args.size, args.size, args.batch),
                       "global_batch": args.batch * world, "classes": args.classes, "alpha": args.alpha,
                       "parallelism": "dp%d" % world, "tokens_per_view": (args.size // 16) ** 2 + 1},
            "loss": head["loss"], "step_mfma_frac": head["step_mfma_frac"], "peak_mem_gb": head["peak_mem_gb"],
            "tuned_library_gemms": bool(tuned),
        }
        # `loss` is the loss after warmup + steps chaotic SGD updates (lr 0.05 on random weights) and says nothing across modes;
        # `loss_step0` is the loss (and its terms) of the very first step -- identical weights and batch in every mode, no update
        # yet -- so the two fp32 arithmetics must agree to fp32 rounding (VERDICT r4 #1c; tests hold it to 5e-5 relative)
        out["loss_step0"] = {m: runs[m]["loss_step0"] for m in modes}
        if "f32" in runs and "f32_split" in runs and runs["f32"]["loss_step0"] and runs["f32_split"]["loss_step0"]:
            a, b = runs["f32"]["loss_step0"], runs["f32_split"]["loss_step0"]
            out["loss_step0"]["f32_split_vs_f32_rel"] = {k: abs(b[k] - a[k]) / max(abs(a[k]), 1e-30) for k in a}
            out["loss_step0"]["agree_5e-5"] = all(v <= 5e-5 for v in out["loss_step0"]["f32_split_vs_f32_rel"].values())
            # short top-level twins (the driver's `parsed` keeps top-level scalars and truncates nested records)
            out["loss_step0_agree"] = bool(out["loss_step0"]["agree_5e-5"])
            out["loss_step0_max_rel"] = float("%.3e" % max(out["loss_step0"]["f32_split_vs_f32_rel"].values()))
        # rank 0 probes its own GPU while the other ranks wait at the closing barrier (N > 1: the line says how much of each
        # kernel's peak a rank reaches and what the exchange looked like)
        if not args.no_roofline:
            out["roofline"] = roofline_probe(args, dev, head["dtype"], head_live)
            log("roofline probe (%s) done" % head["dtype"])
        for m in modes[1:]:
            sub = dict(runs[m])
            if not args.no_roofline:
                sub["roofline"] = roofline_probe(args, dev, m, lives[m])
                log("roofline probe (%s) done" % m)
            if "f32" in runs and "f32_split" in runs and m in ("f32", "f32_split"):
                sub["vs_" + head["dtype"]] = round(sub["value"] / head["value"], 3)
            if m == "f32":
                sub["note"] = ("the exact-fp32-MFMA arithmetic (rounds 1-3's headline), kept beside the split-product headline: same tensors, "
                               "same kernels' tile structure, v_mfma_f32_32x32x2_f32 products; its bound is 157.3 TF / 1.131 TF per image = 139 img/s")
            out[m] = sub
        if dist_info is not None:
            out["dist"] = dist_info
        if world == 1 and not args.no_infer:
            out["infer"] = infer_record(args, dev, not args.no_cpu_baseline)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
