#!/usr/bin/env python3
"""bench.py -- ACR training-step throughput (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N=1: run directly)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W  (N>1: one rank per GPU, RCCL)

A "step" is one iteration of train_acr.py:127-174 on a synthetic batch already resident in HBM: h-flip view,
ViT-hybrid-base forward over both views, ACR loss, full backward, (RCCL gradient all-reduce,) PolyOptimizer
step.  Workload = BASELINE configs[1]: 448x448, batch 16 per GPU (weak scaling: configs[2] is 16 x 8).
Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_IMG_448 = 1.1307e12          # SURVEY 6 [probe]: 2 views, fwd+bwd, hybrid-base @448^2
PEAK_MFMA = {"f32": 157.3e12, "bf16": 2.5e15}   # MI355X_MICROARCH.md: dense matrix peaks


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU (BASELINE configs[1]: 16)")
    ap.add_argument("--size", type=int, default=448)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default=os.environ.get("ACR_BENCH_DTYPE", "bf16"))
    ap.add_argument("--classes", type=int, default=20)
    ap.add_argument("--alpha", type=int, default=125)
    ap.add_argument("--amp", choices=["master", "autocast"], default="master",
                    help="bf16 mode: bf16 model + fp32 master weights (default) or torch.autocast")
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--stock-linear", action="store_true", help="qkv/proj on torch F.linear instead of acr_linear_bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--probe-only", action="store_true", help="run only the kernel roofline probe (for rocprofv3 --pmc passes)")
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


def roofline_probe(args, dev):
    """Dominant hand-written kernel of the step at the bench geometry: the attention backward's dK/dV sweep
    (4 of the 11 T^2 x 64 products per head per layer).  Launched directly through the C ABI and timed with
    events; algorithmic FLOPs = 4 products x 2*T*T*64 x B*H per launch (DESIGN.md 'kernels')."""
    from acr_wsss_amd import _lib as L, ops
    lib = L.load()
    B, H, T = 2 * args.batch, 12, (args.size // 16) ** 2 + 1
    dt = torch.float32 if args.dtype == "f32" else torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(0)
    qkv = torch.randn(B, T, 3 * H * 64, generator=g).to(dev).to(dt)
    d_o = torch.randn(B, T, H * 64, generator=g).to(dev).to(dt)
    gm = (torch.randn(B, T, ops.pad4(T), generator=g).to(dev) * 1e-3)[:, :, :T]
    o = torch.empty(B, T, H * 64, dtype=dt, device=dev)
    lse2 = torch.empty(B, H, T, dtype=torch.float32, device=dev)
    pm = torch.empty(B, T, T, dtype=torch.float32, device=dev)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, H, T, dtype=torch.float32, device=dev)
    d = ops._desc(B, H, T, dt)
    qp, kp, vp = ops._qkv_ptrs(qkv, H)
    dqp, dkp, dvp = ops._qkv_ptrs(dqkv, H)
    st = L.stream_ptr()

    def fwd():
        L.check(lib.acr_attn_fwd(d, qp, kp, vp, L.ptr(o), L.ptr(lse2), L.ptr(pm), T * T, T, st), "fwd")

    def bwd():
        L.check(lib.acr_attn_bwd(d, qp, kp, vp, L.ptr(o), L.ptr(d_o), L.ptr(lse2), L.ptr(gm), gm.stride(0), gm.stride(1), dqp, dkp, dvp,
                                 L.ptr(delta), st), "bwd")

    t_fwd = time_kernel(fwd)
    t_bwd = time_kernel(bwd)
    unit = 2.0 * T * T * 64 * B * H                      # one T x T x 64 product over all (b, h)
    fl_fwd, fl_bwd = 3 * unit, 8 * unit                  # fwd: S, PV (+S for head-mean); bwd: 1 + 4 + 3 products
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
    # second MFMA kernel of the step by time: the weight-gradient GEMM dW = dY^T X (fc1's shape), bf16 mode only
    wg = None
    if args.dtype == "bf16":
        Mtok, Nw, Kw = B * T, 3072, 768
        dyw = torch.randn(Mtok, Nw, generator=g).to(dev).bfloat16()
        xw = torch.randn(Mtok, Kw, generator=g).to(dev).bfloat16()
        wsw = torch.empty(lib.acr_wgrad_ws_floats(Mtok, Nw, Kw), dtype=torch.float32, device=dev)
        dww = torch.empty(Nw, Kw, dtype=torch.bfloat16, device=dev)

        def wgrad():
            L.check(lib.acr_wgrad_bf16(L.ptr(dyw), Nw, L.ptr(xw), Kw, Mtok, Nw, Kw, L.ptr(wsw), L.ptr(dww), st), "wgrad")
        t_wg = time_kernel(wgrad)
        wg = {"bound": "mfma", "achieved": round(2.0 * Mtok * Nw * Kw / t_wg / 1e12, 2), "peak": PEAK_MFMA["bf16"] / 1e12,
              "unit": "TFLOP/s", "frac": round(2.0 * Mtok * Nw * Kw / t_wg / PEAK_MFMA["bf16"], 4),
              "launch_ms": round(t_wg * 1e3, 3), "shape": "dW(3072x768) over %d tokens, incl. slab reduction" % Mtok}
    peak = PEAK_MFMA[args.dtype] if args.dtype == "f32" else PEAK_MFMA["bf16"]
    ach = fl_bwd / t_bwd / 1e12
    # HBM traffic of the same launches from the committed rocprofv3 PMC passes (bench.py --probe-only under
    # --pmc FETCH_SIZE / --pmc WRITE_SIZE, gfx950 x2 fetch correction applied); only valid for the default geometry
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            if args.dtype == "bf16" and args.batch == 16 and args.size == 448:
                traffic = json.load(f)["acr_attn_bwd_bytes_per_launch"]
    except OSError:
        pass
    return {
        "bound": "mfma", "kernel": "acr_attn_bwd (delta + dkdv + dq)", "achieved": round(ach, 2),
        "peak": peak / 1e12, "unit": "TFLOP/s", "frac": round(ach / (peak / 1e12), 4), "traffic": traffic,
        "launch_ms": round(t_bwd * 1e3, 3), "flops_per_launch": fl_bwd,
        "attn_fwd": {"achieved": round(fl_fwd / t_fwd / 1e12, 2), "launch_ms": round(t_fwd * 1e3, 3)},
        "consistency_fwd": {"bound": "hbm", "achieved": round(k3_bytes / t_k3 / 1e9, 1), "peak": 8000.0,
                            "unit": "GB/s", "frac": round(k3_bytes / t_k3 / 8e12, 4), "launch_ms": round(t_k3 * 1e3, 3)},
        "wgrad_gemm": wg,
    }


def cpu_baseline(args):
    """The CPU oracle (oracle/acr_oracle.py, a parity-pinned restatement of the reference) timed on this
    host's cores on a bounded sample of the same workload: 1 step at 448^2 with a small batch."""
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
    warm = torch.randn(1, 3, 64, 64, generator=g)
    lab = torch.zeros(1, 20)
    lab[:, 0] = 1
    O.train_step(sd, O.HYBRID_BASE, warm, lab, args.alpha)[0].backward()
    log("cpu_baseline: warm-up step (64x64) done")
    img = torch.randn(bsz, 3, args.size, args.size, generator=g)
    lab = torch.zeros(bsz, 20)
    lab[:, 0] = 1
    t0 = time.time()
    loss, _ = O.train_step(sd, O.HYBRID_BASE, img, lab, args.alpha)
    loss.backward()
    dt = time.time() - t0
    return {"value": round(bsz / dt, 4), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 oracle train step (fwd mirror + ACR loss + bwd), hybrid-base %dx%d, batch %d, fp32, %.1f s"
                      % (args.size, args.size, bsz, dt)}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.time()


def main():
    args = parse()
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
    if args.probe_only:
        print(json.dumps(roofline_probe(args, dev)), flush=True)
        return

    from acr_wsss_amd.DPT.ACR import ACR
    from acr_wsss_amd.train import MasterWeights, PolyOptimizer, train_step
    from acr_wsss_amd.dp import GradSync, broadcast_parameters

    torch.manual_seed(0)
    model = ACR(num_classes=args.classes, backbone_name="vitb_hybrid", use_pretrain=False,
                channels_last=args.channels_last).to(dev)
    if args.channels_last:
        model = model.to(memory_format=torch.channels_last)
    if args.stock_linear:
        from acr_wsss_amd.backbone import Attention
        Attention.hip_linear = False
    model.train()
    broadcast_parameters(model)
    img, label = make_batch(args.batch, args.size, args.classes, rank, dev)
    amp = None
    if args.dtype == "bf16" and args.amp == "master":
        opt = MasterWeights(model, lambda ps: PolyOptimizer(ps, lr=0.05, weight_decay=5e-4, max_step=100000))
        img = img.to(torch.bfloat16)
    else:
        opt = PolyOptimizer(model.parameters(), lr=0.05, weight_decay=5e-4, max_step=100000)
        amp = torch.bfloat16 if args.dtype == "bf16" else None
    sync = GradSync(model.parameters()) if (world > 1 or os.environ.get("ACR_FORCE_GRADSYNC") == "1") else None

    def step():
        return train_step(model, opt, img, label, args.alpha, grad_sync=sync, amp_dtype=amp)

    log("model built; warmup")
    for i in range(args.warmup):
        loss, _ = step()
        torch.cuda.synchronize()
        log("warmup step %d done" % i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log("timed %d steps in %.3f s" % (args.steps, elapsed))
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss_val = float(loss.detach())

    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / elapsed
        out = {
            "metric": "img/s ACR-ViT-hybrid-base %dx%d train step" % (args.size, args.size), "value": round(value, 3), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "precision": ("bf16 params/activations/grads, fp32 master weights + fp32 softmax/loss" if (args.dtype == "bf16" and args.amp == "master")
                          else ("torch.autocast(bf16)" if args.dtype == "bf16" else "fp32 end to end (reference precision)")),
            "config": {"workload": "%sViT-hybrid-base (DPT) %dx%d, batch %d per GPU, ACR train step "
                                   "(2 views, fwd+bwd+SGD)" % ("BASELINE configs[1]: " if (args.size, args.batch, args.classes) == (448, 16, 20) else "",
                                                               args.size, args.size, args.batch),
                       "global_batch": args.batch * world, "classes": args.classes, "alpha": args.alpha,
                       "parallelism": "dp%d" % world, "tokens_per_view": (args.size // 16) ** 2 + 1},
            "loss": round(loss_val, 5),
            "step_mfma_frac": round(value * FLOP_PER_IMG_448 * (args.size / 448.0) ** 2 / (world * PEAK_MFMA[args.dtype]), 4),
            "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            "tuned_library_gemms": bool(tuned),
        }
        if world == 1 and not args.no_roofline:
            out["roofline"] = roofline_probe(args, dev)
            log("roofline probe done")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
