"""torch.autograd wrappers around the C ABI (include/acr_hip.h).

These are the only places where the Python host surface meets the HIP library.  Tensors are torch
allocations (device memory + stream plumbing); all arithmetic of the hot path happens in the kernels.
"""
import os
import torch
from torch.autograd import Function

from . import _lib as L

HEAD_DIM = 64
# fp32 attention with a backward keeps its logits resident in HBM (csrc/attn_f32_sres.hip); ACR_ATTN_F32_SCORES=0 selects the
# recompute generation (csrc/attn_f32_dma.hip) for A/B runs
ATTN_F32_SCORES = True


def pad4(n):
    """Row pitch (floats) of gradient maps: T rounded up to 16 bytes so kernels can use vector loads."""
    return (n + 3) & ~3


def _desc(B, H, T, dtype, packed_qkv=True, math=0):
    """Descriptor for q/k/v aliasing slices of the packed qkv Linear output (B,T,3,H,64) and o / do in
    the reference's (B,T,H*64) activation layout (models/vision_transformer.py:200-201,211).  ``math`` = 1 on fp32 tensors:
    ACR_F32_BF16X3 (split products on the bf16 MFMA; resident-score entry points only)."""
    D = H * HEAD_DIM
    d = L.AttnDesc()
    d.B, d.H, d.T, d.head_dim = B, H, T, HEAD_DIM
    d.dtype = L.ACR_F32_BF16X3 if (math == 1 and dtype == torch.float32) else L.dtype_code(dtype)
    d.scale = HEAD_DIM ** -0.5
    d.qkv_sb, d.qkv_st, d.qkv_sh = T * 3 * D, 3 * D, HEAD_DIM
    d.o_sb, d.o_st, d.o_sh = T * D, D, HEAD_DIM
    return d


def _qkv_ptrs(qkv, H):
    esz = qkv.element_size()
    base = qkv.data_ptr()
    D = H * HEAD_DIM
    return (L.c_void_p(base), L.c_void_p(base + D * esz), L.c_void_p(base + 2 * D * esz))


class MeanStack:
    """Owner of the (B, L, T, T) fp32 head-mean stack of one forward pass (DPT/ACR.py:107-112).

    Each attention layer's kernel writes its slice in place (strided output), so the reference's 12
    ``mean(dim=1)`` kernels and the ``torch.stack`` copy never happen.  Not a tensor on purpose: autograd
    must not see the buffer as an input of the per-layer Functions."""

    def __init__(self, B, Lyr, T, device):
        self.buf = torch.empty((B, Lyr, T, T), dtype=torch.float32, device=device)
        self.n_layers = Lyr



class KernelTimer:
    """bench.py only: HIP-event timing of selected launches INSIDE the timed training steps (the first matching launch of every
    key in every step), on the stream the kernels are launched on -- so that a kernel's reported duration is the in-step one,
    not that of an isolated replay.  Off (None) everywhere else; an `is None` test is all the product path pays."""

    def __init__(self):
        self.samples, self._seen, self._open = {}, set(), []

    def next_step(self):
        self._seen.clear()

    def begin(self, key):
        if key in self._seen:
            return None
        self._seen.add(key)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        return key, e0, e1

    def end(self, tok):
        tok[2].record()
        self._open.append(tok)

    def collect(self):
        """-> {key: [ms, ...]} (synchronises)"""
        torch.cuda.synchronize()
        for key, e0, e1 in self._open:
            self.samples.setdefault(key, []).append(e0.elapsed_time(e1))
        self._open = []
        return self.samples


KERNEL_TIMER = None


def _t0(name, *dims):
    return KERNEL_TIMER.begin("%s %s" % (name, "x".join(str(int(d)) for d in dims))) if KERNEL_TIMER is not None else None


def _t1(tok):
    if tok is not None:
        KERNEL_TIMER.end(tok)

class AttnCoreFn(Function):
    """o = softmax(q k^T d^-0.5) v  (+ head-mean side output), packed qkv in, (B,T,D) out."""

    @staticmethod
    def forward(ctx, qkv, heads, stack, layer, owner, math=0, want_oimg=False):
        L.require_gpu(qkv)
        # an output nobody differentiated arrives as None in backward, not as a zero tensor: CAM generation back-propagates
        # the class logit through `o` only, and a materialised zero head-mean gradient would cost a (B,T,T) fill + re-layout
        # per layer and send the backward down its with-G path
        ctx.set_materialize_grads(False)
        if not qkv.is_contiguous():
            qkv = qkv.contiguous()
        B, T, D3 = qkv.shape
        D = heads * HEAD_DIM
        assert D3 == 3 * D, "qkv last dim %d != 3*heads*64" % D3
        lib = L.load()
        keep_scores = ctx.needs_input_grad[0]
        # split products (math = 1) exist for the resident-score kernels of fp32 tensors; every other path is exact
        ctx.math = math = int(math == 1 and ATTN_F32_SCORES and qkv.dtype == torch.float32 and (keep_scores or stack is not None))
        d = _desc(B, heads, T, qkv.dtype, math=math)
        o = torch.empty((B, T, D), dtype=qkv.dtype, device=qkv.device)
        lse2 = torch.empty((B, heads, T), dtype=torch.float32, device=qkv.device)
        pm = None
        if stack is not None:
            pm = stack.buf[:, layer]
        qp, kp, vp = _qkv_ptrs(qkv, heads)
        # fp32 with a backward to come: keep the scaled logits resident (983 MB per layer at 32 views x 12 heads x 785 tokens;
        # the card has 288 GB) -- the head mean and the backward sweeps stream them back instead of recomputing q.k on the
        # fp32 MFMA (csrc/attn_f32_sres.hip)
        # (also without a backward when the head mean is asked for -- CAM generation below its start layer: storing the
        # logits once and streaming them into the head mean beats the recompute generation's second q.k product; the buffer
        # is then a temporary)
        scores = None
        if ATTN_F32_SCORES and qkv.dtype == torch.float32 and (keep_scores or pm is not None):
            scores = torch.empty(lib.acr_attn_scores_floats(d), dtype=torch.float32, device=qkv.device)
        tok = _t0("attn_fwd" if pm is not None else "attn_fwd_nomean", B, heads, T)
        pm_sb, pm_st = (pm.stride(0), pm.stride(1)) if pm is not None else (0, 0)
        # the output as the operand image of the Linear behind it (proj), written by the forward's epilogue instead of by an image
        # pass over o -- split products with resident scores only, and not where the forward runs split-tail workgroups
        # (T = 1025, 2305, ...: they write fp32 o only)
        oimg = None
        if want_oimg and scores is not None and math == 1 and lib.acr_attn_fwd_oimg_offered(d):
            oimg = x3_image_empty(B * T, D, qkv.device)     # rows past B*T of the last 128-row block: zeroed by the kernel's epilogue
        if oimg is not None:
            L.check(lib.acr_attn_fwd_scores_oimg(d, qp, kp, vp, L.ptr(o), L.ptr(lse2), L.ptr(scores), L.ptr(pm), pm_sb, pm_st, L.ptr(oimg),
                                                 L.stream_ptr()), "acr_attn_fwd_scores_oimg")
        elif scores is not None:
            L.check(lib.acr_attn_fwd_scores(d, qp, kp, vp, L.ptr(o), L.ptr(lse2), L.ptr(scores), L.ptr(pm), pm_sb, pm_st,
                                            L.stream_ptr()), "acr_attn_fwd_scores")
        else:
            L.check(lib.acr_attn_fwd(d, qp, kp, vp, L.ptr(o), L.ptr(lse2), L.ptr(pm), pm_sb, pm_st, L.stream_ptr()),
                    "acr_attn_fwd")
        _t1(tok)
        if scores is not None and keep_scores:
            ctx.save_for_backward(qkv, o, lse2, scores)
        else:
            ctx.save_for_backward(qkv, o, lse2)
        ctx.heads = heads
        # the state API (get_attn / get_attn_gradients / getam) needs q, k, lse2 and later dO; in training mode nobody reads
        # it and keeping it would pin qkv + dO of all 12 layers between steps (2.8 GB at B = 32 views in fp32), so it is
        # only recorded in eval mode (CAM inference) or when the module asks for it (keep_state_in_training)
        keep = owner is not None and (not owner.training or getattr(owner, "keep_state_in_training", False))
        ctx.owner = owner if keep else None
        if owner is not None:
            owner._saved = (qkv, lse2, heads) if keep else None
            owner._saved_do = None
            owner._saved_gpm = None
        if want_oimg:
            if oimg is not None:
                ctx.mark_non_differentiable(oimg)
            return o, pm, oimg
        if pm is None:
            return o, None
        return o, pm

    @staticmethod
    def backward(ctx, d_o, g_pm, *unused):
        saved = ctx.saved_tensors
        qkv, o, lse2 = saved[:3]
        scores = saved[3] if len(saved) > 3 else None
        heads = ctx.heads
        B, T, _ = qkv.shape
        lib = L.load()
        if d_o is None:
            d_o = torch.zeros_like(o)
        if not d_o.is_contiguous():
            d_o = d_o.contiguous()
        if d_o.dtype != qkv.dtype:
            d_o = d_o.to(qkv.dtype)
        gm_sb = gm_st = 0
        if g_pm is not None:
            # kernels want fp32 rows with a pitch that is a multiple of 4 floats (16-byte groups); the fused loss
            # (ConsistencyFn.backward) delivers exactly that, anything else is re-laid-out once here
            ok = (g_pm.dtype == torch.float32 and g_pm.stride(2) == 1 and g_pm.stride(1) % 4 == 0
                  and g_pm.stride(1) >= pad4(T) and g_pm.stride(0) % 4 == 0 and g_pm.data_ptr() % 16 == 0)
            if not ok:
                buf = torch.zeros((B, T, pad4(T)), dtype=torch.float32, device=qkv.device)
                buf[:, :, :T].copy_(g_pm)
                g_pm = buf[:, :, :T]
            gm_sb, gm_st = g_pm.stride(0), g_pm.stride(1)
        d = _desc(B, heads, T, qkv.dtype, math=ctx.math if scores is not None else 0)
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(lib.acr_attn_bwd_ws_floats(d) if scores is not None else B * heads * T, dtype=torch.float32, device=qkv.device)
        qp, kp, vp = _qkv_ptrs(qkv, heads)
        dqp, dkp, dvp = _qkv_ptrs(dqkv, heads)
        tok = _t0("attn_bwd" if g_pm is not None else "attn_bwd_nomean", B, heads, T)
        if scores is not None:
            L.check(lib.acr_attn_bwd_scores(d, qp, kp, vp, L.ptr(o), L.ptr(d_o), L.ptr(lse2), L.ptr(scores), L.ptr(g_pm), gm_sb,
                                            gm_st, dqp, dkp, dvp, L.ptr(delta), L.stream_ptr()), "acr_attn_bwd_scores")
        else:
            L.check(lib.acr_attn_bwd(d, qp, kp, vp, L.ptr(o), L.ptr(d_o), L.ptr(lse2), L.ptr(g_pm), gm_sb, gm_st,
                                     dqp, dkp, dvp, L.ptr(delta), L.stream_ptr()), "acr_attn_bwd")
        _t1(tok)
        if ctx.owner is not None:
            ctx.owner._saved_do = d_o
            ctx.owner._saved_gpm = g_pm       # dLoss/d(mean_h P): reaches every head's P as G/H (get_attn_gradients)
        return dqkv, None, None, None, None, None, None


class StackAliasFn(Function):
    """Expose MeanStack.buf as the autograd-visible (B,L,T,T) tensor without copying: forward returns the
    buffer the per-layer kernels already wrote, backward hands each layer its strided slice of the
    incoming gradient (what torch.stack's backward does with 12 copies in the reference)."""

    @staticmethod
    def forward(ctx, stack, *pms):
        ctx.n = len(pms)
        return stack.buf

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(g[:, l] for l in range(ctx.n))


def attention_core(qkv, heads, stack=None, layer=0, owner=None, math=0):
    return AttnCoreFn.apply(qkv, heads, stack, layer, owner, math)


ATTN_O_IMAGE = True      # A/B: the attention output's image from the forward's epilogue


def attention_core_oimg(qkv, heads, stack=None, layer=0, owner=None, math=0):
    """attention_core that also returns o's split-product image (or None where the forward cannot write it): (o, pmean, o_image)."""
    return AttnCoreFn.apply(qkv, heads, stack, layer, owner, math, True)


def attn_probs(qkv, lse2, heads):
    """Per-head P (B,H,T,T) fp32 recomputed from q, k, lse2 (Attention.get_attn compatibility)."""
    B, T, _ = qkv.shape
    lib = L.load()
    out = torch.empty((B, heads, T, T), dtype=torch.float32, device=qkv.device)
    qp, kp, _ = _qkv_ptrs(qkv, heads)
    L.check(lib.acr_attn_probs(_desc(B, heads, T, qkv.dtype), qp, kp, L.ptr(lse2), L.ptr(out), L.stream_ptr()),
            "acr_attn_probs")
    return out


def attn_dprobs(qkv, d_o, heads):
    """Per-head dO V^T (B,H,T,T) fp32 (Attention.get_attn_gradients compatibility)."""
    B, T, _ = qkv.shape
    lib = L.load()
    out = torch.empty((B, heads, T, T), dtype=torch.float32, device=qkv.device)
    _, _, vp = _qkv_ptrs(qkv, heads)
    L.check(lib.acr_attn_dprobs(_desc(B, heads, T, qkv.dtype), L.ptr(d_o), vp, L.ptr(out), L.stream_ptr()),
            "acr_attn_dprobs")
    return out


def getam_row_accum(qkv, d_o, lse2, heads, batch, func, cam_row):
    B, T, _ = qkv.shape
    lib = L.load()
    qp, kp, vp = _qkv_ptrs(qkv, heads)
    L.check(lib.acr_getam_row_accum(_desc(B, heads, T, qkv.dtype), qp, kp, vp, L.ptr(d_o), L.ptr(lse2), batch,
                                    L.GETAM_FUNCS[func], L.ptr(cam_row), L.stream_ptr()), "acr_getam_row_accum")


def getam_rows_accum(qkv, d_o, lse2, heads, func, cam_rows):
    """cam_rows (B, T) fp32 += one layer's GETAM row of EVERY sample of the batch, one launch (acr_getam_rows_accum)."""
    B, T, _ = qkv.shape
    lib = L.load()
    qp, kp, vp = _qkv_ptrs(qkv, heads)
    assert cam_rows.shape == (B, T) and cam_rows.stride(1) == 1
    L.check(lib.acr_getam_rows_accum(_desc(B, heads, T, qkv.dtype), qp, kp, vp, L.ptr(d_o), L.ptr(lse2), L.GETAM_FUNCS[func],
                                     L.ptr(cam_rows), cam_rows.stride(0), L.stream_ptr()), "acr_getam_rows_accum")


def linear_bf16(x, weight, bias=None, resid=None, out=None):
    """y = x @ weight.T (+ bias) (+ resid) on the hand-written bf16 MFMA GEMM (acr_linear_bf16).
    x (M,K), weight (N,K), bias (N,), resid (M,N), all bf16 with unit inner stride."""
    L.require_gpu(x, weight)
    M, K = x.shape
    N = weight.shape[0]
    assert x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.stride(1) == 1 and weight.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    tok = _t0("linear_bf16", M, N, K)
    L.check(L.load().acr_linear_bf16(L.ptr(x), x.stride(0), L.ptr(weight), weight.stride(0), L.ptr(bias),
                                     L.ptr(resid), resid.stride(0) if resid is not None else 0, L.ptr(out),
                                     out.stride(0), M, N, K, L.stream_ptr()), "acr_linear_bf16")
    _t1(tok)
    return out


def colsum_bf16(dy2):
    """Column sums of a (M, N) bf16 matrix -> (N,) bf16 (bias gradient), deterministic two-stage HIP reduction."""
    M, N = dy2.shape
    lib = L.load()
    if N % 8 or dy2.stride(1) != 1 or dy2.stride(0) % 8 or dy2.data_ptr() % 16:
        return dy2.sum(dim=0)
    ws = torch.empty(lib.acr_colsum_ws_floats(M, N), dtype=torch.float32, device=dy2.device)
    out = torch.empty(N, dtype=torch.bfloat16, device=dy2.device)
    L.check(lib.acr_colsum_bf16(L.ptr(dy2), dy2.stride(0), M, N, L.ptr(ws), L.ptr(out), L.stream_ptr()), "acr_colsum_bf16")
    return out


def wgrad_bf16(dy2, x2):
    """dW = dy2^T @ x2 ((M,N),(M,K) bf16 -> (N,K) bf16) on the hand-written split-M TN GEMM; shapes it does not
    cover (N or K not a multiple of 128) go to torch.mm."""
    M, N = dy2.shape
    K = x2.shape[1]
    lib = L.load()
    nws = lib.acr_wgrad_ws_floats(M, N, K)
    if (nws == 0 or dy2.stride(1) != 1 or x2.stride(1) != 1 or dy2.stride(0) % 8 or x2.stride(0) % 8
            or dy2.data_ptr() % 16 or x2.data_ptr() % 16):
        return torch.mm(dy2.t(), x2)
    ws = torch.empty(nws, dtype=torch.float32, device=dy2.device)
    dw = torch.empty((N, K), dtype=torch.bfloat16, device=dy2.device)
    tok = _t0("wgrad_bf16", M, N, K)
    L.check(lib.acr_wgrad_bf16(L.ptr(dy2), dy2.stride(0), L.ptr(x2), x2.stride(0), M, N, K, L.ptr(ws), L.ptr(dw),
                               L.stream_ptr()), "acr_wgrad_bf16")
    _t1(tok)
    return dw


FUSED_BIAS_GRAD = True      # A/B switch


def wgrad_bias_bf16(dy2, x2):
    """(dW, db) = (dy2^T @ x2, column sums of dy2) in ONE sweep over dy2 (acr_wgrad_bias_bf16: the bias gradient is
    accumulated from the dY fragments inside the weight-gradient kernel); unsupported shapes take the two separate ops."""
    M, N = dy2.shape
    K = x2.shape[1]
    lib = L.load()
    nws = lib.acr_wgrad_bias_ws_floats(M, N, K) if FUSED_BIAS_GRAD else 0
    if (nws == 0 or N % 8 or dy2.stride(1) != 1 or x2.stride(1) != 1 or dy2.stride(0) % 8 or x2.stride(0) % 8
            or dy2.data_ptr() % 16 or x2.data_ptr() % 16):
        return wgrad_bf16(dy2, x2), colsum_bf16(dy2)
    ws = torch.empty(nws, dtype=torch.float32, device=dy2.device)
    dw = torch.empty((N, K), dtype=torch.bfloat16, device=dy2.device)
    db = torch.empty(N, dtype=torch.bfloat16, device=dy2.device)
    tok = _t0("wgrad_bf16", M, N, K)
    L.check(lib.acr_wgrad_bias_bf16(L.ptr(dy2), dy2.stride(0), L.ptr(x2), x2.stride(0), M, N, K, L.ptr(ws), L.ptr(dw), L.ptr(db),
                                    L.stream_ptr()), "acr_wgrad_bias_bf16")
    _t1(tok)
    return dw, db


class WeightTransposes:
    """(in, out) copies of Linear weights for the input-gradient GEMMs, refreshed in ONE launch (acr_transpose_many_bf16 /
    _f32; in fp32 the GEMM on the transposed copy is 12-17 % faster than on W as stored, scripts/lab/gemm_nn_vs_nt.py).

    ``refresh()`` is called by the training step's owner right after the optimizer step; a copy is used only while the
    weight's autograd version, storage address and device equal those recorded at refresh (an in-place update through the
    parameter, a re-allocated storage or ``module.to(other_gpu)`` make the consumer fall back to W as stored / a transpose
    on the fly).  What this CANNOT see: writes through ``.data`` (``p.data.mul_()``, ``p.data.copy_()``, an EMA swap) --
    they do not move the parameter's version counter -- so code that updates weights that way must call ``refresh()``
    itself (train.refresh_weight_transposes(model)); INTEGRATION.md says so."""

    def __init__(self, modules, dtype=torch.bfloat16):
        import numpy as np
        mult = 8 if dtype == torch.bfloat16 else 4
        self.dtype = dtype
        self.lins = [m for m in modules if isinstance(m, torch.nn.Linear) and m.weight.is_cuda and m.weight.dtype == dtype
                     and m.weight.is_contiguous() and m.weight.shape[0] % mult == 0 and m.weight.shape[1] % mult == 0
                     and m.weight.shape[0] >= 64 and m.weight.requires_grad]
        self.ok = bool(self.lins)
        if not self.ok:
            return
        dev = self.lins[0].weight.device
        rec = np.zeros(len(self.lins), dtype=[("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("tile0", "<i4"),
                                              ("tiles_c", "<i4")])
        bt, t0 = [], 0
        self.bufs = []
        for i, m in enumerate(self.lins):
            rows, cols = m.weight.shape
            wt = torch.empty((cols, rows), dtype=dtype, device=dev)
            self.bufs.append(wt)
            tr, tc = (rows + 63) // 64, (cols + 63) // 64
            rec[i] = (m.weight.data_ptr(), wt.data_ptr(), rows, cols, t0, tc)
            bt.append(np.full(tr * tc, i, dtype=np.int32))
            t0 += tr * tc
        self._ptrs = [m.weight.data_ptr() for m in self.lins]
        self.table = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        self.blk = torch.from_numpy(np.concatenate(bt)).to(dev)
        self.nblocks = t0

    @torch.no_grad()
    def refresh(self):
        if not self.ok:
            return
        if any(m.weight.data_ptr() != p for m, p in zip(self.lins, self._ptrs)):
            self.ok = False                                   # storages were re-allocated: stop serving copies
            for m in self.lins:
                m._acr_wt = None
            return
        fn = L.load().acr_transpose_many_bf16 if self.dtype == torch.bfloat16 else L.load().acr_transpose_many_f32
        L.check(fn(L.ptr(self.table), L.ptr(self.blk), self.nblocks, L.stream_ptr()), "acr_transpose_many")
        for m, wt in zip(self.lins, self.bufs):
            m._acr_wt, m._acr_wt_ver, m._acr_wt_ptr = wt, m.weight._version, m.weight.data_ptr()


def weight_t(weight, owner=None, make=True):
    """weight^T contiguous: the cached copy of ``owner`` (an nn.Linear) when it is current, else a fresh transpose
    (``make`` = False: None instead, for callers that can work on W as stored)."""
    wt = getattr(owner, "_acr_wt", None) if owner is not None else None
    if (wt is not None and owner._acr_wt_ver == weight._version and wt.shape[0] == weight.shape[1] and wt.dtype == weight.dtype
            and wt.device == weight.device and owner.weight.data_ptr() == weight.data_ptr()
            and getattr(owner, "_acr_wt_ptr", None) == weight.data_ptr()):
        return wt
    return weight.t().contiguous() if make else None


F32_WT = True      # A/B switch: fp32 input gradients on the cached W^T (NT) vs W as stored (NN)


def _dx_f32(dy2, weight, owner, out, aux=None, act=0, math=0):
    """out = dy2 W (optionally * GELU'(aux)): NT on the cached (in, out) copy when it is current, else NN on W as stored."""
    wt = weight_t(weight, owner, make=False) if F32_WT else None
    if wt is not None:
        return gemm_f32_raw("nt", dy2, wt, out, aux=aux, act=act, math=math)
    return gemm_f32_raw("nn", dy2, weight, out, aux=aux, act=act, math=math)


class LinearBf16Fn(Function):
    """y = x W^T + b (+ resid) for the attention block's qkv / proj Linears in the bf16 mode, on the hand-written
    MFMA GEMM for forward and input gradient; the weight gradient (a reduction over all tokens) stays on
    hipBLASLt through torch.mm."""

    @staticmethod
    def forward(ctx, x, weight, bias, resid, hip_dx=True, hip_dw=True, hip_fwd=True, owner=None):
        ctx.hip_dx, ctx.hip_dw, ctx.owner = hip_dx, hip_dw, owner
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        r2 = resid.reshape(-1, weight.shape[0]) if resid is not None else None
        if hip_fwd:
            y = linear_bf16(x2, weight, bias, r2)
        else:                                               # library GEMM forward, hand-written backward pieces
            y = torch.nn.functional.linear(x2, weight, bias)
            if r2 is not None:
                y.add_(r2)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias, ctx.has_resid = bias is not None, resid is not None
        return y.reshape(*shp[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if ctx.hip_dx and weight.shape[0] % 64 == 0:
                dx = linear_bf16(dy2, weight_t(weight, ctx.owner))
            else:                                           # contraction length not a multiple of the K tile
                dx = torch.mm(dy2, weight)
            dx = dx.reshape(*dy.shape[:-1], weight.shape[1])
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and ctx.hip_dw and want_db:
            dw, db = wgrad_bias_bf16(dy2, x2)               # one sweep over dy for both
        else:
            if ctx.needs_input_grad[1]:
                dw = wgrad_bf16(dy2, x2) if ctx.hip_dw else torch.mm(dy2.t(), x2)
            if want_db:
                db = colsum_bf16(dy2)
        return dx, dw, db, (dy if ctx.has_resid else None), None, None, None, None


def mlp_fusable(x, fc1, fc2):
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and fc1.weight.dtype == torch.bfloat16
            and fc1.bias is not None and fc2.bias is not None and fc1.weight.shape[1] % 64 == 0
            and fc1.weight.shape[0] % 64 == 0 and fc2.weight.shape[0] % 64 == 0)


class MlpFn(Function):
    """y = fc2(GELU(fc1(x))) [+ resid] of a transformer block (models/vision_transformer.py:158-164) with the activation
    inside the GEMM epilogues: fc1 writes h and GELU(h) in one pass, and the input gradient of fc2 comes out already
    multiplied by GELU'(h).  Replaces the separate GELU forward / backward passes over the (tokens x 3072) activations."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, resid, fc1=None, fc2=None):
        ctx.fc1, ctx.fc2 = fc1, fc2
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        M, K = x2.shape
        Hd = w1.shape[0]
        lib = L.load()
        h = torch.empty((M, Hd), dtype=torch.bfloat16, device=x.device)
        a = torch.empty((M, Hd), dtype=torch.bfloat16, device=x.device)
        tok = _t0("linear_bf16", M, Hd, K)
        L.check(lib.acr_linear_gelu_bf16(L.ptr(x2), x2.stride(0), L.ptr(w1), w1.stride(0), L.ptr(b1), L.ptr(h), L.ptr(a), Hd,
                                         M, Hd, K, L.stream_ptr()), "acr_linear_gelu_bf16")
        _t1(tok)
        r2 = resid.reshape(-1, w2.shape[0]) if resid is not None else None
        y = linear_bf16(a, w2, b2, r2)
        ctx.save_for_backward(x2, h, a, w1, w2)
        ctx.has_resid = resid is not None
        return y.reshape(*shp[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, h, a, w1, w2 = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        M, Hd = h.shape
        lib = L.load()
        need = ctx.needs_input_grad                             # (x, w1, b1, w2, b2, resid, ...): CAM inference wants dx only
        dw1 = db1 = dw2 = db2 = None
        if need[3] and need[4]:
            dw2, db2 = wgrad_bias_bf16(dy2, a)
        elif need[3]:
            dw2 = wgrad_bf16(dy2, a)
        elif need[4]:
            db2 = colsum_bf16(dy2)
        w2t = weight_t(w2, ctx.fc2)                           # (hidden, out): dA = dY W2
        dh = torch.empty_like(h)
        L.check(lib.acr_linear_dgelu_bf16(L.ptr(dy2), dy2.stride(0), L.ptr(w2t), w2t.stride(0), L.ptr(h), h.stride(0), L.ptr(dh),
                                          dh.stride(0), M, Hd, dy2.shape[1], L.stream_ptr()), "acr_linear_dgelu_bf16")
        if need[1] and need[2]:
            dw1, db1 = wgrad_bias_bf16(dh, x2)
        elif need[1]:
            dw1 = wgrad_bf16(dh, x2)
        elif need[2]:
            db1 = colsum_bf16(dh)
        dx = None
        if need[0]:
            dx = linear_bf16(dh, weight_t(w1, ctx.fc1)).reshape(*dy.shape[:-1], w1.shape[1])
        return dx, dw1, db1, dw2, db2, (dy if ctx.has_resid else None), None, None


def mlp(x, fc1, fc2, resid=None):
    return MlpFn.apply(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias, resid, fc1, fc2)


# ------------------------------------------------------------------------------------------------
# fp32 (reference precision) Linears on the exact-fp32 MFMA GEMM (acr_gemm_f32)
# ------------------------------------------------------------------------------------------------
GEMM_MODES = {"nt": 0, "nn": 1, "tn": 2}


def gemm_f32_raw(mode, a, b, c, bias=None, aux=None, act=0, c2=None, colsum=None, math=0):
    """c = op(a) op(b) through acr_gemm_f32 (see include/acr_hip.h): 'nt' c[M,N] = a[M,K] b[N,K]^T, 'nn' c = a[M,K] b[K,N],
    'tn' c = a[K,M]^T b[K,N] (+ colsum[M] = column sums of a).  fp32, unit inner strides.  ``math``: _lib.MATH code (0 = exact-fp32
    MFMA, 1 = six bf16-MFMA terms of a three-way operand split), a per-call argument."""
    lib = L.load()
    md = GEMM_MODES[mode]
    M, N = c.shape
    K = a.shape[0] if md == 2 else a.shape[1]
    nws = lib.acr_gemm_f32_ws_floats(md, math, M, N, K)
    ws = torch.empty(nws, dtype=torch.float32, device=a.device) if nws else None
    tok = _t0("gemm_f32_" + mode, M, N, K)
    L.check(lib.acr_gemm_f32(md, math, act, L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), L.ptr(bias), L.ptr(aux),
                             aux.stride(0) if aux is not None else 0, L.ptr(c), c.stride(0), L.ptr(c2), L.ptr(colsum), M, N, K,
                             L.ptr(ws), L.stream_ptr()), "acr_gemm_f32")
    _t1(tok)
    return c


X3_IMAGES = True      # A/B: split-product Linears keep operand images across forward / backward


def x3_image(x2, colsum=None):
    """Split-product image (include/acr_hip.h "split-product images") of the fp32 matrix x2 (unit inner stride); ``colsum``
    (cols) receives x2's column sums from the same pass."""
    lib = L.load()
    rows, cols = x2.shape
    img = torch.empty(lib.acr_x3_image_floats(rows, cols), dtype=torch.float32, device=x2.device)
    cws = torch.empty(lib.acr_x3_colsum_ws_floats(rows, cols), dtype=torch.float32, device=x2.device) if colsum is not None else None
    L.check(lib.acr_x3_image(L.ptr(x2), x2.stride(0), rows, cols, L.ptr(img), L.ptr(colsum), L.ptr(cws), L.stream_ptr()), "acr_x3_image")
    return img


def x3_image_t(x2):
    """Image of x2's transpose."""
    lib = L.load()
    rows, cols = x2.shape
    img = torch.empty(lib.acr_x3_image_floats(cols, rows), dtype=torch.float32, device=x2.device)
    L.check(lib.acr_x3_image_t(L.ptr(x2), x2.stride(0), rows, cols, L.ptr(img), L.stream_ptr()), "acr_x3_image_t")
    return img


def x3_image_many(specs, device):
    """Split-product images of MANY small operands in one launch (acr_x3_image_many): ``specs`` is a list of
    (src tensor, element offset, rows, K, sr, kin, sko, ski) -- element (r, k) of the rows x K operand is
    src[offset + r*sr + (k // kin)*sko + (k % kin)*ski] (include/acr_hip.h).  Returns the images, views of ONE buffer."""
    import numpy as np
    lib = L.load()
    if torch.cuda.is_current_stream_capturing():
        # the descriptor table below travels through a temporary pinned buffer: a captured copy node would re-read that (by then
        # recycled) host memory on every replay and the images would silently be garbage.  Captures must HIT the
        # caches the warm-up passes filled; a miss falls back to eager launches loudly (backbone.pass_graph / prefix_graph)
        raise RuntimeError("acr_x3_image_many inside a hipGraph capture: its descriptor table is a one-shot host upload")
    sizes = [int(lib.acr_x3_image_floats(rows, K)) for (_, _, rows, K, _, _, _, _) in specs]
    buf = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    rec = np.zeros(len(specs), dtype=[("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("K", "<i4"), ("sr", "<i4"), ("kin", "<i4"),
                                      ("sko", "<i4"), ("ski", "<i4"), ("wg0", "<i4"), ("nkb", "<i4")])
    blk, wg0, off, imgs = [], 0, 0, []
    for i, ((src, eoff, rows, K, sr, kin, sko, ski), n) in enumerate(zip(specs, sizes)):
        assert src.dtype == torch.float32 and K % 8 == 0 and kin % 8 == 0, (K, kin)
        nkb = (K + 15) // 16
        nwg = ((rows + 127) // 128) * ((nkb + 3) // 4)
        rec[i] = (src.data_ptr() + 4 * eoff, buf.data_ptr() + 4 * off, rows, K, sr, kin, sko, ski, wg0, nkb)
        blk.append(np.full(nwg, i, dtype=np.int32))
        imgs.append(buf[off:off + n])
        wg0 += nwg
        off += n
    raw = np.concatenate([rec.view(np.uint8), np.concatenate(blk).view(np.uint8)])
    host = torch.empty(raw.size, dtype=torch.uint8, pin_memory=True)          # pinned + asynchronous: no host-side stream drain
    host.numpy()[:] = raw
    table = host.to(device, non_blocking=True)
    nb = rec.nbytes
    L.check(lib.acr_x3_image_many(L.ptr(table), L.c_void_p(table.data_ptr() + nb), wg0, L.stream_ptr()), "acr_x3_image_many")
    for im in imgs:                                         # the launch reads the table asynchronously: it lives as long as any image
        im._acr_keep = table
    return imgs


def weight_image(weight, owner=None, transposed=False):
    """Image of a Linear's weight (``transposed``: of W^T, what the input gradient multiplies by), cached on ``owner`` (the
    nn.Linear) per weight version: inference makes each image once, a training step once per forward resp. backward (the
    optimizers move the version counters on, train.py).  The cache entry remembers the stream that built the image and an
    event behind the build: a consumer on ANOTHER stream (CAM generation runs each scale on a stream of its own) waits for
    that event first -- a cache hit must not read an image whose split pass is still queued elsewhere.  What the key cannot see
    -- as with WeightTransposes -- are writes through ``.data``: ``invalidate_weight_images(model)`` /
    ``train.refresh_weight_transposes(model)`` drop the cached images."""
    key = "_acr_x3_wt_img" if transposed else "_acr_x3_w_img"
    c = getattr(owner, key, None) if owner is not None else None
    if c is not None and c[0] == weight._version and c[1] == weight.data_ptr() and c[2].device == weight.device:
        cur = torch.cuda.current_stream(weight.device)
        if c[3] != cur and not torch.cuda.is_current_stream_capturing():
            cur.wait_event(c[4])
        return c[2]
    w2 = weight.detach()
    img = x3_image_t(w2) if transposed else x3_image(w2)
    if owner is not None and not torch.cuda.is_current_stream_capturing():      # a captured pass would only be run at replay
        cur = torch.cuda.current_stream(weight.device)
        ev = torch.cuda.Event()
        ev.record(cur)
        setattr(owner, key, (weight._version, weight.data_ptr(), img, cur, ev))
    return img


def prebuild_weight_images(linears):
    """The images of W and of W^T of every given nn.Linear in ONE launch (ops.x3_image_many), stored in the per-owner caches
    weight_image() reads -- a training step calls this once after the optimizer step instead of paying 8 small image launches per
    block when the forward / backward first ask for them (train.refresh_weight_transposes)."""
    lins = [m for m in linears if m.weight.is_cuda and m.weight.dtype == torch.float32 and m.weight.is_contiguous()
            and m.weight.shape[0] % 8 == 0 and m.weight.shape[1] % 8 == 0]
    if not lins:
        return 0
    specs = []
    for m in lins:
        n, k = m.weight.shape
        w = m.weight.detach()
        specs.append((w, 0, n, k, k, k, 0, 1))              # W (out x in): the forward's B operand
        specs.append((w, 0, k, n, 1, n, 0, k))              # W^T (in x out): the input gradient's
    dev = lins[0].weight.device
    imgs = x3_image_many(specs, dev)
    cur = torch.cuda.current_stream(dev)
    ev = torch.cuda.Event()
    ev.record(cur)
    for i, m in enumerate(lins):
        m._acr_x3_w_img = (m.weight._version, m.weight.data_ptr(), imgs[2 * i], cur, ev)
        m._acr_x3_wt_img = (m.weight._version, m.weight.data_ptr(), imgs[2 * i + 1], cur, ev)
    return len(lins)


def invalidate_weight_images(model):
    """Drop every cached split-product weight image under ``model`` (after a weight was written through ``.data``, which moves
    neither the version counter nor the address the cache is keyed on)."""
    for m in model.modules():
        for key in ("_acr_x3_w_img", "_acr_x3_wt_img"):
            if key in m.__dict__:
                delattr(m, key)


def gemm_x3(mode, a_img, b_img, c, K, bias=None, aux=None, act=0, c2=None, colsum=None, shape=None):
    """c[M,N] from split-product images through acr_gemm_x3: 'nt' c = A[M,K] B[N,K]^T (images of A and B), 'tn' c = A[K,M]^T B[K,N]
    (images of A and B as stored, K rows each).  act 3 / 4: the output leaves as an image in ``c2`` (include/acr_hip.h); act 4
    has no fp32 output (``c`` = None, ``shape`` = (M, N)) and gives its column sums in ``colsum``."""
    lib = L.load()
    md = GEMM_MODES[mode]
    M, N = c.shape if c is not None else shape
    dev = a_img.device
    nws = lib.acr_gemm_x3_ws_floats(md, act, M, N, K)
    ws = torch.empty(nws, dtype=torch.float32, device=dev) if nws else None
    tok = _t0("gemm_x3_" + mode, M, N, K)
    L.check(lib.acr_gemm_x3(md, act, L.ptr(a_img), L.ptr(b_img), L.ptr(bias), L.ptr(aux), aux.stride(0) if aux is not None else 0, L.ptr(c),
                            c.stride(0) if c is not None else N, L.ptr(c2), L.ptr(colsum), M, N, K, L.ptr(ws), L.stream_ptr()), "acr_gemm_x3")
    _t1(tok)
    return c


X3_IMAGE_EPILOGUES = True      # A/B: the MLP's 4x-wide tensors leave their GEMMs as images


def x3_image_empty(rows, cols, device):
    return torch.empty(L.load().acr_x3_image_floats(rows, cols), dtype=torch.float32, device=device)


def _f32_ok(*ts):
    return all(t is None or (t.dtype == torch.float32 and t.is_cuda and t.stride(-1) == 1 and (t.dim() == 1 or t.stride(0) % 4 == 0)
                             and t.data_ptr() % 16 == 0) for t in ts)


def linear_f32_usable(x, weight):
    K = x.shape[-1]
    N = weight.shape[0]
    return (F32_HIP_LINEAR and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.is_contiguous()
            and weight.is_contiguous() and K % 4 == 0 and N % 4 == 0 and K >= 32 and N >= 32 and x.numel() // K >= 1)


F32_HIP_LINEAR = True      # A/B switch: fp32 Linears on acr_gemm_f32 vs hipBLASLt


class LinearF32Fn(Function):
    """y = x W^T + b (+ resid) in fp32 on acr_gemm_f32: forward NT, input gradient NN (W as stored), weight + bias gradient
    in one TN sweep over dy (models/vision_transformer.py:200,212 and their autograd backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, resid, owner=None, math=0, x_image=None):
        ctx.math = math
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]) if x_image is None else None      # x_image: x exists only as its image (layer_norm_image)
        N = weight.shape[0]
        r2 = resid.reshape(-1, N) if resid is not None else None
        if r2 is not None and not r2.is_contiguous():
            r2 = r2.contiguous()
        M = x.numel() // shp[-1]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        ctx.images = math == 1 and X3_IMAGES and _f32_ok(x2, weight, y, r2, bias)
        assert ctx.images or x_image is None, "a LayerNorm image needs the split-product Linear behind it (ops.ln_image_usable)"
        if ctx.images:                                      # x's image serves this product and the weight gradient
            xi = x_image if x_image is not None else x3_image(x2)
            gemm_x3("nt", xi, weight_image(weight, owner), y, shp[-1], bias=bias, aux=r2)
            ctx.save_for_backward(xi if any(ctx.needs_input_grad[1:3]) else None, weight)
            ctx.M = M
        else:
            gemm_f32_raw("nt", x2, weight, y, bias=bias, aux=r2, math=math)
            ctx.save_for_backward(x2, weight)
        ctx.has_bias, ctx.has_resid, ctx.owner = bias is not None, resid is not None, owner
        return y.reshape(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        N, K = weight.shape
        dy2 = dy.reshape(-1, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.images:                                      # dy's image serves the input and the weight gradient (+ its column sums = db)
            if not _f32_ok(dy2):
                dy2 = dy2.clone()                           # a fresh, aligned allocation
            xi = x2
            db = torch.empty(N, dtype=torch.float32, device=dy.device) if want_db else None
            dyi = x3_image(dy2, colsum=db)
            if ctx.needs_input_grad[0]:
                dx = torch.empty((ctx.M, K), dtype=torch.float32, device=dy.device)
                gemm_x3("nt", dyi, weight_image(weight, ctx.owner, True), dx, N)
                dx = dx.reshape(*dy.shape[:-1], K)
            if ctx.needs_input_grad[1]:
                dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
                gemm_x3("tn", dyi, xi, dw, ctx.M)
            return dx, dw, db, (dy if ctx.has_resid else None), None, None, None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((dy2.shape[0], K), dtype=torch.float32, device=dy.device)
            _dx_f32(dy2, weight, ctx.owner, dx, math=ctx.math)
            dx = dx.reshape(*dy.shape[:-1], K)
        if ctx.needs_input_grad[1]:
            dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
            db = torch.empty(N, dtype=torch.float32, device=dy.device) if want_db else None
            gemm_f32_raw("tn", dy2, x2, dw, colsum=db, math=ctx.math)
        elif want_db:
            db = dy2.sum(0)
        return dx, dw, db, (dy if ctx.has_resid else None), None, None, None


class MlpF32Fn(Function):
    """y = fc2(GELU(fc1(x))) [+ resid] in fp32 (models/vision_transformer.py:158-164) with the activation in the GEMM
    epilogues: fc1 writes h and GELU(h) in one pass; fc2's input gradient comes out multiplied by GELU'(h)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, resid, fc1=None, fc2=None, math=0, x_image=None):
        ctx.fc1, ctx.fc2, ctx.math = fc1, fc2, math
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]) if x_image is None else None      # x_image: x exists only as its image (layer_norm_image)
        M = x.numel() // shp[-1]
        Hd, D = w1.shape[0], w2.shape[0]
        h = torch.empty((M, Hd), dtype=torch.float32, device=x.device)
        r2 = resid.reshape(-1, D) if resid is not None else None
        if r2 is not None and not r2.is_contiguous():
            r2 = r2.contiguous()
        y = torch.empty((M, D), dtype=torch.float32, device=x.device)
        ctx.images = math == 1 and X3_IMAGES and _f32_ok(x2, w1, w2, r2, b1, b2)
        ctx.has_resid = resid is not None
        assert ctx.images or x_image is None, "a LayerNorm image needs the split-product MLP behind it (ops.ln_image_usable)"
        if ctx.images:                                      # the images of x and GELU(h) serve the forward and the weight gradients
            xi = x_image if x_image is not None else x3_image(x2)
            ctx.epi = X3_IMAGE_EPILOGUES and Hd % 8 == 0
            if ctx.epi:                                     # GELU(h) leaves fc1 as fc2's operand image; it never exists in fp32
                ai = x3_image_empty(M, Hd, x.device)
                gemm_x3("nt", xi, weight_image(w1, fc1), h, shp[-1], bias=b1, act=3, c2=ai)
            else:
                a = torch.empty((M, Hd), dtype=torch.float32, device=x.device)
                gemm_x3("nt", xi, weight_image(w1, fc1), h, shp[-1], bias=b1, act=1, c2=a)
                ai = x3_image(a)
                del a
            gemm_x3("nt", ai, weight_image(w2, fc2), y, Hd, bias=b2, aux=r2)
            ctx.save_for_backward(xi, h, ai, w1, w2)        # fp32 GELU(h) is not kept: its image is all the backward reads
            return y.reshape(*shp[:-1], D)
        a = torch.empty((M, Hd), dtype=torch.float32, device=x.device)
        gemm_f32_raw("nt", x2, w1, h, bias=b1, act=1, c2=a, math=math)          # a = GELU(h), and GELU'(h) in place of h (all backward needs)
        gemm_f32_raw("nt", a, w2, y, bias=b2, aux=r2, math=math)
        ctx.save_for_backward(x2, h, a, w1, w2)
        return y.reshape(*shp[:-1], D)

    @staticmethod
    def backward(ctx, dy):
        x2, h, a, w1, w2 = ctx.saved_tensors
        need = ctx.needs_input_grad
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        M, Hd = h.shape
        dev = dy.device
        math = ctx.math
        dw1 = db1 = dw2 = db2 = dx = None
        if ctx.images:
            if not _f32_ok(dy2):
                dy2 = dy2.clone()                           # a fresh, aligned allocation
            xi, ai = x2, a
            D = w2.shape[0]
            db2 = torch.empty(D, dtype=torch.float32, device=dev) if need[4] else None
            dyi = x3_image(dy2, colsum=db2)
            if need[3]:
                dw2 = torch.empty_like(w2)
                gemm_x3("tn", dyi, ai, dw2, M)
            db1 = torch.empty(Hd, dtype=torch.float32, device=dev) if need[2] else None
            if ctx.epi:                                     # (dY W2) * GELU'(h) leaves the product as fc1's dy image (+ its column sums)
                dhi = x3_image_empty(M, Hd, dev)
                gemm_x3("nt", dyi, weight_image(w2, ctx.fc2, True), None, D, aux=h, act=4, c2=dhi, colsum=db1, shape=(M, Hd))
            else:
                dh = torch.empty_like(h)
                gemm_x3("nt", dyi, weight_image(w2, ctx.fc2, True), dh, D, aux=h, act=2)          # `h` holds GELU'(h) (see forward)
                dhi = x3_image(dh, colsum=db1)
            if need[1]:
                dw1 = torch.empty_like(w1)
                gemm_x3("tn", dhi, xi, dw1, M)
            if need[0]:
                dx = torch.empty((M, w1.shape[1]), dtype=torch.float32, device=dev)
                gemm_x3("nt", dhi, weight_image(w1, ctx.fc1, True), dx, Hd)
                dx = dx.reshape(*dy.shape[:-1], w1.shape[1])
            return dx, dw1, db1, dw2, db2, (dy if ctx.has_resid else None), None, None, None, None
        if need[3]:
            dw2 = torch.empty_like(w2)
            db2 = torch.empty(w2.shape[0], dtype=torch.float32, device=dev) if need[4] else None
            gemm_f32_raw("tn", dy2, a, dw2, colsum=db2, math=math)
        elif need[4]:
            db2 = dy2.sum(0)
        dh = torch.empty_like(h)
        _dx_f32(dy2, w2, ctx.fc2, dh, aux=h, act=2, math=math)                   # (dY W2) * GELU'(h); `h` holds GELU'(h) (see forward)
        if need[1]:
            dw1 = torch.empty_like(w1)
            db1 = torch.empty(Hd, dtype=torch.float32, device=dev) if need[2] else None
            gemm_f32_raw("tn", dh, x2, dw1, colsum=db1, math=math)
        elif need[2]:
            db1 = dh.sum(0)
        if need[0]:
            dx = torch.empty_like(x2)
            _dx_f32(dh, w1, ctx.fc1, dx, math=math)
            dx = dx.reshape(*dy.shape[:-1], w1.shape[1])
        return dx, dw1, db1, dw2, db2, (dy if ctx.has_resid else None), None, None, None, None


def mlp_f32_usable(x, fc1, fc2):
    return (linear_f32_usable(x, fc1.weight) and fc2.weight.dtype == torch.float32 and fc1.bias is not None
            and fc2.bias is not None and fc2.weight.is_contiguous() and fc2.weight.shape[0] % 4 == 0)


def mlp_f32(x, fc1, fc2, resid=None, math=0, x_image=None):
    return MlpF32Fn.apply(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias, resid, fc1, fc2, math, x_image)


def linear_or_hip(x, lin, resid=None, use_hip=True, hip_dx=True, hip_dw=True, hip_fwd=True, math=0, x_image=None):
    """nn.Linear forward; bf16 CUDA tensors with K % 64 == 0 take the hand-written GEMM (resid fused).
    ``hip_dx`` = False leaves the input gradient on hipBLASLt (shapes where the library kernel is faster).
    ``x_image``: x came out of layer_norm_image -- it exists only as that split-product image."""
    if x_image is not None:
        return LinearF32Fn.apply(x, lin.weight, lin.bias, resid, lin, math, x_image)
    if (use_hip and x.is_cuda and x.dtype == torch.bfloat16 and lin.weight.dtype == torch.bfloat16
            and lin.weight.shape[1] % 64 == 0 and x.is_contiguous()):
        return LinearBf16Fn.apply(x, lin.weight, lin.bias, resid, hip_dx, hip_dw, hip_fwd, lin)
    if use_hip and linear_f32_usable(x, lin.weight) and not torch.is_autocast_enabled():
        return LinearF32Fn.apply(x, lin.weight, lin.bias, resid, lin, math)
    y = torch.nn.functional.linear(x, lin.weight, lin.bias)
    return y if resid is None else resid + y


class MaxPoolSameFn(Function):
    """3x3/2 max-pool with TF-SAME -inf padding (pad_top, pad_left given, ph/pw = total padding)."""

    @staticmethod
    def forward(ctx, x, pt, pl, ph, pw):
        x = x.contiguous()
        N, C, H, W = x.shape
        Ho, Wo = (H + ph - 3) // 2 + 1, (W + pw - 3) // 2 + 1
        y = torch.empty((N, C, Ho, Wo), dtype=x.dtype, device=x.device)
        amax = torch.empty((N, C, Ho, Wo), dtype=torch.uint8, device=x.device)
        fn = L.load().acr_maxpool3x3s2_fwd_f32 if x.dtype == torch.float32 else L.load().acr_maxpool3x3s2_fwd_bf16
        L.check(fn(L.ptr(x), L.ptr(y), L.ptr(amax), N * C, H, W, Ho, Wo, pt, pl, L.stream_ptr()), "acr_maxpool3x3s2_fwd")
        ctx.save_for_backward(amax)
        ctx.geom = (N, C, H, W, Ho, Wo, pt, pl)
        return y

    @staticmethod
    def backward(ctx, dy):
        (amax,) = ctx.saved_tensors
        N, C, H, W, Ho, Wo, pt, pl = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty((N, C, H, W), dtype=dy.dtype, device=dy.device)
        fn = L.load().acr_maxpool3x3s2_bwd_f32 if dy.dtype == torch.float32 else L.load().acr_maxpool3x3s2_bwd_bf16
        L.check(fn(L.ptr(dy), L.ptr(amax), L.ptr(dx), N * C, H, W, Ho, Wo, pt, pl, L.stream_ptr()), "acr_maxpool3x3s2_bwd")
        return dx, None, None, None, None


def maxpool3x3s2_same(x, pt, pl, ph, pw):
    return MaxPoolSameFn.apply(x, pt, pl, ph, pw)


class Subsample2Fn(Function):
    """x[:, :, ::2, ::2] as a contiguous tensor (what a stride-2 1x1 convolution reads) -- one pass, and one pass back (the even
    pixels of dx, zeros elsewhere) instead of autograd's two zero fills + two strided copies."""

    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        y = torch.empty((N, C, (H + 1) // 2, (W + 1) // 2), dtype=torch.float32, device=x.device)
        L.check(L.load().acr_subsample2_fwd_f32(L.ptr(x), L.ptr(y), N * C, H, W, L.stream_ptr()), "acr_subsample2_fwd_f32")
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W = ctx.shape
        dy = (dy if dy.dtype == torch.float32 else dy.float()).contiguous()
        dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
        L.check(L.load().acr_subsample2_bwd_f32(L.ptr(dy), L.ptr(dx), N * C, H, W, L.stream_ptr()), "acr_subsample2_bwd_f32")
        return dx


def subsample2(x):
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and not torch.is_autocast_enabled():
        return Subsample2Fn.apply(x)
    return x[:, :, ::2, ::2].contiguous()


F32_HIP_CONV1X1 = True      # A/B switch: fp32 1x1 convolutions on acr_conv1x1_f32 vs MIOpen


def conv1x1_fusable(x, weight, stride):
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and weight.dtype == x.dtype and x.dim() == 4 and x.is_contiguous()):
        return False
    N, C, H, W = x.shape
    co, ci = weight.shape[0], weight.shape[1]
    if x.dtype == torch.float32 and (not F32_HIP_CONV1X1 or torch.is_autocast_enabled()):
        return False
    return (stride == 1 and weight.shape[2] == 1 and weight.shape[3] == 1 and ci % 64 == 0 and co % 64 == 0
            and (H * W) % 8 == 0)


CONV1X1_WIMG = True      # A/B: split-product 1x1 convolutions with the weight as an image


def _conv1x1_f32_launch(math, w2, w_transposed, x, addend, y, N, co, ci, hw, img=None):
    """y[n] (co x hw) = W . x[n] (+ addend).  w_transposed = 0: w2 is (co, ci); 1: w2 is the forward's (ci, co) weight (input gradient).
    ``img``: the weight's split-product image when the caller has it already (the stem makes all of them in one launch)."""
    lib = L.load()
    nws = lib.acr_conv1x1_ws_floats(math, N, co, ci, hw)
    ws = torch.empty(nws, dtype=torch.float32, device=x.device) if nws else None
    if math == 1 and CONV1X1_WIMG and ci % 32 == 0:
        wi = img if img is not None else (x3_image_t(w2) if w_transposed else x3_image(w2))
        L.check(lib.acr_conv1x1_x3(L.ptr(wi), L.ptr(x), L.ptr(addend), L.ptr(y), N, co, ci, hw, L.ptr(ws), L.stream_ptr()), "acr_conv1x1_x3")
        return
    L.check(lib.acr_conv1x1_f32(math, L.ptr(w2), w_transposed, L.ptr(x), L.ptr(addend), L.ptr(y), N, co, ci, hw, L.ptr(ws), L.stream_ptr()),
            "acr_conv1x1_f32")


class Conv1x1Fn(Function):
    """Stride-1 1x1 convolution in NCHW on the hand-written GEMM kernels (forward, input and weight gradient).

    Returns (y, x_skip) with x_skip aliasing x: the bottleneck's shortcut consumes x_skip, so the gradient arriving over
    the shortcut is added in the epilogue of the input-gradient GEMM (no separate accumulation pass)."""

    @staticmethod
    def forward(ctx, x, weight, wt=None, math=0, imgs=None):
        ctx.wt, ctx.math = wt, math                          # wt: (cin, cout) copy of the weight, or None
        ctx.imgs = imgs                                      # (image of W, image of W^T) under split products, or None
        N, C, H, W = x.shape
        co = weight.shape[0]
        w2 = weight.reshape(co, C)
        y = torch.empty((N, co, H, W), dtype=x.dtype, device=x.device)
        if x.dtype == torch.float32:
            w2 = w2.contiguous()
            _conv1x1_f32_launch(math, w2, 0, x, None, y, N, co, C, H * W, imgs[0] if imgs else None)
        else:
            L.check(L.load().acr_conv1x1_bf16(L.ptr(w2), w2.stride(0), L.ptr(x), None, L.ptr(y), N, co, C, H * W, L.stream_ptr()),
                    "acr_conv1x1_bf16")
        ctx.save_for_backward(x, weight)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, weight = ctx.saved_tensors
        N, C, H, W = x.shape
        co = weight.shape[0]
        lib = L.load()
        if dy is None:
            return dskip, None, None, None, None
        if not dy.is_contiguous():
            dy = dy.contiguous()
        if dskip is not None and (not dskip.is_contiguous() or dskip.dtype != x.dtype):
            dskip = dskip.to(x.dtype).contiguous()
        dx = dw = None
        if x.dtype == torch.float32:                                          # reference precision: fp32 GEMM kernels, W as stored
            if dy.dtype != torch.float32:
                dy = dy.float()
            if ctx.needs_input_grad[0]:
                w2 = weight.reshape(co, C).contiguous()
                dx = torch.empty_like(x)
                _conv1x1_f32_launch(ctx.math, w2, 1, dy, dskip, dx, N, C, co, H * W, ctx.imgs[1] if ctx.imgs else None)
            if ctx.needs_input_grad[1]:
                ws = torch.empty(lib.acr_conv1x1_wgrad_f32_ws_floats(N, co, C, H * W), dtype=torch.float32, device=x.device)
                dw = torch.empty((co, C, 1, 1), dtype=torch.float32, device=x.device)
                L.check(lib.acr_conv1x1_wgrad_f32(ctx.math, L.ptr(dy), L.ptr(x), N, co, C, H * W, L.ptr(ws), L.ptr(dw), L.stream_ptr()),
                        "acr_conv1x1_wgrad_f32")
            return dx, dw, None, None, None
        if ctx.needs_input_grad[0]:
            wt = ctx.wt if ctx.wt is not None else weight.reshape(co, C).t().contiguous()     # (cin, cout): dX = W^T . dY
            dx = torch.empty_like(x)
            L.check(lib.acr_conv1x1_bf16(L.ptr(wt), wt.stride(0), L.ptr(dy), L.ptr(dskip), L.ptr(dx), N, C, co, H * W,
                                         L.stream_ptr()), "acr_conv1x1_bf16")
        if ctx.needs_input_grad[1]:
            ws = torch.empty(lib.acr_conv1x1_wgrad_ws_floats(N, co, C, H * W), dtype=torch.float32, device=x.device)
            dw = torch.empty((co, C, 1, 1), dtype=weight.dtype, device=x.device)
            L.check(lib.acr_conv1x1_wgrad_bf16(L.ptr(dy), L.ptr(x), N, co, C, H * W, L.ptr(ws), L.ptr(dw), L.stream_ptr()),
                    "acr_conv1x1_wgrad_bf16")
        return dx, dw, None, None, None


def conv3x3_fusable(x, weight, stride, math):
    """fp32 NCHW 3x3 stride-1 convolution with split products that acr_conv3x3_{f32,wgrad_f32} cover (the stem's conv2s)."""
    return (math == 1 and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and weight.dtype == torch.float32 and stride == 1
            and tuple(weight.shape[2:]) == (3, 3) and x.is_contiguous() and not torch.is_autocast_enabled()
            and x.shape[1] % 16 == 0 and weight.shape[0] % 16 == 0 and x.shape[3] % 4 == 0 and x.shape[3] >= 16
            and (x.shape[2] * x.shape[3]) % 16 == 0)


CONV3X3_WIMG = True      # A/B: 3x3 convolutions with the packed weight as a split-product image


def _conv3x3_launch(wp, x, y, N, co, ci, H, W, img=None):
    """y = conv3x3(x) with the packed weight wp (co, 9 ci) -- a callable that packs it on demand --: through its split-product image
    (conv3x3_wimg_kernel: only the activation tile is split in registers; ``img`` when the caller has it already) unless
    switched off."""
    lib = L.load()
    nws = lib.acr_conv3x3_ws_floats(N, co, ci, H, W)
    ws = torch.empty(nws, dtype=torch.float32, device=x.device) if nws else None
    if CONV3X3_WIMG:
        wi = img if img is not None else x3_image(wp() if callable(wp) else wp)
        L.check(lib.acr_conv3x3_x3(L.ptr(wi), L.ptr(x), L.ptr(y), N, co, ci, H, W, L.ptr(ws), L.stream_ptr()), "acr_conv3x3_x3")
        return
    if callable(wp):
        wp = wp()
    L.check(lib.acr_conv3x3_f32(1, L.ptr(wp), L.ptr(x), L.ptr(y), N, co, ci, H, W, L.ptr(ws), L.stream_ptr()), "acr_conv3x3_f32")


class Conv3x3Fn(Function):
    """3x3 stride-1 SAME convolution in NCHW fp32 with split products as implicit GEMMs (csrc/conv3x3.hip): forward, input
    gradient (the same kernel on dy with the taps flipped and the channel roles swapped) and weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, imgs=None):
        N, C, H, W = x.shape
        co = weight.shape[0]
        y = torch.empty((N, co, H, W), dtype=torch.float32, device=x.device)
        _conv3x3_launch(lambda: weight.permute(0, 2, 3, 1).reshape(co, 9 * C).contiguous(), x, y, N, co, C, H, W, imgs[0] if imgs else None)
        ctx.save_for_backward(x, weight)
        ctx.imgs = imgs                                      # (image of the packed weight, image of its input-gradient pack) or None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        N, C, H, W = x.shape
        co = weight.shape[0]
        lib = L.load()
        dy = (dy if dy.dtype == torch.float32 else dy.float()).contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _conv3x3_launch(lambda: weight.flip(2, 3).permute(1, 2, 3, 0).reshape(C, 9 * co).contiguous(), dy, dx, N, C, co, H, W,
                            ctx.imgs[1] if ctx.imgs else None)
        if ctx.needs_input_grad[1]:
            ws = torch.empty(lib.acr_conv3x3_wgrad_ws_floats(N, co, C, H, W), dtype=torch.float32, device=x.device)
            dwp = torch.empty((co, 3, 3, C), dtype=torch.float32, device=x.device)
            L.check(lib.acr_conv3x3_wgrad_f32(1, L.ptr(dy), L.ptr(x), N, co, C, H, W, L.ptr(ws), L.ptr(dwp), L.stream_ptr()),
                    "acr_conv3x3_wgrad_f32")
            dw = dwp.permute(0, 3, 1, 2)
        return dx, dw, None


def conv3x3(x, weight, imgs=None):
    return Conv3x3Fn.apply(x, weight, imgs)


# ---- stride-2 SAME convolutions (7x7 stem convolution, the two stride-2 3x3s) on the tap-table kernels -------------------------
CONV_S2_HIP = True      # A/B: strided convolutions under f32_split on csrc/conv3x3.hip vs MIOpen


class _S2Plan:
    """Everything a k x k stride-2 TF-SAME convolution of a C-channel H x W input needs on the space-to-depth grid
    (include/acr_hip.h, acr_conv_taps_x3): input row u = 2*o + ky - pb (pb = the SAME padding in front) is row o + dy of pixel
    phase py with py = (ky - pb) mod 2, dy = (ky - pb - py) / 2.

    C % 16 == 0 (the 3x3s): tap t = (ky, kx) reads the C rows of ONE phase; the packed weight is the stride-1 one
    (w[co][t*C + c]).  The input gradient is one launch per pixel phase of dx over the taps that feed it, shifts negated.
    4C <= 16 (the stem's 7x7 on 3 channels): a tap is a (dy, dx) pair over all 4C phase-channel rows (+ zero rows up to 16);
    the packed weight is a gather of the 147 columns (``gather``; column 147 = zero), its gradient the inverse gather."""

    def __init__(self, k, C, H, W, device):
        import ctypes
        arr = lambda v: (ctypes.c_int32 * len(v))(*v)

        def split(kk, n):
            pb = max((-(-n // 2) - 1) * 2 + k - n, 0) // 2
            py = (kk - pb) % 2
            return py, (kk - pb - py) // 2

        self.k, self.C = k, C
        self.H2, self.W2 = (H + 1) // 2, (W + 1) // 2
        ys, xs = [split(ky, H) for ky in range(k)], [split(kx, W) for kx in range(k)]
        if C % 16 == 0:
            self.cin, self.xrows, self.gather = C, 4 * C, None
            tdy = [ys[t // k][1] for t in range(k * k)]
            tdx = [xs[t % k][1] for t in range(k * k)]
            phase = [ys[t // k][0] * 2 + xs[t % k][0] for t in range(k * k)]
            self.ntap = k * k
            self.fwd = (arr(tdy), arr(tdx), arr([p * C for p in phase]))
            self.order, self.phases = [], []                 # input gradient: (phase, first tap slot, ntap, tables)
            for p in range(4):
                taps = [t for t in range(k * k) if phase[t] == p]
                self.phases.append((p, len(self.order), len(taps), (arr([-tdy[t] for t in taps]), arr([-tdx[t] for t in taps]), arr([0] * len(taps)))))
                self.order += taps
            self.order_t = torch.tensor(self.order, device=device)
        else:
            assert 4 * C <= 16
            self.cin, self.xrows, self.phases = 16, 16, None
            dys, dxs = sorted(set(d for _, d in ys)), sorted(set(d for _, d in xs))
            self.ntap = len(dys) * len(dxs)
            assert self.ntap <= 16
            kyof = {(py, d): ky for ky, (py, d) in enumerate(ys)}
            kxof = {(px, d): kx for kx, (px, d) in enumerate(xs)}
            gather = []
            for d in dys:
                for e in dxs:
                    for row in range(16):
                        py, px, c = row // (2 * C), (row // C) % 2, row % C
                        ky, kx = kyof.get((py, d)), kxof.get((px, e))
                        gather.append(c * k * k + ky * k + kx if row < 4 * C and ky is not None and kx is not None else C * k * k)
            self.fwd = (arr([d for d in dys for _ in dxs]), arr([e for _ in dys for e in dxs]), arr([0] * self.ntap))
            self.gather = torch.tensor(gather, device=device)
            inv = [0] * (C * k * k)
            for j, g in enumerate(gather):
                if g < C * k * k:
                    inv[g] = j
            self.scatter = torch.tensor(inv, device=device)

    def pack(self, w):
        """(cout, ntap*cin) operand of the forward product."""
        co = w.shape[0]
        if self.gather is None:
            return w.permute(0, 2, 3, 1).reshape(co, self.k * self.k * self.C).contiguous()
        return torch.cat([w.reshape(co, -1), w.new_zeros(co, 1)], 1).index_select(1, self.gather)

    def pack_dgrad(self, w):
        """(cin, 9, cout) with the taps grouped by the pixel phase they feed: phase p's operand is rows x [first*cout, (first+n)*cout)."""
        co, ci = w.shape[0], w.shape[1]
        return w.reshape(co, ci, self.k * self.k).index_select(2, self.order_t).permute(1, 2, 0).contiguous()

    def dgrad_specs(self, wd):
        ci, _, co = wd.shape
        return tuple((wd, first * co, ci, n * co, self.k * self.k * co, n * co, 0, 1) for _, first, n, _ in self.phases)

    def unpack_grad(self, dwp, shape):
        co, ci = shape[0], shape[1]
        if self.gather is None:
            return dwp.view(co, self.k, self.k, ci).permute(0, 3, 1, 2)
        return dwp.index_select(1, self.scatter).view(shape)


_S2_PLANS = {}


def conv_s2_plan(k, C, H, W, device):
    key = (k, C, H % 2, W % 2, H, W, str(device))
    p = _S2_PLANS.get(key)
    if p is None:
        p = _S2_PLANS[key] = _S2Plan(k, C, H, W, device)
    return p


def conv_s2_fusable(x, weight, stride, math):
    """fp32 NCHW k x k stride-2 SAME convolution with split products that the tap-table kernels cover (k = 3 on C % 16 == 0 channels,
    k = 7 on <= 4): forward always; with gradients only on the shapes the weight-gradient kernel and the depth-to-space pass take."""
    if not (CONV_S2_HIP and math == 1 and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and weight.dtype == torch.float32 and stride == 2
            and x.is_contiguous() and not torch.is_autocast_enabled() and weight.shape[2] == weight.shape[3]):
        return False
    k, C, (H, W) = weight.shape[2], x.shape[1], x.shape[2:]
    if not ((k == 3 and C % 16 == 0 and weight.shape[0] % 16 == 0) or (k == 7 and 4 * C <= 16 and weight.shape[0] % 16 == 0)):
        return False
    H2, W2 = (H + 1) // 2, (W + 1) // 2
    if (H2 * W2) % 4 != 0:
        return False
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        return H % 2 == 0 and W % 8 == 0 and W2 >= 16 and (H2 * W2) % 16 == 0
    return True


def _conv_taps(wi, x, y, y_ptr, N, co, ci, H, W, ntap, tab, xrows, yrows):
    lib = L.load()
    nws = lib.acr_conv_taps_ws_floats(N, co, ci, H, W, ntap) if yrows == co else 0
    ws = torch.empty(nws, dtype=torch.float32, device=x.device) if nws else None
    L.check(lib.acr_conv_taps_x3(L.ptr(wi), L.ptr(x), y_ptr, N, co, ci, H, W, ntap, tab[0], tab[1], tab[2], xrows, yrows, L.ptr(ws), L.stream_ptr()),
            "acr_conv_taps_x3")


class ConvS2Fn(Function):
    """k x k stride-2 SAME convolution in NCHW fp32 with split products (csrc/conv3x3.hip, tap-table kernels): a space-to-depth
    copy of the input, then forward / input gradient (one launch per pixel phase + depth-to-space) / weight gradient as implicit
    GEMMs -- what the reference runs as F.pad + F.conv2d (models/layers/std_conv.py:56-65).  ``imgs``: (forward image, the four
    phase images of the input gradient) of the weight for EVEN H, W, or None (built here)."""

    @staticmethod
    def forward(ctx, x, weight, imgs=None):
        N, C, H, W = x.shape
        co, k = weight.shape[0], weight.shape[2]
        lib = L.load()
        plan = conv_s2_plan(k, C, H, W, x.device)
        if imgs is not None and (H % 2 or W % 2):
            imgs = None                                      # the cached images are those of the even-size tap layout
        xs = torch.empty((N, plan.xrows, plan.H2, plan.W2), dtype=torch.float32, device=x.device)
        L.check(lib.acr_space_to_depth2_f32(L.ptr(x), L.ptr(xs), N, C, H, W, plan.xrows, L.stream_ptr()), "acr_space_to_depth2_f32")
        wi = imgs[0] if imgs else x3_image(plan.pack(weight.detach()))
        y = torch.empty((N, co, plan.H2, plan.W2), dtype=torch.float32, device=x.device)
        _conv_taps(wi, xs, y, L.ptr(y), N, co, plan.cin, plan.H2, plan.W2, plan.ntap, plan.fwd, plan.xrows, co)
        ctx.save_for_backward(xs, weight)
        ctx.plan, ctx.imgs, ctx.xshape = plan, imgs, (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, weight = ctx.saved_tensors
        plan, (N, C, H, W) = ctx.plan, ctx.xshape
        co = weight.shape[0]
        lib = L.load()
        dy = (dy if dy.dtype == torch.float32 else dy.float()).contiguous()
        H2, W2 = plan.H2, plan.W2
        dx = dw = None
        if ctx.needs_input_grad[0]:
            pim = ctx.imgs[1:] if ctx.imgs else x3_image_many(plan.dgrad_specs(plan.pack_dgrad(weight.detach())), dy.device)
            dxs = torch.empty((N, 4 * C, H2, W2), dtype=torch.float32, device=dy.device)
            for (p, _, n, tab), im in zip(plan.phases, pim):
                _conv_taps(im, dy, dxs, L.c_void_p(dxs.data_ptr() + 4 * p * C * H2 * W2), N, C, co, H2, W2, n, tab, co, 4 * C)
            dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
            L.check(lib.acr_depth_to_space2_f32(L.ptr(dxs), L.ptr(dx), N, C, H, W, L.stream_ptr()), "acr_depth_to_space2_f32")
        if ctx.needs_input_grad[1]:
            ws = torch.empty(lib.acr_conv_taps_wgrad_ws_floats(N, co, plan.cin, H2, W2, plan.ntap), dtype=torch.float32, device=dy.device)
            dwp = torch.empty((co, plan.ntap * plan.cin), dtype=torch.float32, device=dy.device)
            L.check(lib.acr_conv_taps_wgrad_f32(1, L.ptr(dy), L.ptr(xs), N, co, plan.cin, H2, W2, plan.ntap, plan.fwd[0], plan.fwd[1], plan.fwd[2],
                                                plan.xrows, L.ptr(ws), L.ptr(dwp), L.stream_ptr()), "acr_conv_taps_wgrad_f32")
            dw = plan.unpack_grad(dwp, weight.shape)
        return dx, dw, None


def conv_s2(x, weight, imgs=None):
    return ConvS2Fn.apply(x, weight, imgs)


def conv1x1(x, weight, wt=None, math=0, imgs=None):
    return Conv1x1Fn.apply(x, weight, wt, math, imgs)[0]


def conv1x1_skip(x, weight, wt=None, math=0, imgs=None):
    """(conv(x), x_skip): see Conv1x1Fn.  ``wt``: the (cin, cout) copy of the weight when the caller already has one; ``imgs``:
    its split-product images (W, W^T) likewise."""
    if not SKIP_FUSION:
        return Conv1x1Fn.apply(x, weight, wt, math, imgs)[0], x
    return Conv1x1Fn.apply(x, weight, wt, math, imgs)


class LayerNormFn(Function):
    """LayerNorm over the last dim of a bf16 (.., C) tensor on acr_layernorm_{fwd,bwd}_bf16.

    Returns (LN(x), x_skip): x_skip aliases x and is what the residual connection should consume, so that the gradient
    arriving over the skip path reaches this node's backward and is added inside the LayerNorm backward kernel instead of
    by a separate autograd accumulation pass (blocks: x + f(LN(x)), models/vision_transformer.py:224-226)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, image=False):
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        M = x2.shape[0]
        lib = L.load()
        stats = torch.empty(2 * M, dtype=torch.float32, device=x.device)
        ctx.save_for_backward(x2, weight, stats)
        ctx.set_materialize_grads(False)
        if image:
            # LN(x) leaves as the split-product image its one consumer (a Linear) multiplies by and keeps for its weight
            # gradient; the differentiable output is a PLACEHOLDER of the right shape (one element, expanded: no memory, never
            # read) that only carries the Linear's input gradient back here
            img = x3_image_empty(M, C, x.device)
            L.check(lib.acr_layernorm_image_f32(L.ptr(x2), L.ptr(weight), L.ptr(bias), L.ptr(img), L.ptr(stats), M, C, eps, L.stream_ptr()),
                    "acr_layernorm_image_f32")
            ctx.mark_non_differentiable(img)
            return x.new_empty(1).expand(x.shape), x.view_as(x), img
        y = torch.empty_like(x2)
        fwd = lib.acr_layernorm_fwd_f32 if x.dtype == torch.float32 else lib.acr_layernorm_fwd_bf16
        L.check(fwd(L.ptr(x2), L.ptr(weight), L.ptr(bias), L.ptr(y), L.ptr(stats), M, C, eps, L.stream_ptr()), "acr_layernorm_fwd")
        return y.reshape(x.shape), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip, dimg=None):
        x2, weight, stats = ctx.saved_tensors
        M, C = x2.shape
        if dy is None:
            return dskip, None, None, None, None
        lib = L.load()
        dy2 = dy.reshape(M, C)
        if not dy2.is_contiguous() or dy2.dtype != x2.dtype:
            dy2 = dy2.to(x2.dtype).contiguous()
        ds2 = None
        if dskip is not None:
            ds2 = dskip.reshape(M, C)
            if not ds2.is_contiguous() or ds2.dtype != x2.dtype:
                ds2 = ds2.to(x2.dtype).contiguous()
        dx = torch.empty_like(x2)
        ws = torch.empty(lib.acr_layernorm_ws_floats(M, C), dtype=torch.float32, device=x2.device)
        dg = torch.empty(C, dtype=x2.dtype, device=x2.device)
        db = torch.empty(C, dtype=x2.dtype, device=x2.device)
        bwd = lib.acr_layernorm_bwd_f32 if x2.dtype == torch.float32 else lib.acr_layernorm_bwd_bf16
        L.check(bwd(L.ptr(dy2), L.ptr(x2), L.ptr(weight), L.ptr(stats), L.ptr(ds2), L.ptr(dx), L.ptr(ws), L.ptr(dg), L.ptr(db), M, C,
                    L.stream_ptr()), "acr_layernorm_bwd")
        return dx.reshape(dy.shape), dg, db, None, None


def layer_norm_fusable(x, ln):
    C = x.shape[-1]
    return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and ln.weight.dtype == x.dtype and x.is_contiguous()
            and C % 256 == 0 and C <= 1024 and not (x.dtype == torch.float32 and torch.is_autocast_enabled()))


def layer_norm(x, ln, use_hip=True):
    """nn.LayerNorm forward; contiguous bf16 CUDA rows with C % 256 == 0 (<= 1024) take the HIP kernels."""
    if use_hip and layer_norm_fusable(x, ln):
        return LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps, False)[0]
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), ln.weight, ln.bias, ln.eps)


SKIP_FUSION = True      # A/B switch for the fused skip-gradient adds


def layer_norm_skip(x, ln, use_hip=True):
    """(LN(x), x_skip) -- see LayerNormFn; on the stock path x_skip is x itself."""
    if use_hip and layer_norm_fusable(x, ln):
        if not SKIP_FUSION:
            return LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps, False)[0], x
        return LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps, False)
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), ln.weight, ln.bias, ln.eps), x


TOKENS_HIP = True      # A/B: the hybrid ViT's token assembly as one kernel each way


class TokensFn(Function):
    """tokens = cat(prefix, (y + bias)^T) + pos in one pass each way (csrc/tokens.hip; vision_transformer.py:449-467): y (B, D, h, w) is
    the patch projection's output WITHOUT its bias, prefix (P, D) the class (+ distillation) token, pos (1, P + h*w, D)."""

    @staticmethod
    def forward(ctx, y, bias, prefix, pos):
        B, D = y.shape[:2]
        T = y.shape[2] * y.shape[3]
        P = prefix.shape[0]
        tok = torch.empty((B, P + T, D), dtype=torch.float32, device=y.device)
        L.check(L.load().acr_tokens_fwd_f32(L.ptr(y), L.ptr(bias), L.ptr(prefix), L.ptr(pos), L.ptr(tok), B, D, T, P, L.stream_ptr()), "acr_tokens_fwd_f32")
        ctx.geom = (tuple(y.shape), P)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        (B, D, h, w), P = ctx.geom
        T = h * w
        dtok = (dtok if dtok.dtype == torch.float32 else dtok.float()).contiguous()
        dy = torch.empty((B, D, h, w), dtype=torch.float32, device=dtok.device)
        dpos = torch.empty((1, P + T, D), dtype=torch.float32, device=dtok.device)
        L.check(L.load().acr_tokens_bwd_f32(L.ptr(dtok), L.ptr(dy), L.ptr(dpos), B, D, T, P, L.stream_ptr()), "acr_tokens_bwd_f32")
        dbias = dpos[0, P:].sum(0) if ctx.needs_input_grad[1] else None
        return dy, dbias, dpos[0, :P], dpos


def tokens_fusable(y, bias, prefix, pos):
    return (TOKENS_HIP and y.is_cuda and y.dim() == 4 and all(t.dtype == torch.float32 and t.is_contiguous() for t in (y, bias, prefix, pos))
            and prefix.shape[0] <= 8 and pos.shape[1] == prefix.shape[0] + y.shape[2] * y.shape[3] and not torch.is_autocast_enabled())


def tokens(y, bias, prefix, pos):
    return TokensFn.apply(y, bias, prefix, pos)


LN_IMAGE = True      # A/B: the blocks' LayerNorms write their consumer's operand image directly


def ln_image_usable(x, ln, lin, math, use_hip=True):
    """norm -> Linear pairs of a block under split products where LN(x) can leave as the Linear's operand image (LayerNormFn with
    image=True + LinearF32Fn / MlpF32Fn with x_image): exactly the conditions under which that Linear takes its image path."""
    C = x.shape[-1]
    return (LN_IMAGE and use_hip and math == 1 and X3_IMAGES and F32_HIP_LINEAR and SKIP_FUSION and x.is_cuda and x.dtype == torch.float32
            and x.is_contiguous() and ln.weight.dtype == torch.float32 and ln.bias is not None and C % 256 == 0 and C <= 1024
            and not torch.is_autocast_enabled() and lin.weight.dtype == torch.float32 and lin.weight.is_contiguous() and lin.weight.shape[1] == C
            and lin.weight.shape[0] % 4 == 0 and lin.weight.shape[0] >= 32 and _f32_ok(x, ln.weight, ln.bias, lin.weight, lin.bias))


def layer_norm_image(x, ln):
    """(placeholder for LN(x), x_skip, image of LN(x)) -- see LayerNormFn."""
    return LayerNormFn.apply(x, ln.weight, ln.bias, ln.eps, True)


GN_ACT = {"none": 0, "relu": 1, "add_relu": 2}
F32_HIP_NORMS = True      # A/B switch: fp32 GroupNorm on the HIP kernels vs torch


def groupnorm_fusable(x, resid=None):
    """Shapes/dtypes the fused bf16 GroupNorm kernel handles (everything else stays on torch ops)."""
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.dim() == 4 and x.is_contiguous()):
        return False
    N, C, H, W = x.shape
    if x.dtype == torch.float32:                            # streaming fp32 kernels: any group size, HW % 4 == 0
        if C % 32 or (H * W) % 4 or torch.is_autocast_enabled() or not F32_HIP_NORMS:
            return False
    elif C % 32 or (H * W) % 8 or (C // 32) * H * W // 8 > 1024 * 13:
        return False
    return resid is None or (resid.shape == x.shape and resid.dtype == x.dtype and resid.is_contiguous())


GN_RELU_MASK = True      # A/B: fp32 relu(gn(x) + resid) leaves a one-byte-per-vector ReLU mask for its backward (no residual read there)


class GroupNormActFn(Function):
    """y = act(GroupNorm32(x) [+ resid]) on acr_groupnorm_{fwd,bwd}_bf16 (bf16 NCHW)."""

    @staticmethod
    def forward(ctx, x, weight, bias, resid, act, eps):
        N, C, H, W = x.shape
        lib = L.load()
        y = torch.empty_like(x)
        stats = torch.empty(N * 32 * 2, dtype=torch.float32, device=x.device)
        mask = None
        if x.dtype == torch.float32 and act == 2 and GN_RELU_MASK and any(ctx.needs_input_grad):
            # relu(gn(x) + resid) with a backward to come: the forward leaves its ReLU mask, one byte per 16-byte vector, and the
            # backward reads that instead of the residual (1/16 of its bytes: all it needed the residual for)
            mask = torch.empty(x.numel() // 4, dtype=torch.uint8, device=x.device)
            L.check(lib.acr_groupnorm_fwd_mask_f32(L.ptr(x), L.ptr(resid), L.ptr(weight), L.ptr(bias), L.ptr(y), L.ptr(stats), N, C, H * W, eps,
                                                   L.ptr(mask), L.stream_ptr()), "acr_groupnorm_fwd_mask_f32")
        elif x.dtype == torch.float32:
            # gradient-free passes (CAM generation) of a few samples: the (sample, group) pairs are cut into parts (two launches)
            nws = lib.acr_groupnorm_fwd_ws_floats(N, C, H * W) if not any(ctx.needs_input_grad) else 0
            ws = torch.empty(nws, dtype=torch.float32, device=x.device) if nws else None
            L.check(lib.acr_groupnorm_fwd_f32(L.ptr(x), L.ptr(resid), L.ptr(weight), L.ptr(bias), L.ptr(y), L.ptr(stats), N, C, H * W, eps, act,
                                              L.ptr(ws), L.stream_ptr()), "acr_groupnorm_fwd_f32")
        else:
            L.check(lib.acr_groupnorm_fwd_bf16(L.ptr(x), L.ptr(resid), L.ptr(weight), L.ptr(bias), L.ptr(y), L.ptr(stats), N, C, H * W, eps, act,
                                               L.stream_ptr()), "acr_groupnorm_fwd_bf16")
        ctx.save_for_backward(x, weight, bias, stats, mask if mask is not None else (resid if act == 2 else None))
        ctx.act = act
        ctx.by_mask = mask is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, stats, resid = ctx.saved_tensors
        N, C, H, W = x.shape
        lib = L.load()
        if not dy.is_contiguous() or dy.dtype != x.dtype:
            dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.act == 2 else None
        part = torch.empty((2, N, C), dtype=torch.float32, device=x.device)
        dgb = torch.empty((2, C), dtype=x.dtype, device=x.device)             # summed over samples inside the call
        if ctx.by_mask:
            L.check(lib.acr_groupnorm_bwd_mask_f32(L.ptr(dy), L.ptr(x), L.ptr(resid), L.ptr(weight), L.ptr(bias), L.ptr(stats), L.ptr(dx), L.ptr(dres),
                                                   L.ptr(part[0]), L.ptr(part[1]), L.ptr(dgb[0]), L.ptr(dgb[1]), N, C, H * W, L.stream_ptr()),
                    "acr_groupnorm_bwd_mask_f32")
            return dx, dgb[0], dgb[1], dres, None, None
        bwd = lib.acr_groupnorm_bwd_f32 if x.dtype == torch.float32 else lib.acr_groupnorm_bwd_bf16
        L.check(bwd(L.ptr(dy), L.ptr(x), L.ptr(resid), L.ptr(weight), L.ptr(bias), L.ptr(stats), L.ptr(dx), L.ptr(dres), L.ptr(part[0]),
                    L.ptr(part[1]), L.ptr(dgb[0]), L.ptr(dgb[1]), N, C, H * W, ctx.act, L.stream_ptr()), "acr_groupnorm_bwd")
        return dx, dgb[0], dgb[1], dres, None, None


def groupnorm_act(x, weight, bias, act="relu", resid=None, eps=1e-5):
    return GroupNormActFn.apply(x, weight.to(x.dtype), bias.to(x.dtype), resid, GN_ACT[act], eps)


def _wstd_desc(p0s, p1s, p2s, device, p3s=None):
    import numpy as np
    rec = np.zeros(len(p0s), dtype=[("p0", "<u8"), ("p1", "<u8"), ("p2", "<u8"), ("p3", "<u8"), ("cout", "<i4"),
                                    ("n", "<i4"), ("ch_start", "<i4"), ("pad", "<i4")])
    ch = 0
    for i, w in enumerate(p0s):
        cout = w.shape[0]
        rec[i] = (w.data_ptr(), p1s[i].data_ptr(), p2s[i].data_ptr() if p2s is not None else 0,
                  p3s[i].data_ptr() if (p3s is not None and p3s[i] is not None) else 0, cout, w.numel() // cout, ch, 0)
        ch += cout
    # pinned staging + asynchronous copy: a pageable .to(device) drains the whole stream on the host (twice per step here --
    # the forward's first and the backward's last launch -- which kept Python from running ahead of the GPU)
    raw = rec.view(np.uint8)
    host = torch.empty(raw.size, dtype=torch.uint8, pin_memory=True)
    host.numpy()[:] = raw
    return host.to(device, non_blocking=True), ch


WSTD_TRANSPOSED = True      # A/B: transposed 1x1 weights from the weight-std launch


class WeightStdAllFn(Function):
    """w_hat_i = (w_i - mean) / (std + eps) for ALL conv weights of the stem at once (one launch forward, one
    backward) on acr_weight_std_bf16.  Inputs and outputs are tuples of (cout, cin, k, k) bf16 tensors."""

    @staticmethod
    def forward(ctx, eps, *weights):
        ws = [w.contiguous() for w in weights]
        outs = [torch.empty_like(w) for w in ws]
        # bf16 1x1 convolutions: the standardised weight is also written transposed, (cin, cout), for Conv1x1Fn's input
        # gradient (the fp32 kernels read W as stored and need no copy)
        outs_t = [torch.empty((w.shape[1], w.shape[0]), dtype=w.dtype, device=w.device)
                  if (WSTD_TRANSPOSED and w.dtype == torch.bfloat16 and w.shape[2] == 1 and w.shape[3] == 1) else None for w in ws]
        desc, total = _wstd_desc(ws, outs, None, ws[0].device, outs_t)
        ctx.transposed = outs_t
        fn = L.load().acr_weight_std_f32 if ws[0].dtype == torch.float32 else L.load().acr_weight_std_bf16
        L.check(fn(L.ptr(desc), len(ws), total, eps, 0, L.stream_ptr()), "acr_weight_std")
        ctx.save_for_backward(*ws)
        ctx.eps = eps
        ctx._keep = desc
        WeightStdAllFn.last_transposed = outs_t              # picked up by the caller right after apply()
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        ws = ctx.saved_tensors
        gs = [g.contiguous() if g is not None else torch.zeros_like(w) for g, w in zip(grads, ws)]
        dws = [torch.empty_like(w) for w in ws]
        desc, total = _wstd_desc(list(ws), gs, dws, ws[0].device)
        fn = L.load().acr_weight_std_f32 if ws[0].dtype == torch.float32 else L.load().acr_weight_std_bf16
        L.check(fn(L.ptr(desc), len(ws), total, ctx.eps, 1, L.stream_ptr()), "acr_weight_std")
        ctx._keep_b = desc
        return (None,) + tuple(dws)


def weight_std_all(weights, eps=1e-5):
    return WeightStdAllFn.apply(eps, *weights)


class ConsistencyFn(Function):
    """(cls_align, aff_align) of train_acr.py:143-161 on one (2B,L,T,T) stack holding view 1 in [:B] and
    view 2 in [B:] (both views run as one 2B batch; GroupNorm/LayerNorm are per-sample so this is exact)."""

    @staticmethod
    def forward(ctx, a, p):
        L.require_gpu(a)
        assert a.dtype == torch.float32 and a.dim() == 4 and a.shape[0] % 2 == 0
        B2, Ly, T, T2 = a.shape
        B = B2 // 2
        assert T == T2 == p * p + 1, "attention map side %d != p*p+1 (p=%d)" % (T, p)
        assert a.stride(3) == 1 and a.stride(2) == T and a.stride(1) == T * T, "maps must be dense per sample"
        lib = L.load()
        ws = torch.empty(lib.acr_consistency_ws_floats(B, Ly, T), dtype=torch.float32, device=a.device)
        out = torch.empty(2, dtype=torch.float32, device=a.device)
        L.check(lib.acr_consistency_fwd(L.ptr(a[:B]), L.ptr(a[B:]), a.stride(0), B, Ly, T, p, L.ptr(ws), L.ptr(out),
                                        L.stream_ptr()), "acr_consistency_fwd")
        ctx.save_for_backward(a)
        ctx.p = p
        return out

    @staticmethod
    def backward(ctx, gout):
        (a,) = ctx.saved_tensors
        B2, Ly, T, _ = a.shape
        B = B2 // 2
        lib = L.load()
        gout = gout.contiguous().float()
        # gradient stack with a 16-byte-aligned row pitch; what autograd sees is the [..., :T] view of it
        g = torch.empty((B2, Ly, T, pad4(T)), dtype=torch.float32, device=a.device)
        L.check(lib.acr_consistency_bwd(L.ptr(a[:B]), L.ptr(a[B:]), a.stride(0), B, Ly, T, ctx.p, L.ptr(gout),
                                        L.ptr(g[:B]), L.ptr(g[B:]), g.stride(0), g.stride(2), L.stream_ptr()),
                "acr_consistency_bwd")
        return g[..., :T], None


def consistency(a, p, a2=None):
    """Returns (cls_align, aff_align).  ``a`` is the (2B,L,T,T) two-view stack; if ``a2`` is given, ``a`` and
    ``a2`` are separate (B,L,T,T) stacks (reference-style call) and are concatenated first."""
    if a2 is not None:
        a = torch.cat([a, a2], dim=0)
    out = ConsistencyFn.apply(a, p)
    return out[0], out[1]


class MlsmFn(Function):
    """F.multilabel_soft_margin_loss(x, y) (mean reduction) of fp32 logits as one launch each way (csrc/mlsm.hip); the upstream
    gradient stays on the device."""

    @staticmethod
    def forward(ctx, x, y):
        N, C = x.shape
        out = torch.empty((), dtype=torch.float32, device=x.device)
        L.check(L.load().acr_mlsm_fwd_f32(L.ptr(x), x.stride(0), L.ptr(y), y.stride(0), N, C, L.ptr(out), L.stream_ptr()), "acr_mlsm_fwd_f32")
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        N, C = x.shape
        g = g.to(torch.float32).contiguous()
        dx = torch.empty((N, C), dtype=torch.float32, device=x.device)
        L.check(L.load().acr_mlsm_bwd_f32(L.ptr(x), x.stride(0), L.ptr(y), y.stride(0), L.ptr(g), N, C, L.ptr(dx), L.stream_ptr()),
                "acr_mlsm_bwd_f32")
        return dx, None


def mlsm_loss(x, y):
    """multilabel_soft_margin_loss(x, y): the HIP kernel pair for fp32 (N, C) logits on the GPU, the stock op otherwise."""
    if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and y.shape == x.shape and y.is_cuda and x.stride(1) == 1
            and not torch.is_autocast_enabled()):
        y = y.to(torch.float32)
        if y.stride(1) != 1:
            y = y.contiguous()
        return MlsmFn.apply(x, y)
    return torch.nn.functional.multilabel_soft_margin_loss(x, y)


def patch_cam(x, weight, bias):
    """relu(x @ weight.T + bias) for patch tokens x (N, D) -> (N, C) fp32 (DPT/ACR.py:133-134)."""
    L.require_gpu(x, weight, bias)
    N, D = x.shape
    C = weight.shape[0]
    assert x.stride(1) == 1 and weight.is_contiguous() and bias.is_contiguous()
    weight, bias = weight.to(x.dtype), bias.to(x.dtype)
    out = torch.empty((N, C), dtype=torch.float32, device=x.device)
    L.check(L.load().acr_patch_cam(L.ptr(x), x.stride(0), L.ptr(weight), L.ptr(bias), N, D, C, L.dtype_code(x.dtype),
                                   L.ptr(out), L.stream_ptr()), "acr_patch_cam")
    return out


def bilinear_resize(src, out_hw, align_corners, chan_mul=None, hflip=False, out=None, channels_last=False):
    """Resize (C,ih,iw) [or (ih,iw,C) when channels_last] fp32 to (C,oh,ow), optional per-channel multiply and
    horizontal flip; accumulates into ``out`` when given (infer_cam.py:157-160,187,195-196,201,208)."""
    L.require_gpu(src)
    assert src.dtype == torch.float32 and src.is_contiguous()
    if channels_last:
        ih, iw, C = src.shape
        sc, sp = 1, C
    else:
        C, ih, iw = src.shape
        sc, sp = ih * iw, 1
    oh, ow = out_hw
    acc = out is not None
    if out is None:
        out = torch.empty((C, oh, ow), dtype=torch.float32, device=src.device)
    assert out.shape == (C, oh, ow) and out.is_contiguous()
    if chan_mul is not None:
        chan_mul = chan_mul.to(device=src.device, dtype=torch.float32).contiguous()
    L.check(L.load().acr_bilinear_resize(L.ptr(src), sc, sp, C, ih, iw, L.ptr(out), oh, ow, int(align_corners),
                                         L.ptr(chan_mul), int(hflip), int(acc), L.stream_ptr()), "acr_bilinear_resize")
    return out


def aff_refine_batch(stack, cams):
    """patch_aff @ cam for every sample in one launch: stack (S,L,T,T) fp32 head-mean maps (dense per sample), cams (S,n,T-1)
    -> (S,n,T-1)."""
    L.require_gpu(stack, cams)
    S, Ly, T, _ = stack.shape
    assert stack.stride(3) == 1 and stack.stride(2) == T and stack.stride(1) == T * T and cams.is_contiguous()
    assert cams.shape[0] == S and cams.shape[2] == T - 1
    out = torch.empty_like(cams)
    L.check(L.load().acr_aff_refine_batch(L.ptr(stack), stack.stride(0), Ly, T, L.ptr(cams), cams.shape[1], S, L.ptr(out),
                                          L.stream_ptr()), "acr_aff_refine_batch")
    return out


def aff_refine(stack_b, cams):
    """patch_aff @ cam for one sample: stack_b (L,T,T) fp32 head-mean maps, cams (n, T-1) -> (n, T-1)."""
    L.require_gpu(stack_b, cams)
    Ly, T, _ = stack_b.shape
    assert stack_b.is_contiguous() and cams.is_contiguous() and cams.shape[1] == T - 1
    out = torch.empty_like(cams)
    L.check(L.load().acr_aff_refine(L.ptr(stack_b), Ly, T, L.ptr(cams), cams.shape[0], L.ptr(out), L.stream_ptr()),
            "acr_aff_refine")
    return out
