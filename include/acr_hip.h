/*
 * acr_hip.h -- C ABI of libacr_hip.so, the MI355X (gfx950) implementation of the ACR_WSSS hot path.
 *
 * The reference (OpenNLPLab/ACR_WSSS) has no FFI layer: its hot path is stock PyTorch ops called from
 * Python.  This header is the boundary the replacement defines underneath the reference's Python
 * surface.  Each entry point names the reference code it replaces (file:line under
 * the reference root).  INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - plain C, no torch types; device pointers are raw `void*` / `float*` owned by the caller
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); no hidden synchronisation,
 *     no allocation -> safe to capture into a hipGraph, re-entrant.  The ONLY process-wide state is
 *     the explicit option table below (acr_set_option): kernel-variant selectors for A/B measurements,
 *     atomically readable, never read from the environment by the library itself
 *   - returns 0 on success, a negative acr_status on failure; acr_last_error() gives a
 *     thread-local message for the last failure on the calling thread
 *   - tensors are row-major; "stride" arguments are in ELEMENTS of the tensor's dtype
 */
#ifndef ACR_HIP_H
#define ACR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACR_ABI_VERSION 2      /* 2: per-call `math` argument of the fp32 GEMM / 1x1-convolution entries, ACR_F32_BF16X3, acr_attn_bwd_ws_floats, acr_split3_bf16 */

typedef enum acr_status {
    ACR_OK = 0,
    ACR_ERR_INVALID = -1,      /* bad argument (null pointer, unsupported head_dim, ...) */
    ACR_ERR_LAUNCH = -2,       /* hipLaunch / hipGetLastError failure */
    ACR_ERR_UNSUPPORTED = -3   /* dtype / shape not built into this library */
} acr_status;

/* ACR_F32: fp32 tensors, exact-fp32 MFMA (parity / inference precision).
 * ACR_BF16: bf16 tensors, bf16 MFMA with fp32 accumulate and fp32 softmax (training throughput precision).
 * ACR_BF16_F32MATH: bf16 tensors, every product in exact fp32 (debug / reference for the bf16 kernels).
 * ACR_F32_BF16X3: fp32 tensors; every matrix product is evaluated on the bf16 MFMA as six exact terms of a three-way
 *   operand split (acr_math below), fp32 accumulate, fp32 softmax / lse / delta / head mean -- fp32-accurate results at the
 *   bf16 matrix rate.  Accepted by acr_attn_fwd_scores / acr_attn_bwd_scores (training attention). */
typedef enum acr_dtype { ACR_F32 = 0, ACR_BF16 = 1, ACR_BF16_F32MATH = 2, ACR_F32_BF16X3 = 3 } acr_dtype;

/* How an fp32 entry point multiplies (a PER-CALL argument: two models, or two calls of one model, may differ).
 * ACR_MATH_F32: v_mfma_f32_32x32x2_f32, exact fp32 products (157 TF peak).
 * ACR_MATH_BF16X3: a = a0 + a1 + a2 with a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1) (3 x 8 = 24 mantissa
 *   bits = all of an fp32 mantissa), a.b ~ a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0 on v_mfma_f32_32x32x16_bf16 (the dropped
 *   terms are <= 2^-24 |a.b|); every bf16 x bf16 product is exact in fp32 and the sums accumulate in fp32.  Same tensors,
 *   layouts, epilogues and determinism as ACR_MATH_F32; as accurate against float64 as the fp32 FMA chain
 *   (tests/test_kernels_gpu.py::test_split_math_adversarial_operands).  Peak: bf16 MFMA peak / 6 = 417 TF-equivalent. */
typedef enum acr_math { ACR_MATH_F32 = 0, ACR_MATH_BF16X3 = 1 } acr_math;

typedef enum acr_getam_func {   /* DPT/ACR.py:189-205 */
    ACR_GETAM_GRAD = 0, ACR_GETAM_CAM_GRAD = 1, ACR_GETAM_GRAD_S = 2, ACR_GETAM_CAM_GRAD_S = 3
} acr_getam_func;

/* Geometry of one multi-head attention problem.  q, k, v share one stride triple so that they can
 * alias slices of the packed output of the qkv Linear, (B, T, 3, H, d): sb = 3*H*d*T, st = 3*H*d,
 * sh = d (no permute copy, models/vision_transformer.py:200-201).  o / do use (o_sb, o_st, o_sh);
 * for the reference's (B, T, H*d) activation layout: o_sb = T*H*d, o_st = H*d, o_sh = d
 * (the transpose(1,2).reshape of vision_transformer.py:211 is folded into the store).
 * head_dim must be 64 (every ViT the reference's backbone_dict reaches: tiny/small/base/large). */
typedef struct acr_attn_desc {
    int32_t B, H, T, head_dim;
    int32_t dtype;              /* acr_dtype of q/k/v/o/do/dq/dk/dv */
    float   scale;              /* head_dim^-0.5, vision_transformer.py:173 */
    int64_t qkv_sb, qkv_st, qkv_sh;
    int64_t o_sb, o_st, o_sh;
} acr_attn_desc;

int         acr_version(void);
const char* acr_last_error(void);

/* Kernel-variant selectors (A/B measurement switches; defaults = the measured-fastest settings).
 * Process-wide, set explicitly by the host (acr_wsss_amd/_lib.py maps the documented ACR_* environment variables
 * onto these calls at load time); results are identical under every setting up to the documented tolerances.
 * acr_set_option returns ACR_ERR_INVALID for an unknown option; acr_get_option returns INT32_MIN for one. */
typedef enum acr_option {
    ACR_OPT_GEMM_VARIANT = 0,   /* acr_linear_bf16: 2 = by size (128x128 LDS-DMA / 320x256 8-wave), 3 = 256x256 4-wave, 4 = always 320x256 */
    ACR_OPT_GEMM_NOWIDE = 1,    /* 1: never take the 320x256 8-wave kernel */
    ACR_OPT_GEMM_REGSTAGE = 2,  /* 1: register-staged 128x128 kernel (oldest variant) */
    ACR_OPT_WGRAD_VARIANT = 3,  /* acr_wgrad_bf16: 1 = 128x128 tiles, 2 = 256x256 */
    ACR_OPT_WGRAD_WAVES = 4,    /* 4 or 8 waves per 256x256 workgroup */
    ACR_OPT_DQ_VARIANT = 5,     /* acr_attn_bwd (bf16) dQ sweep: 0 = by presence of G, 2 = 2-wave, 4 = 4-wave */
    ACR_OPT_GEMM_F32_REGSTAGE = 6, /* 1: acr_gemm_f32 always takes the register-staged kernel (A/B of the LDS-DMA kernel) */
    ACR_OPT_ATTN_DELTA_1HEAD = 7, /* 1: the delta pass of the resident-score backward keeps one wave per (query block, head) reading the gradient block from HBM itself (A/B of the 4-head LDS-staged kernel) */
    ACR_OPT_RESERVED_8 = 8,      /* was ACR_OPT_ATTN_F32_NW (five-wave forward workgroups: measured slower, out of the library since round 5) */
    ACR_OPT_GEMM_F32_NOTAIL = 9, /* 1: acr_gemm_f32 NT / NN never K-splits the tiles beyond the last whole half-round (A/B) */
    ACR_OPT_ATTN_F32_NOSPLITTAIL = 10, /* 1: resident-score attention keeps the leftover 32-row block as an ordinary (1 live wave) workgroup (A/B) */
    ACR_OPT_GEMM_X3_INKERNEL = 11, /* 1: split-product acr_gemm_f32 splits operand tiles inside the GEMM kernel instead of once per product into bf16 planes (A/B) */
    ACR_OPT_GN_PLAN = 12,        /* fp32 GroupNorm, register-resident kernels: workgroup size preference -- 0 = 1024 threads for groups of more than 2048 vectors, 1 = 512 threads where the slots allow, 2 = the smallest of 256 / 512 / 1024 threads whose lanes can hold the group, 3 (default) = 0 in the forward and 2 in the backward: measured fastest, profiles/r06_gn_plans.txt (A/B) */
    ACR_OPT_COUNT_
} acr_option;
int     acr_set_option(int32_t option, int32_t value);
int32_t acr_get_option(int32_t option);

/* ---- attention (models/vision_transformer.py:198-214 `Attention.forward`, minus the two Linears) ----
 * O = softmax(q k^T * scale) v without materialising P.  lse2 (B,H,T) fp32 receives the row
 * log-sum-exp in base-2 units of the scaled logits: P[b,h,i,j] = exp2(s2 - lse2), s2 = q.k*scale*log2(e).
 * If pmean != NULL it receives mean_h P (B,T,T) fp32 with batch stride pmean_sb and row pitch pmean_st
 * (>= T): the per-layer slice of the (B,L,T,T) stack DPT/ACR.py:107-112 builds with 12 mean kernels + stack. */
int acr_attn_fwd(const acr_attn_desc* desc, const void* q, const void* k, const void* v,
                 void* o, float* lse2, float* pmean, int64_t pmean_sb, int64_t pmean_st, void* stream);

/* Backward of acr_attn_fwd.  gmean (nullable) is dLoss/d(mean_h P), (B,T,T) fp32, batch stride
 * gmean_sb, row pitch gmean_st >= T (what autograd delivers to the `torch.mean(attn, dim=1)` node of
 * DPT/ACR.py:109; a pitch that is a multiple of 4 floats lets the bf16 kernels pull it in 16-byte groups).  dP_h = dO_h V_h^T + gmean/H.  delta_ws: caller-owned (B,H,T) fp32 scratch.
 * dq/dk/dv use the q/k/v strides (they may alias slices of one packed (B,T,3,H,d) buffer). */
int acr_attn_bwd(const acr_attn_desc* desc, const void* q, const void* k, const void* v,
                 const void* o, const void* d_o, const float* lse2,
                 const float* gmean, int64_t gmean_sb, int64_t gmean_st,
                 void* dq, void* dk, void* dv, float* delta_ws, void* stream);

/* ---- the same attention with RESIDENT SCORES (fp32 tensors only; acr_wsss_amd/csrc/attn_f32_sres.hip) ----
 * The exact-fp32 MFMA runs at the fp32 vector rate (157 TF) while HBM moves 6+ TB/s: recomputing a logit (128 FLOP per 4
 * bytes) costs more than reading it back.  acr_attn_fwd_scores computes exactly what acr_attn_fwd computes and additionally
 * stores the scaled base-2 logits s2 of every (b, h) ONCE into `scores` (caller-owned, acr_attn_scores_floats(desc) floats,
 * 16-byte aligned; opaque blocked layout: 32 x 32 tiles in MFMA accumulator order, keys >= T hold -inf).  The head mean
 * (pmean, as in acr_attn_fwd) is then a stream over `scores`, and acr_attn_bwd_scores -- same contract as acr_attn_bwd plus
 * the `scores` of the matching forward call -- runs 5 matrix products instead of 8.  gmean additionally has to be 16-byte
 * aligned with row pitch and batch stride multiples of 4 floats.  Replaces the same reference lines as acr_attn_fwd /
 * acr_attn_bwd (models/vision_transformer.py:203-211 and its autograd backward; there P itself, (B,H,T,T), is what stays
 * resident between forward and backward).
 *
 * desc->dtype = ACR_F32_BF16X3 (csrc/attn_f32_x3.hip): the same contract on the same fp32 tensors, with S, PV, dP, dQ, dK, dV
 * as split products on the bf16 MFMA (acr_math ACR_MATH_BF16X3); softmax, lse2, delta and the head mean stay fp32.  The
 * forward splits q, k, v once into bf16 planes kept BEHIND the score blocks (acr_attn_scores_floats(desc) accounts for them:
 * nine bf16 planes of B*T*H*64 elements), the backward reads them from there (its q / k / v arguments must still be the
 * forward's tensors: when T leaves exactly ONE 32-row block beyond a whole number of workgroups -- T = 785, 1025, 2305 -- that
 * block's rows are computed by the exact-fp32 split-tail workgroups of the ACR_F32 kernels from the fp32 operands) and splits
 * d_o into planes behind delta: delta_ws must hold acr_attn_bwd_ws_floats(desc) floats
 * (B*H*T for ACR_F32) and be 16-byte aligned.  head strides must be >= 64. */
int64_t acr_attn_scores_floats(const acr_attn_desc* desc);
int64_t acr_attn_bwd_ws_floats(const acr_attn_desc* desc);
/* x (rows, cols) fp32 with row pitch ld -> three dense bf16 (rows, cols) planes p0, p1, p2 at planes + i * plane_stride
 * (elements) with x = p0 + p1 + p2 exactly (barring bf16 underflow): p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1).
 * The operand form of acr_math ACR_MATH_BF16X3.  cols %% 8 == 0, 16-byte aligned pointers. */
int acr_split3_bf16(const float* x, int64_t rows, int64_t cols, int64_t ld, void* planes, int64_t plane_stride, void* stream);
int acr_attn_fwd_scores(const acr_attn_desc* desc, const void* q, const void* k, const void* v,
                        void* o, float* lse2, float* scores, float* pmean, int64_t pmean_sb, int64_t pmean_st, void* stream);
/* acr_attn_fwd_scores (desc->dtype = ACR_F32_BF16X3 only) whose output additionally leaves as the split-product image of the
 * (B*T) x (H*64) matrix o (o dense: o_sh = 64, o_st = H*64, o_sb = T*o_st) -- what acr_x3_image(o, H*64, B*T, H*64, o_image, ...) would
 * write, bit for bit, for every row < B*T: the operand of the Linear behind the attention (models/vision_transformer.py:211-212).
 * o_image: acr_x3_image_floats(B*T, H*64) floats; the rows past B*T of the last 128-row block are written as zeros by the call
 * (the image contract; round 6 -- the caller used to zero them).  The leftover 32-row block of T = 1025, 2305, ... stays on an
 * ordinary workgroup in this call: acr_attn_fwd_oimg_offered(desc) = 1 where this entry point is the faster forward, 0 where
 * acr_attn_fwd_scores (split-tail workgroups) + acr_x3_image(o) is. */
int acr_attn_fwd_scores_oimg(const acr_attn_desc* desc, const void* q, const void* k, const void* v,
                             void* o, float* lse2, float* scores, float* pmean, int64_t pmean_sb, int64_t pmean_st, float* o_image,
                             void* stream);
int acr_attn_fwd_oimg_offered(const acr_attn_desc* desc);
int acr_attn_bwd_scores(const acr_attn_desc* desc, const void* q, const void* k, const void* v,
                        const void* o, const void* d_o, const float* lse2, const float* scores,
                        const float* gmean, int64_t gmean_sb, int64_t gmean_st,
                        void* dq, void* dk, void* dv, float* delta_ws, void* stream);

/* Materialise per-head maps for API compatibility with `Attention.get_attn()` /
 * `get_attn_gradients()` (vision_transformer.py:186-196): probs -> P (B,H,T,T) fp32 contiguous,
 * dprobs -> dO V^T (B,H,T,T) fp32 contiguous.  Not used by the fused training / GETAM paths. */
int acr_attn_probs(const acr_attn_desc* desc, const void* q, const void* k, const float* lse2,
                   float* probs, void* stream);
int acr_attn_dprobs(const acr_attn_desc* desc, const void* d_o, const void* v,
                    float* dprobs, void* stream);

/* ---- projections of the attention block (models/vision_transformer.py:200 `self.qkv(x)`, :212 `self.proj(x)`) ----
 * y[M,N] = a[M,K] . b[N,K]^T (+ bias[N]) (+ resid[M,N]); bf16 tensors, fp32 accumulate, hand-written MFMA GEMM.
 * a = activations (rows = tokens), b = nn.Linear weight as stored (N,K).  Also computes the input gradient
 * dX = dY . W when given b = W^T.  ld* are row pitches in elements (multiples of 8), K %% 64 == 0.
 * bias / resid may be NULL.  With y = the packed (B*T, 3*H*64) buffer this is the projection half of the
 * "fused qkv-projection + attention" path: acr_linear_bf16 -> acr_attn_fwd read it in place, no permute. */
int acr_linear_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, const void* bias,
                    const void* resid, int64_t ldr, void* y, int64_t ldy, int32_t M, int32_t N, int32_t K,
                    void* stream);
/* The MLP of a block with its GELU folded into the GEMM epilogues (models/vision_transformer.py:158-164):
 * acr_linear_gelu_bf16: with h = bf16(a w^T + bias) (M,N): act = GELU(h) (exact erf form) and, written to `h`, GELU'(h)
 * -- fc1 forward keeps the derivative its backward needs in place of the pre-activation (nothing else reads it);
 * acr_linear_dgelu_bf16: y = (a w^T) * h -- the input gradient of fc2 taken through the activation
 * (a = dY (M,K), w = W2^T (N,K), h = the GELU'(h) saved by acr_linear_gelu_bf16 (M,N)).  Same operand rules as
 * acr_linear_bf16. */
int acr_linear_gelu_bf16(const void* a, int64_t lda, const void* w, int64_t ldw, const void* bias, void* h, void* act,
                         int64_t ldy, int32_t M, int32_t N, int32_t K, void* stream);
int acr_linear_dgelu_bf16(const void* a, int64_t lda, const void* w, int64_t ldw, const void* h, int64_t ldh, void* y,
                          int64_t ldy, int32_t M, int32_t N, int32_t K, void* stream);


/* ---- the same Linears at the REFERENCE precision: fp32 tensors, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) ----
 * One entry point, three operand arrangements (models/vision_transformer.py:158-164,200,212 and their autograd backward):
 *   ACR_GEMM_NT  c[M,N] = a[M,K] . b[N,K]^T      Linear forward  (a = x, b = W as stored)
 *   ACR_GEMM_NN  c[M,N] = a[M,K] . b[K,N]        input gradient  (a = dy, b = W as stored -- no transposed weight copies)
 *   ACR_GEMM_TN  c[M,N] = a[K,M]^T . b[K,N]      weight gradient (a = dy, b = x, K = tokens), contraction split over
 *                workgroups into fp32 slabs summed in split order (deterministic); colsum (nullable, (M)) receives
 *                sum_k a[k][m] -- the bias gradient -- from the same sweep over dy
 * Epilogues of NT / NN (act): 0: c = acc (+ bias[N]) (+ aux[M,N] = the block's residual);
 *   1: with h = acc + bias: c2 = GELU(h) (exact erf form) and c = GELU'(h) -- fc1 forward writes the activation and, in
 *      place of the pre-activation, the derivative its backward needs (one erff serves both; h itself is used nowhere else);
 *   2: c = acc * aux with aux = the saved GELU'(h) (fc2's input gradient taken through the activation).
 * Pitches in elements, multiples of 4; pointers 16-byte aligned; K %% 4 == 0 (NT/NN), M, N %% 4 == 0 and ldc == N (TN).
 * math: acr_math (how the products are evaluated; everything else is identical).
 * ws: caller-owned scratch of acr_gemm_f32_ws_floats(mode, math, M, N, K) floats (TN: the split slabs; NT / NN: slabs for the
 * K-split tail tiles, 0 when the tile count needs none -- ws may then be NULL; without ws the product runs unsplit.
 * math = ACR_MATH_BF16X3 adds the bf16 planes both operands are split into once per call (csrc/gemm_f32.hip
 * gemm_f32_planes_kernel); with ws == NULL the operand tiles are split inside the GEMM kernel instead, same results). */
typedef enum acr_gemm_mode { ACR_GEMM_NT = 0, ACR_GEMM_NN = 1, ACR_GEMM_TN = 2 } acr_gemm_mode;
size_t acr_gemm_f32_ws_floats(int32_t mode, int32_t math, int32_t M, int32_t N, int32_t K);
int acr_gemm_f32(int32_t mode, int32_t math, int32_t act, const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias,
                 const float* aux, int64_t ldaux, float* c, int64_t ldc, float* c2, float* colsum, int32_t M, int32_t N,
                 int32_t K, float* ws, void* stream);

/* Weight gradient of a projection: dw[N,K] = dy[M,N]^T . x[M,K] (bf16, fp32 accumulate), contraction over the M tokens
 * split over workgroups with fp32 partial slabs summed in a fixed order (deterministic).  N, K multiples of 128.
 * ws: caller-owned fp32 scratch of acr_wgrad_ws_floats(M, N, K) floats (0 = shape not supported). */
size_t acr_wgrad_ws_floats(int32_t M, int32_t N, int32_t K);
int acr_wgrad_bf16(const void* dy, int64_t ldy, const void* x, int64_t ldx, int32_t M, int32_t N, int32_t K,
                   float* ws, void* dw, void* stream);
/* Weight and bias gradient in one sweep over dy: dbias[n] = sum_m dy[m][n] is accumulated from the dy fragments inside
 * the weight-gradient kernel (no second pass over dy).  ws: acr_wgrad_bias_ws_floats(M, N, K) floats. */
size_t acr_wgrad_bias_ws_floats(int32_t M, int32_t N, int32_t K);
int acr_wgrad_bias_bf16(const void* dy, int64_t ldy, const void* x, int64_t ldx, int32_t M, int32_t N, int32_t K, float* ws,
                        void* dw, void* dbias, void* stream);


/* Bias gradient of a projection: out[n] = sum_m dy[m, n], bf16 in/out, fp32 two-stage deterministic accumulation.
 * ws: caller-owned fp32 scratch of acr_colsum_ws_floats(M, N) floats.  N and ld multiples of 8. */
size_t acr_colsum_ws_floats(int32_t M, int32_t N);
int acr_colsum_bf16(const void* dy, int64_t ld, int32_t M, int32_t N, float* ws, void* out, void* stream);

/* ---- 1x1 convolutions of the ResNetV2 stem, NCHW bf16, stride 1 (models/resnetv2.py:186-190 conv1/conv3/downsample) ----
 * y[n][co][p] = sum_ci w[co][ci] x[n][ci][p] as one MFMA GEMM per sample without any layout change; the same entry
 * point gives the input gradient with w = W^T (cin/cout swapped).  cin %% 64 == 0, hw = H*W %% 8 == 0.
 * addend (nullable, shaped like y): added to the result in fp32 -- the gradient arriving over the block's shortcut.
 * acr_conv1x1_wgrad_bf16: dw[co][ci] = sum_n sum_p dy[n][co][p] x[n][ci][p] (fp32 slabs in ws, fixed-order reduction). */
int acr_conv1x1_bf16(const void* w, int64_t ldw, const void* x, const void* addend, void* y, int32_t nsamp, int32_t cout,
                     int32_t cin, int32_t hw, void* stream);
size_t acr_conv1x1_wgrad_ws_floats(int32_t nsamp, int32_t cout, int32_t cin, int32_t hw);
int acr_conv1x1_wgrad_bf16(const void* dy, const void* x, int32_t nsamp, int32_t cout, int32_t cin, int32_t hw,
                           float* ws, void* dw, void* stream);

/* The same 1x1 convolutions at the reference precision (fp32 NCHW) on the fp32 GEMM kernels, one slice per sample.
 * acr_conv1x1_f32: y[n] = W . x[n] (+ addend[n]); w_transposed = 0: w is (cout, cin); w_transposed = 1: w is stored (cin, cout),
 * i.e. the forward weight handed over as is for the input gradient (no transposed copy).  cout / cin / hw multiples of 4.
 * acr_conv1x1_wgrad_f32: dw (cout, cin) = sum_n dy[n] . x[n]^T through one fp32 slab per sample (ws: nsamp*cout*cin floats).
 * math: acr_math, as for acr_gemm_f32. */
size_t acr_conv1x1_ws_floats(int32_t math, int32_t nsamp, int32_t cout, int32_t cin, int32_t hw);
int acr_conv1x1_f32(int32_t math, const float* w, int32_t w_transposed, const float* x, const float* addend, float* y, int32_t nsamp,
                    int32_t cout, int32_t cin, int32_t hw, float* ws, void* stream);
/* acr_conv1x1_x3: the same product under ACR_MATH_BF16X3 with the weight given as a split-product image -- acr_x3_image of W
 * (cout, cin) for the forward, acr_x3_image_t of the forward's weight for the input gradient -- so that only the activation tile
 * is split inside the kernel (half the vector work per MFMA).  cin %% 16 == 0; ws as acr_conv1x1_f32 (acr_conv1x1_ws_floats). */
int acr_conv1x1_x3(const float* w_img, const float* x, const float* addend, float* y, int32_t nsamp, int32_t cout, int32_t cin, int32_t hw,
                   float* ws, void* stream);
size_t acr_conv1x1_wgrad_f32_ws_floats(int32_t nsamp, int32_t cout, int32_t cin, int32_t hw);
int acr_conv1x1_wgrad_f32(int32_t math, const float* dy, const float* x, int32_t nsamp, int32_t cout, int32_t cin, int32_t hw, float* ws,
                          float* dw, void* stream);

/* ---- split-product images: operands of acr_math ACR_MATH_BF16X3 products made ONCE and used by several products -------------
 * (a Linear's input serves its forward and its weight gradient, its output gradient the input and the weight gradient:
 * models/vision_transformer.py:158-164,200,212 and their autograd backward).  An image holds the three bf16 planes of a
 * row-major fp32 matrix x[rows][cols], tiled [128 rows][16 cols] exactly as the GEMM kernels copy them to LDS
 * (csrc/gemm_f32.hip "PRE-TILED"); zero outside the matrix, so no shape conditions beyond the alignment ones.
 *   acr_x3_image_floats(rows, cols): size of an image in floats.
 *   acr_x3_image: image of x (pitch ld floats).  colsum (nullable, (cols)) receives the column sums of x from the same pass
 *                 -- the bias gradient when x = dy -- through colsum_ws (acr_x3_colsum_ws_floats(rows, cols) floats).
 *   acr_x3_image_t: image of x^T (an image of a (cols) x (rows) matrix): what ACR_GEMM_NT needs of a weight W[out][in] to form
 *                 dx = dy . W without a transposed copy.
 *   acr_gemm_x3: c[M,N] from images, epilogues / workspace / determinism exactly as acr_gemm_f32:
 *     ACR_GEMM_NT  c = A[M,K] . B[N,K]^T   a_img = image of A (M x K), b_img = image of B (N x K)
 *     ACR_GEMM_TN  c = A[K,M]^T . B[K,N]   a_img = image of A (K x M), b_img = image of B (K x N) -- the SAME images of dy and x
 *                  the other two products read (fragments are read transposed from LDS).
 *     act (NT): 0, 1, 2 as acr_gemm_f32; and two epilogues whose output leaves the kernel AS the image the next product reads
 *       (an MLP's 4x-wide tensors never exist in fp32; N %% 8 == 0):
 *       3: with h = acc + bias: c = GELU'(h) (fp32, as act 1) and c2 = IMAGE of GELU(h) (acr_x3_image_floats(M, N) floats);
 *       4: c2 = IMAGE of acc * aux, c unused (may be NULL), colsum (nullable, (N)) = its column sums (the bias gradient of
 *          the Linear whose dy this is), deterministic.
 *   ws: acr_gemm_x3_ws_floats(mode, act, M, N, K) floats (TN: required; NT: K-split tail slabs (+ act 4: column-sum parts), may be
 *       NULL when colsum is NULL).
 * acr_gemm_f32(math = ACR_MATH_BF16X3) is these calls on images it makes in its own workspace. */
size_t acr_x3_image_floats(int32_t rows, int32_t cols);
size_t acr_x3_colsum_ws_floats(int32_t rows, int32_t cols);
int acr_x3_image(const float* x, int64_t ld, int32_t rows, int32_t cols, float* image, float* colsum, float* colsum_ws, void* stream);
int acr_x3_image_t(const float* x, int64_t ld, int32_t rows, int32_t cols, float* image, void* stream);
size_t acr_gemm_x3_ws_floats(int32_t mode, int32_t act, int32_t M, int32_t N, int32_t K);
/* Many small images in ONE launch (the stem's standardised convolution weights: W, W^T and the packed 3x3 forms).  `descs`: device
 * array of 48-byte records {const float* src; void* dst; int32 rows, K, sr, kin, sko, ski, wg0, nkb}: image `dst`
 * (acr_x3_image_floats(rows, K) floats) of the rows x K operand whose element (r, k) is src[r*sr + (k/kin)*sko + (k%kin)*ski]
 * (K, kin multiples of 8; nkb = ceil(K / 16)); record i owns workgroups wg0 .. wg0 + ceil(rows/128) * ceil(nkb/4) - 1 and `blk`
 * (device, nwg int32) maps every workgroup to its record. */
int acr_x3_image_many(const void* descs, const int32_t* blk, int32_t nwg, void* stream);
int acr_gemm_x3(int32_t mode, int32_t act, const float* a_img, const float* b_img, const float* bias, const float* aux, int64_t ldaux, float* c,
                int64_t ldc, float* c2, float* colsum, int32_t M, int32_t N, int32_t K, float* ws, void* stream);

/* ---- 3x3 stride-1 SAME convolutions of the stem's bottlenecks (models/resnetv2.py:171-216 `conv2`; std_conv.py:40-65) in NCHW fp32
 * as implicit GEMMs with split products on the bf16 MFMA (math = ACR_MATH_BF16X3 only: ACR_ERR_UNSUPPORTED otherwise -- the
 * exact-fp32 arithmetic keeps the library's Winograd kernels).  No im2col buffer, no layout change.
 * acr_conv3x3_f32: y[n][co][p] = sum_t sum_ci w_packed[co][t*cin + ci] * x[n][ci][p + off_t] (zero outside the image), with
 * w_packed (cout, 9*cin) = w.permute(0,2,3,1) of the (cout, cin, 3, 3) weight.  The input gradient is the same call on dy with
 * w_packed = w.flip(2,3).permute(1,2,3,0) as (cin, 9*cout).  cin %% 16 == 0, H*W %% 4 == 0, 16-byte aligned pointers.
 * ws: acr_conv3x3_ws_floats(...) floats or NULL (0 for launches that fill the chip: only small ones -- CAM generation on one image --
 * are split along the contraction into slabs, summed in a fixed order).
 * No byte outside [x, x + nsamp*cin*H*W) is addressed: the workgroups whose shifted tap windows touch the tensor's two ends clamp
 * their reads into it (ABI 2 rounds 3-4 asked the caller for ACR_CONV3X3_PAD floats of readable slack there; that contract is gone).
 * acr_conv3x3_wgrad_f32: dw_packed (cout, 9*cin) = sum_n sum_p dy[n][co][p] * x[n][ci][p + off_t]; ws: fp32 slabs of
 * acr_conv3x3_wgrad_ws_floats(...) floats, summed in a fixed order (deterministic). */
size_t acr_conv3x3_ws_floats(int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W);
int acr_conv3x3_f32(int32_t math, const float* w_packed, const float* x, float* y, int32_t nsamp, int32_t cout, int32_t cin,
                    int32_t H, int32_t W, float* ws, void* stream);
/* acr_conv3x3_f32 (math = ACR_MATH_BF16X3) with the packed weight given as a split-product image -- w_img = acr_x3_image of w_packed
 * (rows = cout, cols = 9*cin; for the input gradient of the flipped / role-swapped pack) --: only the activation tile is split in
 * registers, half the vector work per MFMA.  Same shapes, same ws (acr_conv3x3_ws_floats). */
int acr_conv3x3_x3(const float* w_img, const float* x, float* y, int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W, float* ws,
                   void* stream);
size_t acr_conv3x3_wgrad_ws_floats(int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W);
int acr_conv3x3_wgrad_f32(int32_t math, const float* dy, const float* x, int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W,
                          float* ws, float* dw_packed, void* stream);

/* ---- strided SAME convolutions of the stem under split products: the 7x7 stride-2 stem convolution (models/resnetv2.py:337-340 via
 * models/layers/std_conv.py:40-65) and conv2 of the first bottleneck of stages 1 and 2 (3x3 stride 2, resnetv2.py:196-199), which the
 * reference hands to cuDNN through F.conv2d on a padded copy.  Here: one space-to-depth pass, then the 3x3 kernels driven by a TAP TABLE.
 * acr_space_to_depth2_f32: xs[n][(py*2+px)*C + c][y][x] = x[n][c][2y+py][2x+px] (zero past the image), a half-resolution grid of
 *   H2 = ceil(H/2) x W2 = ceil(W/2) with `xrows` >= 4C channel rows per sample, rows 4C.. zeroed.  acr_depth_to_space2_f32: the inverse
 *   (xrows = 4C; H even, W % 8 == 0).
 * acr_conv_taps_x3: y[n][co][p] = sum_t sum_c W[co][t*cin + c] * x[n][tcb[t] + c][p + tdy[t]*W + tdx[t]], zero where (py + tdy[t], px + tdx[t])
 *   leaves the H x W grid; w_img = acr_x3_image of W (rows = cout, cols = ntap*cin); x has xrows channel rows per sample, y has yrows
 *   (>= cout: y may point at a channel slice of a larger tensor -- the input gradient writes one pixel phase per launch);
 *   tdy / tdx / tcb: HOST arrays of ntap <= 16 entries, |tdy|, |tdx| <= 7.  cin % 16 == 0, H*W % 4 == 0.  ws as acr_conv3x3_x3
 *   (acr_conv_taps_ws_floats; ignored when yrows != cout).  No byte outside [x, x + nsamp*xrows*H*W) is addressed.
 * acr_conv_taps_wgrad_f32: dw_packed[co][t*cin + c] = sum_n sum_p dy[n][co][p] * x[n][tcb[t] + c][p + tdy[t]*W + tdx[t]] (same zeros);
 *   W % 4 == 0, W >= 16, H*W % 16 == 0; ws = acr_conv_taps_wgrad_ws_floats floats (slabs, summed in a fixed order). */
int acr_space_to_depth2_f32(const float* x, float* xs, int32_t nsamp, int32_t C, int32_t H, int32_t W, int32_t xrows, void* stream);
int acr_depth_to_space2_f32(const float* xs, float* x, int32_t nsamp, int32_t C, int32_t H, int32_t W, void* stream);
size_t acr_conv_taps_ws_floats(int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W, int32_t ntap);
int acr_conv_taps_x3(const float* w_img, const float* x, float* y, int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W, int32_t ntap,
                     const int32_t* tdy, const int32_t* tdx, const int32_t* tcb, int32_t xrows, int32_t yrows, float* ws, void* stream);
size_t acr_conv_taps_wgrad_ws_floats(int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W, int32_t ntap);
int acr_conv_taps_wgrad_f32(int32_t math, const float* dy, const float* x, int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W,
                            int32_t ntap, const int32_t* tdy, const int32_t* tdx, const int32_t* tcb, int32_t xrows, float* ws, float* dw_packed,
                            void* stream);

/* ---- 3x3 stride-2 max-pool of the stem with TF-SAME -inf padding folded in (models/resnetv2.py:322-328) ----
 * x (nc, h, w) -> y (nc, ho, wo); amax = 1-byte window argmax (i*3+j, first maximum like ATen) kept for the backward,
 * which gathers (no atomics).  Window (ho, wo) starts at (2 ho - pad_top, 2 wo - pad_left). */
int acr_maxpool3x3s2_fwd_bf16(const void* x, void* y, uint8_t* amax, int64_t nc, int32_t h, int32_t w, int32_t ho,
                              int32_t wo, int32_t pad_top, int32_t pad_left, void* stream);
int acr_maxpool3x3s2_bwd_bf16(const void* dy, const uint8_t* amax, void* dx, int64_t nc, int32_t h, int32_t w,
                              int32_t ho, int32_t wo, int32_t pad_top, int32_t pad_left, void* stream);
/* the same for fp32 maps (the reference's precision) */
int acr_maxpool3x3s2_fwd_f32(const void* x, void* y, uint8_t* amax, int64_t nc, int32_t h, int32_t w, int32_t ho,
                             int32_t wo, int32_t pad_top, int32_t pad_left, void* stream);
int acr_maxpool3x3s2_bwd_f32(const void* dy, const uint8_t* amax, void* dx, int64_t nc, int32_t h, int32_t w,
                             int32_t ho, int32_t wo, int32_t pad_top, int32_t pad_left, void* stream);
/* y[nc][i][j] = x[nc][2i][2j] (Ho = ceil(h/2), Wo = ceil(w/2)): the input of a stride-2 1x1 convolution (DownsampleConv, resnetv2.py:232-249),
 * and its backward dx = dy on the even pixels, zero elsewhere.  fp32. */
int acr_subsample2_fwd_f32(const float* x, float* y, int64_t nc, int32_t h, int32_t w, void* stream);
int acr_subsample2_bwd_f32(const float* dy, float* dx, int64_t nc, int32_t h, int32_t w, void* stream);

/* ---- Fused optimizer step, bf16 weights + fp32 masters (tool/torchutils.py:10-31 PolyOptimizer = SGD with
 * momentum slot = wt_dec, weight decay 0):  mom = momentum*mom + g;  master -= lr*mom;  param = bf16(master).
 * table: device array of acr_sgd_tensor; block b of the launch updates chunk blk_chunk[b] (acr_sgd_chunk_elems()
 * elements) of tensor blk_tensor[b].  Entries with grad == NULL are skipped (parameter without gradient). */
typedef struct acr_sgd_tensor {
    const void* grad;   /* bf16 (n) or NULL */
    float* master;      /* fp32 (n) */
    float* mom;         /* fp32 (n) momentum buffer */
    void* param;        /* bf16 (n) working copy */
    int64_t n;
} acr_sgd_tensor;
int32_t acr_sgd_chunk_elems(void);
int acr_sgd_step_bf16(const void* table, const int32_t* blk_tensor, const int32_t* blk_chunk, int32_t nblocks, float lr,
                      float momentum, void* stream);
/* the same for an all-fp32 model: grad fp32, `master` is the parameter itself, `param` unused (NULL) */
int acr_sgd_step_f32(const void* table, const int32_t* blk_tensor, const int32_t* blk_chunk, int32_t nblocks, float lr,
                     float momentum, void* stream);

/* ---- Batched transpose: dst_i (cols, rows) = src_i (rows, cols)^T for a table of bf16 matrices in one launch (the
 * (in, out) copies of the block weights that the input-gradient GEMMs read; refreshed once per optimizer step).
 * rows and cols multiples of 8; block b transposes 64x64 tile (b - tile0) of tensor blk_tensor[b]. */
typedef struct acr_tr_tensor {
    const void* src;
    void* dst;
    int32_t rows, cols;
    int32_t tile0;      /* index of the tensor's first tile in the launch */
    int32_t tiles_c;    /* ceil(cols / 64) */
} acr_tr_tensor;
int acr_transpose_many_bf16(const void* table, const int32_t* blk_tensor, int32_t nblocks, void* stream);
/* the same for fp32 matrices (rows and cols multiples of 4) */
int acr_transpose_many_f32(const void* table, const int32_t* blk_tensor, int32_t nblocks, void* stream);

/* ---- LayerNorm of the transformer blocks (models/vision_transformer.py:219-222,299), bf16 (M, C) rows ----
 * C a multiple of 256, <= 1024.  stats: (M*2) fp32 [mean, rstd].  Backward writes dx, dgamma, dbeta in one pass over
 * x and dy; ws: fp32 scratch of acr_layernorm_ws_floats(M, C) floats (per-wave partials, summed in wave order).
 * dskip (nullable, (M, C) bf16): gradient arriving over the residual connection x -> x + f(LN(x)); it is added to dx
 * in fp32 before the single bf16 rounding, which replaces autograd's separate accumulation pass. */
size_t acr_layernorm_ws_floats(int32_t M, int32_t C);
int acr_layernorm_fwd_bf16(const void* x, const void* gamma, const void* beta, void* y, float* stats, int32_t M,
                           int32_t C, float eps, void* stream);
int acr_layernorm_bwd_bf16(const void* dy, const void* x, const void* gamma, const float* stats, const void* dskip,
                           void* dx, float* ws, void* dgamma, void* dbeta, int32_t M, int32_t C, void* stream);

/* fp32 rows (reference precision), same contract with float tensors. */
int acr_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* stats, int32_t M, int32_t C,
                          float eps, void* stream);
int acr_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* stats, const float* dskip, float* dx,
                          float* ws, float* dgamma, float* dbeta, int32_t M, int32_t C, void* stream);
/* LayerNorm whose output leaves as a split-product image (acr_x3_image layout, rows = M, cols = C: the operand of the Linear that
 * follows -- norm1 -> attn.qkv, norm2 -> mlp.fc1, vision_transformer.py:219-226): no fp32 y is written.  stats as acr_layernorm_fwd_f32
 * (the backward is acr_layernorm_bwd_f32 on x and stats).  C in {256, 512, 768, 1024}. */
int acr_layernorm_image_f32(const float* x, const float* gamma, const float* beta, float* image, float* stats, int32_t M, int32_t C, float eps,
                            void* stream);

/* ---- token assembly of the hybrid ViT (vision_transformer.py:449-467: flatten(2).transpose(1, 2), cat(cls[, dist], x), + pos_embed) ----
 * acr_tokens_fwd_f32: tok[b][P+t][d] = y[b][d][t] + bias[d] + pos[P+t][d], tok[b][p][d] = prefix[p][d] + pos[p][d] for p < P;
 *   y (B, D, T) = the patch projection's output WITHOUT its bias, prefix (P, D) = class (+ distillation) token, pos (P+T, D), P <= 8.
 * acr_tokens_bwd_f32: dy[b][d][t] = dtok[b][P+t][d], dpos[r][d] = sum_b dtok[b][r][d] in batch order (the bias gradient is the sum of
 *   dpos rows >= P, the prefix gradient its first P rows). */
int acr_tokens_fwd_f32(const float* y, const float* bias, const float* prefix, const float* pos, float* tok, int32_t B, int32_t D, int32_t T,
                       int32_t P, void* stream);
int acr_tokens_bwd_f32(const float* dtok, float* dy, float* dpos, int32_t B, int32_t D, int32_t T, int32_t P, void* stream);

/* ---- ResNetV2 stem: fused GroupNorm(32) [+ residual] [+ ReLU], bf16 NCHW ----
 * models/layers/norm_act.py:69-85 (GroupNormAct), models/resnetv2.py:205-215 (norm3 -> act3(x + shortcut)).
 * act: 0 = none, 1 = ReLU, 2 = ReLU(gn(x) + resid).  x/resid/y: (N,C,H,W) contiguous, HW = H*W (multiple of 8),
 * C a multiple of 32; stats: (N*32*2) fp32 [mean, rstd] written by forward, read by backward.
 * Backward writes dx (and dresid for act 2) plus per-sample partials dgamma_part/dbeta_part (N,C) fp32; with
 * dgamma/dbeta (bf16 (C), nullable as a pair) it also sums them over the samples, in sample order. */
int acr_groupnorm_fwd_bf16(const void* x, const void* resid, const void* gamma, const void* beta, void* y,
                           float* stats, int32_t N, int32_t C, int32_t HW, float eps, int32_t act, void* stream);
int acr_groupnorm_bwd_bf16(const void* dy, const void* x, const void* resid, const void* gamma, const void* beta,
                           const float* stats, void* dx, void* dresid, float* dgamma_part, float* dbeta_part,
                           void* dgamma, void* dbeta, int32_t N, int32_t C, int32_t HW, int32_t act, void* stream);

/* The same at the reference precision (fp32 NCHW tensors, fp32 gamma / beta / gradients): a group of up to 100 352 floats (every
 * group of the 448^2 step; 50 176 in the backward, which holds x and dy) is read ONCE into the registers of a 1024-thread workgroup
 * (round 6), larger ones are streamed twice; HW a multiple of 4; deterministic (fixed-order block reductions, no atomics).
 * _mask_ variants (act 2 only): the forward also writes the ReLU mask of relu(gn(x) + resid), ONE BYTE per 16-byte vector of y
 * (relu_mask: N*C*HW/4 bytes, bit e = element e of the vector was positive), and the backward reads those bytes INSTEAD of the
 * residual -- it needs the residual for nothing else (same expression in both directions: the mask is the one the forward applied).
 * ws (forward; nullable): acr_groupnorm_fwd_ws_floats(N, C, HW) floats -- non-zero only for launches of a few samples (CAM
 * generation on one image), whose (sample, group) pairs are then cut into parts over two launches. */
size_t acr_groupnorm_fwd_ws_floats(int32_t N, int32_t C, int32_t HW);
int acr_groupnorm_fwd_f32(const float* x, const float* resid, const float* gamma, const float* beta, float* y, float* stats,
                          int32_t N, int32_t C, int32_t HW, float eps, int32_t act, float* ws, void* stream);
int acr_groupnorm_bwd_f32(const float* dy, const float* x, const float* resid, const float* gamma, const float* beta,
                          const float* stats, float* dx, float* dresid, float* dgamma_part, float* dbeta_part, float* dgamma,
                          float* dbeta, int32_t N, int32_t C, int32_t HW, int32_t act, void* stream);
int acr_groupnorm_fwd_mask_f32(const float* x, const float* resid, const float* gamma, const float* beta, float* y, float* stats,
                               int32_t N, int32_t C, int32_t HW, float eps, uint8_t* relu_mask, void* stream);
int acr_groupnorm_bwd_mask_f32(const float* dy, const float* x, const uint8_t* relu_mask, const float* gamma, const float* beta,
                               const float* stats, float* dx, float* dresid, float* dgamma_part, float* dbeta_part, float* dgamma,
                               float* dbeta, int32_t N, int32_t C, int32_t HW, void* stream);

/* Weight standardisation of all StdConv2dSame weights of the stem in one launch (models/layers/std_conv.py:56-59).
 * desc_dev: device array of n_conv records {uint64 p0,p1,p2,p3; int32 cout, n, ch_start, pad} sorted by ch_start
 * (first global output-channel index of the conv), n = fan-in.  forward (backward = 0): p0 = w, p1 = w_hat out, p3
 * (nullable) = w_hat^T (n, cout) out -- the copy a 1x1 convolution's input-gradient GEMM reads;
 * backward: p0 = w, p1 = dL/dw_hat, p2 = dL/dw out.  bf16 tensors, fp32 statistics. */
int acr_weight_std_bf16(const void* desc_dev, int32_t n_conv, int32_t total_channels, float eps, int32_t backward,
                        void* stream);
int acr_weight_std_f32(const void* desc_dev, int32_t n_conv, int32_t total_channels, float eps, int32_t backward,
                       void* stream);                /* fp32 tensors, same descriptor table */

/* ---- multilabel soft-margin loss of the class logits (train_acr.py:160-161: F.multilabel_soft_margin_loss(x, label) per view) ----
 * x, y: (N, C) fp32 with row pitches ldx, ldy (elements); loss[0] = mean_n (1/C) sum_c -(y ls(x) + (1 - y) ls(-x)), ls = log-sigmoid.
 * Backward: dx (N, C) dense = g[0] * (sigmoid(x) - y) / (N C), g = the upstream gradient of the scalar loss, in DEVICE memory.
 * One launch each way (the stock op: ~20 launches per view and direction); deterministic. */
int acr_mlsm_fwd_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, int32_t N, int32_t C, float* loss, void* stream);
int acr_mlsm_bwd_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* g, int32_t N, int32_t C, float* dx, void* stream);

/* ---- attention-consistency regulariser (train_acr.py:143-161, inline in train()) ----
 * a1, a2: (B,L,T,T) fp32 head-mean stacks of view 1 / view 2 (T = p*p + 1), batch stride a_sb each
 * (so both may live in one (2B,L,T,T) buffer).  With pi(i*p+j) = i*p+(p-1-j):
 *   out[0] = mean_{b,l,c}   |a1[b,l,0,1+c]   - a2[b,l,0,1+pi(c)]|          (cls_align_loss, :160)
 *   out[1] = mean_{b,l,r,c} |a1[b,l,1+r,1+c] - a2[b,l,1+pi(r),1+pi(c)]|    (aff_align_loss, :161)
 * equal to the reference's 3*p in-place block flips (:151-158) followed by two F.l1_loss.
 * partial_ws: caller-owned fp32 scratch of acr_consistency_ws_floats(B,L,T) floats.  Deterministic
 * (fixed-order two-stage reduction, no float atomics). */
size_t acr_consistency_ws_floats(int32_t B, int32_t L, int32_t T);
int acr_consistency_fwd(const float* a1, const float* a2, int64_t a_sb, int32_t B, int32_t L,
                        int32_t T, int32_t p, float* partial_ws, float* out2, void* stream);
/* gout2: device pointer to the two upstream gradients (d/d out[0], d/d out[1]).  Writes
 * g1, g2: (B,L,T,g_st) with row pitch g_st >= T and batch stride g_sb >= L*T*g_st; every element of
 * the T x T maps is written (zeros where the loss does not look); pad columns are left untouched. */
int acr_consistency_bwd(const float* a1, const float* a2, int64_t a_sb, int32_t B, int32_t L,
                        int32_t T, int32_t p, const float* gout2, float* g1, float* g2,
                        int64_t g_sb, int64_t g_st, void* stream);

/* ---- GETAM (DPT/ACR.py:177-215 `ACR.getam`) ----
 * Adds one layer's contribution to cam_row (T fp32, caller zero-initialised):
 *   cam_row[j] += f_h( dP_h[batch,0,j], P_h[batch,0,j] ),  dP_h = dO_h V_h^T (row 0 only)
 * for `func` in acr_getam_func.  Only row 0 of the layer-summed map is consumed by the reference
 * (:213), so nothing else is computed.  The final relu and the [1:] / [2:] slice are the caller's. */
int acr_getam_row_accum(const acr_attn_desc* desc, const void* q, const void* k, const void* v,
                        const void* d_o, const float* lse2, int32_t batch, int32_t func,
                        float* cam_row, void* stream);
/* The same for every sample of the batch in ONE launch (CAM generation batches the flipped and the plain pass and several
 * images: infer_cam.py:147-153 runs them one by one): cam_rows[b*row_stride + j] += f_h(...) of sample b, b = 0 .. desc->B-1. */
int acr_getam_rows_accum(const acr_attn_desc* desc, const void* q, const void* k, const void* v, const void* d_o,
                         const float* lse2, int32_t func, float* cam_rows, int64_t row_stride, void* stream);

/* Affinity refinement (infer_cam.py:164-165,183-184): out[c][r] = sum_l sum_k a[l][1+r][1+k] cam[c][k]
 * for one sample's (L,T,T) head-mean stack `a`, n_cam row vectors cam (n_cam, T-1) -> out (n_cam, T-1). */
int acr_aff_refine(const float* a, int32_t L, int32_t T, const float* cam, int32_t n_cam,
                   float* out, void* stream);
/* nbatch samples in one launch: a[s] = a + s * a_sb, cam / out (nbatch, n_cam, T-1) contiguous. */
int acr_aff_refine_batch(const float* a, int64_t a_sb, int32_t L, int32_t T, const float* cam, int32_t n_cam,
                         int32_t nbatch, float* out, void* stream);

/* ---- input pipeline (myTool.py:1158-1199 get_data_from_chunk_v2, :1364-1403 get_data_from_chunk_val) ----
 * One launch turns a batch of decoded uint8 HWC RGB images into the (B,3,S,S) network input: bilinear resize with
 * cv2.resize's float-path INTER_LINEAR rule (RandomResizeLong :995-1008 / cv2.resize(S,S)), optional horizontal flip
 * (:895-899), (x/255 - mean)/std (:1180-1182) and the zero-padded RandomCrop (:923-955).  The HOST decodes and draws the
 * geometry (the reference uses the unseeded `random` module); packed_u8 holds the images back to back, `table` is a
 * device array of `batch` acr_pre_image records.  Validation batches: rh = rw = S, flip 0, boxes = the whole image.
 * mean3 / std3: HOST pointers to 3 floats.  out_dtype: ACR_F32 or ACR_BF16. */
typedef struct acr_pre_image {
    int64_t offset;        /* byte offset of this image inside packed_u8 */
    int32_t h, w;          /* decoded height, width */
    int32_t rh, rw;        /* height, width after the resize step */
    int32_t flip;          /* 1 = flip horizontally after resizing */
    int32_t cont_top, cont_left, img_top, img_left, ch, cw;   /* RandomCrop: container / image corners, copied extent */
} acr_pre_image;
int acr_preprocess_batch(const void* packed_u8, const void* table, int32_t batch, int32_t S, const float* mean3,
                         const float* std3, int32_t out_dtype, void* out, void* stream);

/* ---- dense-CRF refinement of CAMs (tool/imutils.py:345-362 crf_inference = pydensecrf DenseCRF2D with
 * addPairwiseGaussian + addPairwiseBilateral + inference(t), called by infer_cam.py:27-40,218-225) ----
 * The permutohedral lattice follows wrapper/bilateralfilter/permutohedral.cpp:112-283 (init) and :441-520 (compute).
 * acr_lattice_build: lattice over the H x W pixels with features (x / sxy, y / sxy) when rgb is NULL (d = 2,
 * addPairwiseGaussian) or (x / sxy, y / sxy, r / srgb, g / srgb, b / srgb) for rgb = (H, W, 3) uint8 (d = 5,
 * addPairwiseBilateral; bilateralfilter.cpp:4-20).  ws = acr_lattice_ws_bytes(H * W, d) bytes of device memory (opaque).
 * acr_lattice_info synchronises the stream and returns the number of lattice points (ACR_ERR_INVALID if a key did not
 * fit the 12-bit-per-coordinate packing).  acr_lattice_tables returns device pointers INTO ws: offsets (n, d+1) int32,
 * weights (n, d+1) float, point_keys (n_points) uint64 (coordinate c of a key = ((k >> 12 (d-1-c)) & 4095) - 2048).
 * acr_lattice_filter: out[k][p] = post_scale * post[p] * sum_q kernel(p, q) * pre[q] * in[k][q] for K planes of n pixels
 * (Permutohedral::compute; pre / post nullable, post_scale applied when apply_scale != 0); vals = scratch of
 * 2 * K * (n_points + 2) floats.  Sums run in the CPU code's order: results equal the reference lattice bit for bit. */
int64_t acr_lattice_ws_bytes(int32_t n_pixels, int32_t d);
int acr_lattice_build(const void* rgb, int32_t H, int32_t W, float sxy, float srgb, void* ws, int64_t ws_bytes, void* stream);
int acr_lattice_info(const void* ws, int32_t* n_points, int32_t* key_overflow, void* stream);
int acr_lattice_tables(void* ws, int32_t n_pixels, int32_t d, void** offsets, void** weights, void** point_keys);
int acr_lattice_filter(const void* ws, int32_t n_pixels, int32_t d, int32_t n_points, const void* in, const void* pre, void* out,
                       const void* post, float post_scale, int32_t apply_scale, int32_t K, void* vals, void* stream);
/* mean field (densecrf v2: unary_from_softmax, DenseKernel NORMALIZE_SYMMETRIC, DenseCRF::inference + expAndNormalize):
 * unary = -log(clamp(probs, clip, 1)); norm = 1 / sqrt(norm + 1e-20) in place; q[:, p] = softmax_k(-unary + msg0 + msg1)
 * over K planes of n pixels (msg0 / msg1 nullable). */
int acr_crf_unary(const void* probs, void* unary, int64_t n, float clip, void* stream);
int acr_crf_norm(void* norm, int32_t n_pixels, void* stream);
int acr_crf_update(const void* unary, const void* msg0, const void* msg1, void* q, int32_t n_pixels, int32_t K, void* stream);

/* ---- CAM read-outs ----
 * Patch-token -> class activation (DPT/ACR.py:133-134): out[n][c] = relu(x[n,:] . w[c,:] + bias[c]),
 * x (N, D) with row stride x_st, w (C, D) contiguous, out (N, C) contiguous; dtype of x/w/bias. */
int acr_patch_cam(const void* x, int64_t x_st, const void* w, const void* bias, int32_t N, int32_t D,
                  int32_t C, int32_t dtype, float* out, void* stream);
/* Bilinear resize of a (C, ih, iw) fp32 map given as src[(y*iw + x)*src_sp + c*src_sc] to (C, oh, ow)
 * contiguous, torch semantics (infer_cam.py:157 align_corners=0; :187 align_corners=1), then optional
 * per-channel multiply (label mask, :158; chan_mul nullable) and horizontal flip (:159-160,195-196).
 * If accumulate != 0 the result is added to dst (sum over flips/scales, :201,208). */
int acr_bilinear_resize(const float* src, int64_t src_sc, int64_t src_sp, int32_t C, int32_t ih,
                        int32_t iw, float* dst, int32_t oh, int32_t ow, int32_t align_corners,
                        const float* chan_mul, int32_t hflip, int32_t accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACR_HIP_H */
