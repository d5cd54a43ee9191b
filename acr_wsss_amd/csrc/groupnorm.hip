// Fused GroupNorm(32) [+ residual add] [+ ReLU] for the ResNetV2 stem of the hybrid patch embedding
// (models/layers/norm_act.py:69-85 `GroupNormAct`, models/resnetv2.py:205-215 `norm3` -> `act3(x + shortcut)`),
// bf16 NCHW, forward and backward.
//
// HBM-bound: the stock path costs ~15 ms of an 80 ms step (row-moments kernel + normalise kernel + clamp kernel
// forward; three gradient kernels + threshold backward), each a full pass over up to 411 MB activations.  Here one
// workgroup owns one (sample, group) -- a contiguous run of cg*H*W elements in NCHW -- and keeps it in registers
// (<= 13 x 8 bf16 per thread at 1024 threads), so forward is 1 read + 1 write and backward 2 (3 with residual)
// reads + 1 (2) writes, with exact two-pass statistics in fp32 and no re-read for the normalise / dx passes.
// ReLU's mask is recomputed in backward from x (and the residual), so y is not kept alive for it.
// d(gamma), d(beta): per-(sample, channel) partials -- every wave reduces its lanes per channel with a shuffle butterfly
// (a wave's 64 consecutive vectors span one or two channels) into its own LDS row, the rows are summed in wave order and the
// caller sums over samples in a fixed order: no atomics, bit-reproducible run to run.
#include "acr_common.h"

typedef __bf16 bf16_t;
#define GN_MAXV 13
#define GN_GROUPS 32
enum { GN_ACT_NONE = 0, GN_ACT_RELU = 1, GN_ACT_ADD_RELU = 2 };

template <int NT>
__device__ __forceinline__ float gn_block_sum(float v, float* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                                        // sh may still be read from a previous reduction
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += sh[w];           // same fixed order in every thread
    return t;
}

template <int NT, int ACT>
__global__ __launch_bounds__(NT) void gn_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                    const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                    bf16_t* __restrict__ y, float* __restrict__ stats, int C, int HW,
                                                    int cg, float eps) {
    __shared__ float sh[NT / 64];
    const int g = blockIdx.x % GN_GROUPS, n = blockIdx.x / GN_GROUPS;
    const int64_t base = ((int64_t)n * C + (int64_t)g * cg) * HW;
    const int nvec = (cg * HW) >> 3, vpc = HW >> 3;
    const float inv_n = 1.f / (float)(cg * HW);
    const int tid = threadIdx.x;
    bf16x8 xv[GN_MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < GN_MAXV; ++i) {
        const int v = tid + i * NT;
        xv[i] = *reinterpret_cast<const bf16x8*>(x + base + (int64_t)min(v, nvec - 1) * 8);
        if (v < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)xv[i][e];
        }
    }
    const float mean = gn_block_sum<NT>(sum, sh) * inv_n;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < GN_MAXV; ++i) {
        if (tid + i * NT < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)xv[i][e] - mean; ss = fmaf(d, d, ss); }
        }
    }
    const float rstd = rsqrtf(gn_block_sum<NT>(ss, sh) * inv_n + eps);
#pragma unroll
    for (int i = 0; i < GN_MAXV; ++i) {
        const int v = tid + i * NT;
        if (v < nvec) {
            const int c = g * cg + v / vpc;
            const float ga = (float)gamma[c] * rstd;
            const float be = (float)beta[c] - mean * ga;
            bf16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ACT == GN_ACT_ADD_RELU) rv = *reinterpret_cast<const bf16x8*>(res + base + (int64_t)v * 8);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = fmaf((float)xv[i][e], ga, be);
                if (ACT == GN_ACT_ADD_RELU) t += (float)rv[e];
                if (ACT != GN_ACT_NONE) t = fmaxf(t, 0.f);
                o[e] = (bf16_t)t;
            }
            *reinterpret_cast<bf16x8*>(y + base + (int64_t)v * 8) = o;
        }
    }
    if (tid == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
}

template <int NT, int ACT>
__global__ __launch_bounds__(NT) void gn_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                    const bf16_t* __restrict__ res, const bf16_t* __restrict__ gamma,
                                                    const bf16_t* __restrict__ beta, const float* __restrict__ stats,
                                                    bf16_t* __restrict__ dx, bf16_t* __restrict__ dres,
                                                    float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                                                    int C, int HW, int cg) {
    __shared__ float sh[NT / 64];
    __shared__ float dgw[NT / 64][64], dbw[NT / 64][64];      // [wave][channel of the group]
    const int g = blockIdx.x % GN_GROUPS, n = blockIdx.x / GN_GROUPS;
    const int64_t base = ((int64_t)n * C + (int64_t)g * cg) * HW;
    const int nvec = (cg * HW) >> 3, vpc = HW >> 3;
    const float inv_n = 1.f / (float)(cg * HW);
    const int tid = threadIdx.x;
    const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
    const int wave = tid >> 6, lane = tid & 63;
    dgw[wave][lane] = 0.f;
    dbw[wave][lane] = 0.f;
    bf16x8 xv[GN_MAXV], gv[GN_MAXV];
#pragma unroll
    for (int i = 0; i < GN_MAXV; ++i) {
        const int64_t off = base + (int64_t)min(tid + i * NT, nvec - 1) * 8;
        xv[i] = *reinterpret_cast<const bf16x8*>(x + off);
        gv[i] = *reinterpret_cast<const bf16x8*>(dy + off);
    }
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < GN_MAXV; ++i) {
        const int v = tid + i * NT;
        const int cl = v < nvec ? v / vpc : -1;
        float dgl = 0.f, dbl = 0.f;
        if (v < nvec) {
            const float gam = (float)gamma[g * cg + cl];
            const float bet = (float)beta[g * cg + cl];
            bf16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ACT == GN_ACT_ADD_RELU) rv = *reinterpret_cast<const bf16x8*>(res + base + (int64_t)v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = ((float)xv[i][e] - mean) * rstd;
                float gg = (float)gv[i][e];
                if (ACT != GN_ACT_NONE) {
                    // the forward's exact expression, so the ReLU mask is the one forward applied
                    const float ga = gam * rstd;
                    float pre = fmaf((float)xv[i][e], ga, bet - mean * ga);
                    if (ACT == GN_ACT_ADD_RELU) pre += (float)rv[e];
                    if (!((float)(bf16_t)fmaxf(pre, 0.f) > 0.f)) gg = 0.f;
                    gv[i][e] = (bf16_t)gg;
                }
                dbl += gg;
                dgl = fmaf(gg, xh, dgl);
            }
            if (ACT == GN_ACT_ADD_RELU) *reinterpret_cast<bf16x8*>(dres + base + (int64_t)v * 8) = gv[i];
            s1 = fmaf(dbl, gam, s1);
            s2 = fmaf(dgl, gam, s2);
        }
        // per-channel sums of this wave's 64 vectors: one butterfly per distinct channel (1-2 of them), fixed order
        unsigned long long rem = __ballot(cl >= 0);
        while (rem) {
            const int c = __shfl(cl, __ffsll((long long)rem) - 1);
            const bool mine = cl == c;
            float sa = mine ? dgl : 0.f, sb = mine ? dbl : 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { sa += __shfl_xor(sa, off); sb += __shfl_xor(sb, off); }
            if (lane == 0) { dgw[wave][c] += sa; dbw[wave][c] += sb; }
            rem &= ~__ballot(mine);
        }
    }
    const float c1 = gn_block_sum<NT>(s1, sh) * inv_n;
    const float c2 = gn_block_sum<NT>(s2, sh) * inv_n;
#pragma unroll
    for (int i = 0; i < GN_MAXV; ++i) {
        const int v = tid + i * NT;
        if (v < nvec) {
            const float gam = (float)gamma[g * cg + v / vpc];
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = ((float)xv[i][e] - mean) * rstd;
                o[e] = (bf16_t)(rstd * (fmaf((float)gv[i][e], gam, -c1) - xh * c2));
            }
            *reinterpret_cast<bf16x8*>(dx + base + (int64_t)v * 8) = o;
        }
    }
    if (tid < cg) {                                          // gn_block_sum's barriers ordered the waves' LDS rows
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { a += dgw[w][tid]; b += dbw[w][tid]; }
        dgamma_part[(int64_t)n * C + g * cg + tid] = a;
        dbeta_part[(int64_t)n * C + g * cg + tid] = b;
    }
}

static int gn_threads(int nvec) {
    if (nvec <= 256 * GN_MAXV) return 256;
    if (nvec <= 512 * GN_MAXV) return 512;
    if (nvec <= 1024 * GN_MAXV) return 1024;
    return 0;
}

static int gn_check(const char* who, int N, int C, int HW, int act) {
    ACR_CHECK_ARG(N > 0 && C > 0 && (C % GN_GROUPS) == 0 && C / GN_GROUPS <= 64, "%s: C=%d must be a multiple of 32 (<= 2048)", who, C);
    ACR_CHECK_ARG(HW > 0 && (HW % 8) == 0, "%s: H*W=%d must be a multiple of 8 (16-byte vectors per channel)", who, HW);
    ACR_CHECK_ARG(gn_threads((C / GN_GROUPS) * HW / 8) != 0, "%s: group of %d elements exceeds the register-resident limit", who, (C / GN_GROUPS) * HW);
    ACR_CHECK_ARG(act >= 0 && act <= 2, "%s: unknown act %d", who, act);
    return ACR_OK;
}

#define GN_DISPATCH(KERNEL, ...)                                                                              \
    switch (nt * 4 + act) {                                                                                    \
        case 256 * 4 + 0: hipLaunchKernelGGL((KERNEL<256, 0>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
        case 256 * 4 + 1: hipLaunchKernelGGL((KERNEL<256, 1>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
        case 256 * 4 + 2: hipLaunchKernelGGL((KERNEL<256, 2>), grid, dim3(256), 0, st, __VA_ARGS__); break;    \
        case 512 * 4 + 0: hipLaunchKernelGGL((KERNEL<512, 0>), grid, dim3(512), 0, st, __VA_ARGS__); break;    \
        case 512 * 4 + 1: hipLaunchKernelGGL((KERNEL<512, 1>), grid, dim3(512), 0, st, __VA_ARGS__); break;    \
        case 512 * 4 + 2: hipLaunchKernelGGL((KERNEL<512, 2>), grid, dim3(512), 0, st, __VA_ARGS__); break;    \
        case 1024 * 4 + 0: hipLaunchKernelGGL((KERNEL<1024, 0>), grid, dim3(1024), 0, st, __VA_ARGS__); break; \
        case 1024 * 4 + 1: hipLaunchKernelGGL((KERNEL<1024, 1>), grid, dim3(1024), 0, st, __VA_ARGS__); break; \
        default: hipLaunchKernelGGL((KERNEL<1024, 2>), grid, dim3(1024), 0, st, __VA_ARGS__); break;           \
    }

extern "C" int acr_groupnorm_fwd_bf16(const void* x, const void* resid, const void* gamma, const void* beta, void* y,
                                      float* stats, int32_t N, int32_t C, int32_t HW, float eps, int32_t act,
                                      void* stream) {
    ACR_CHECK_ARG(x && gamma && beta && y && stats && (act != GN_ACT_ADD_RELU || resid), "acr_groupnorm_fwd_bf16: null pointer");
    int rc = gn_check("acr_groupnorm_fwd_bf16", N, C, HW, act);
    if (rc) return rc;
    ACR_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)resid & 15) == 0, "acr_groupnorm_fwd_bf16: 16-byte alignment");
    const int cg = C / GN_GROUPS, nt = gn_threads(cg * HW / 8);
    const dim3 grid(N * GN_GROUPS);
    hipStream_t st = (hipStream_t)stream;
    GN_DISPATCH(gn_fwd_kernel, (const bf16_t*)x, (const bf16_t*)resid, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y,
                stats, C, HW, cg, eps)
    return acr_check_launch("acr_groupnorm_fwd_bf16");
}

// dgamma[c] = sum_n dgamma_part[n][c] (same for dbeta), samples summed in order, one bf16 rounding at the end
__global__ __launch_bounds__(256) void gn_param_reduce_kernel(const float* __restrict__ gpart, const float* __restrict__ bpart,
                                                              int N, int C, bf16_t* __restrict__ dgamma,
                                                              bf16_t* __restrict__ dbeta) {
    // 32 columns x 8 sample groups per block: group s sums samples s, s+8, ... in order, the 8 partials are combined in
    // fixed order through LDS (a single thread walking all N samples is a chain of N dependent L2 round trips)
    __shared__ float sh[8][33];
    const int cl = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;                      // over 2*C columns
    float s = 0.f;
    if (i < 2 * C) {
        const float* src = (i < C) ? gpart + i : bpart + (i - C);
        for (int n = sg; n < N; n += 8) s += src[(int64_t)n * C];
    }
    sh[sg][cl] = s;
    __syncthreads();
    if (sg == 0 && i < 2 * C) {
        float t = sh[0][cl];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sh[k][cl];
        if (i < C) dgamma[i] = (bf16_t)t;
        else dbeta[i - C] = (bf16_t)t;
    }
}

extern "C" int acr_groupnorm_bwd_bf16(const void* dy, const void* x, const void* resid, const void* gamma,
                                      const void* beta, const float* stats, void* dx, void* dresid,
                                      float* dgamma_part, float* dbeta_part, void* dgamma, void* dbeta, int32_t N,
                                      int32_t C, int32_t HW, int32_t act, void* stream) {
    ACR_CHECK_ARG(dy && x && gamma && beta && stats && dx && dgamma_part && dbeta_part &&
                      (act != GN_ACT_ADD_RELU || (resid && dresid)),
                  "acr_groupnorm_bwd_bf16: null pointer");
    int rc = gn_check("acr_groupnorm_bwd_bf16", N, C, HW, act);
    if (rc) return rc;
    const int cg = C / GN_GROUPS, nt = gn_threads(cg * HW / 8);
    const dim3 grid(N * GN_GROUPS);
    hipStream_t st = (hipStream_t)stream;
    GN_DISPATCH(gn_bwd_kernel, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)resid, (const bf16_t*)gamma,
                (const bf16_t*)beta, stats, (bf16_t*)dx, (bf16_t*)dresid, dgamma_part, dbeta_part, C, HW, cg)
    if (dgamma && dbeta)
        hipLaunchKernelGGL(gn_param_reduce_kernel, dim3((2 * C + 31) / 32), dim3(256), 0, st, (const float*)dgamma_part,
                           (const float*)dbeta_part, N, C, (bf16_t*)dgamma, (bf16_t*)dbeta);
    return acr_check_launch("acr_groupnorm_bwd_bf16");
}
