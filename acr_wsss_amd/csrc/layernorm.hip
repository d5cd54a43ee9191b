// LayerNorm of the transformer blocks (models/vision_transformer.py:219,222 norm1/norm2, eps 1e-6 :299), bf16 rows,
// forward and backward.  HBM-bound row kernels: one wave per row (C/64 elements per lane, 8-byte vectors -> 512
// contiguous bytes per wave instruction), exact two-pass statistics in fp32 from registers.
//   forward : 1 read + 1 write of the activation (stock kernel: 42 us for 25120 x 768; this: ~2x fewer)
//   backward: ONE pass reads x and dy and writes dx while the wave accumulates its rows' contributions to dgamma/dbeta
//             in registers; per-wave partials are summed in wave order by a second kernel (deterministic).  The stock
//             path reads x and dy twice (grad-input kernel + two gamma/beta kernels).
#include "acr_common.h"

typedef __bf16 bf16_t;
#define LN_MAXV 4                     // up to 4 vectors of 4 elements per lane -> C <= 1024
#define LN_WAVES 4

__device__ __forceinline__ float ln_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <int NV, typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ gamma,
                                                     const T* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ stats, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * LN_WAVES + (threadIdx.x >> 6), nw = gridDim.x * LN_WAVES;
    const float inv_c = 1.f / (float)C;
    f32x4 ga[NV], be[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ga[i] = acr_load4<T>(gamma + (i * 64 + lane) * 4);
        be[i] = acr_load4<T>(beta + (i * 64 + lane) * 4);
    }
    for (int row = gw; row < M; row += nw) {
        const T* xr = x + (int64_t)row * C;
        f32x4 xv[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            xv[i] = acr_load4<T>(xr + (i * 64 + lane) * 4);
            s += xv[i][0] + xv[i][1] + xv[i][2] + xv[i][3];
        }
        const float mean = ln_wave_sum(s) * inv_c;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = xv[i][e] - mean; ss = fmaf(d, d, ss); }
        const float rstd = rsqrtf(ln_wave_sum(ss) * inv_c + eps);
        T* yr = y + (int64_t)row * C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf((xv[i][e] - mean) * rstd, ga[i][e], be[i][e]);
            acr_store4<T>(yr + (i * 64 + lane) * 4, o);
        }
        if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    }
}

template <int NV, typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const T* __restrict__ gamma, const float* __restrict__ stats,
                                                     const T* __restrict__ dskip, T* __restrict__ dx,
                                                     float* __restrict__ part, int M, int C) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * LN_WAVES + (threadIdx.x >> 6), nw = gridDim.x * LN_WAVES;
    const float inv_c = 1.f / (float)C;
    f32x4 ga[NV], dg[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ga[i] = acr_load4<T>(gamma + (i * 64 + lane) * 4);
        dg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        db[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int row = gw; row < M; row += nw) {
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        f32x4 xh[NV], g[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 xv = acr_load4<T>(x + (int64_t)row * C + (i * 64 + lane) * 4);
            const f32x4 dv = acr_load4<T>(dy + (int64_t)row * C + (i * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[i][e] = (xv[e] - mean) * rstd;
                g[i][e] = dv[e] * ga[i][e];
                s1 += g[i][e];
                s2 = fmaf(g[i][e], xh[i][e], s2);
                dg[i][e] = fmaf(dv[e], xh[i][e], dg[i][e]);
                db[i][e] += dv[e];
            }
        }
        const float c1 = ln_wave_sum(s1) * inv_c, c2 = ln_wave_sum(s2) * inv_c;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rstd * (g[i][e] - c1 - xh[i][e] * c2);
            if (dskip) o += acr_load4<T>(dskip + (int64_t)row * C + (i * 64 + lane) * 4);   // skip-path gradient
            acr_store4<T>(dx + (int64_t)row * C + (i * 64 + lane) * 4, o);
        }
    }
    // the 4 waves of the workgroup are summed in wave order through LDS -> one (2, C) partial per workgroup
    __shared__ float sh[LN_WAVES][2 * 256 * LN_MAXV];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        *reinterpret_cast<f32x4*>(&sh[wv][(i * 64 + lane) * 4]) = dg[i];
        *reinterpret_cast<f32x4*>(&sh[wv][C + (i * 64 + lane) * 4]) = db[i];
    }
    __syncthreads();
    float* pg = part + (int64_t)blockIdx.x * 2 * C;
    for (int c = threadIdx.x; c < 2 * C; c += 256) pg[c] = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
}

// 32 columns x 8 row groups per block: each thread sums its share of the partials with 8 independent loads in flight,
// the 8 groups are combined in a fixed order through LDS.
template <typename T>
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ part, int np, int C,
                                                              T* __restrict__ dgamma, T* __restrict__ dbeta) {
    __shared__ float sh[8][32];
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;                        // over 2*C columns (2*C % 32 == 0)
    float s = 0.f;
    const int per = (np + 7) / 8, w0 = grp * per, w1 = min(w0 + per, np);
    int w = w0;
    for (; w + 8 <= w1; w += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = part[(int64_t)(w + u) * 2 * C + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; w < w1; ++w) s += part[(int64_t)w * 2 * C + c];
    sh[grp][cl] = s;
    __syncthreads();
    if (grp == 0) {
        float tot = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) tot += sh[u][cl];
        if (c < C) acr_store1<T>(dgamma + c, tot); else acr_store1<T>(dbeta + c - C, tot);
    }
}

static int ln_grid(int M) {
    int g = (M + LN_WAVES - 1) / LN_WAVES;
    return g < 1024 ? g : 1024;                                 // 4096 waves: ~6 rows each at M = 25120
}
extern "C" size_t acr_layernorm_ws_floats(int32_t M, int32_t C) { return (size_t)ln_grid(M) * 2 * (size_t)C; }

static int ln_check(const char* who, int M, int C) {
    ACR_CHECK_ARG(M > 0 && C > 0 && (C % 256) == 0 && C <= 256 * LN_MAXV, "%s: C=%d must be a multiple of 256 and <= 1024", who, C);
    return ACR_OK;
}
#define LN_DISPATCH(KERNEL, TT, ...)                                                                  \
    switch (C / 256) {                                                                                 \
        case 1: hipLaunchKernelGGL((KERNEL<1, TT>), grid, dim3(256), 0, st, __VA_ARGS__); break;       \
        case 2: hipLaunchKernelGGL((KERNEL<2, TT>), grid, dim3(256), 0, st, __VA_ARGS__); break;       \
        case 3: hipLaunchKernelGGL((KERNEL<3, TT>), grid, dim3(256), 0, st, __VA_ARGS__); break;       \
        default: hipLaunchKernelGGL((KERNEL<4, TT>), grid, dim3(256), 0, st, __VA_ARGS__); break;      \
    }

extern "C" int acr_layernorm_fwd_bf16(const void* x, const void* gamma, const void* beta, void* y, float* stats,
                                      int32_t M, int32_t C, float eps, void* stream) {
    ACR_CHECK_ARG(x && gamma && beta && y && stats, "acr_layernorm_fwd_bf16: null pointer");
    int rc = ln_check("acr_layernorm_fwd_bf16", M, C);
    if (rc) return rc;
    const dim3 grid(ln_grid(M));
    hipStream_t st = (hipStream_t)stream;
    LN_DISPATCH(ln_fwd_kernel, bf16_t, (const bf16_t*)x, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, stats, M, C, eps)
    return acr_check_launch("acr_layernorm_fwd_bf16");
}

extern "C" int acr_layernorm_bwd_bf16(const void* dy, const void* x, const void* gamma, const float* stats,
                                      const void* dskip, void* dx, float* ws, void* dgamma, void* dbeta, int32_t M,
                                      int32_t C, void* stream) {
    ACR_CHECK_ARG(dy && x && gamma && stats && dx && ws && dgamma && dbeta, "acr_layernorm_bwd_bf16: null pointer");
    int rc = ln_check("acr_layernorm_bwd_bf16", M, C);
    if (rc) return rc;
    const dim3 grid(ln_grid(M));
    hipStream_t st = (hipStream_t)stream;
    LN_DISPATCH(ln_bwd_kernel, bf16_t, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)gamma, stats, (const bf16_t*)dskip,
                (bf16_t*)dx, ws, M, C)
    hipLaunchKernelGGL(ln_param_reduce_kernel<bf16_t>, dim3(2 * C / 32), dim3(256), 0, st, (const float*)ws, ln_grid(M), C,
                       (bf16_t*)dgamma, (bf16_t*)dbeta);
    return acr_check_launch("acr_layernorm_bwd_bf16");
}

static bool ln_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }      // null (optional operand) passes

// ---- fp32 rows (reference precision): the same kernels on float tensors; vision_transformer.py:219-222 in the fp32 step ----
extern "C" int acr_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* stats, int32_t M,
                                     int32_t C, float eps, void* stream) {
    ACR_CHECK_ARG(x && gamma && beta && y && stats, "acr_layernorm_fwd_f32: null pointer");
    ACR_CHECK_ARG(ln_al16(x) && ln_al16(gamma) && ln_al16(beta) && ln_al16(y),
                  "acr_layernorm_fwd_f32: x / gamma / beta / y must be 16-byte aligned (vector loads and stores)");
    int rc = ln_check("acr_layernorm_fwd_f32", M, C);
    if (rc) return rc;
    const dim3 grid(ln_grid(M));
    hipStream_t st = (hipStream_t)stream;
    LN_DISPATCH(ln_fwd_kernel, float, x, gamma, beta, y, stats, M, C, eps)
    return acr_check_launch("acr_layernorm_fwd_f32");
}

extern "C" int acr_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* stats, const float* dskip,
                                     float* dx, float* ws, float* dgamma, float* dbeta, int32_t M, int32_t C, void* stream) {
    ACR_CHECK_ARG(dy && x && gamma && stats && dx && ws && dgamma && dbeta, "acr_layernorm_bwd_f32: null pointer");
    ACR_CHECK_ARG(ln_al16(dy) && ln_al16(x) && ln_al16(gamma) && ln_al16(dskip) && ln_al16(dx),
                  "acr_layernorm_bwd_f32: dy / x / gamma / dskip / dx must be 16-byte aligned (vector loads and stores)");
    int rc = ln_check("acr_layernorm_bwd_f32", M, C);
    if (rc) return rc;
    const dim3 grid(ln_grid(M));
    hipStream_t st = (hipStream_t)stream;
    LN_DISPATCH(ln_bwd_kernel, float, dy, x, gamma, stats, dskip, dx, ws, M, C)
    hipLaunchKernelGGL(ln_param_reduce_kernel<float>, dim3(2 * C / 32), dim3(256), 0, st, (const float*)ws, ln_grid(M), C, dgamma, dbeta);
    return acr_check_launch("acr_layernorm_bwd_f32");
}
