// Fused GroupNorm(32) [+ residual add] [+ ReLU] of the ResNetV2 stem at the REFERENCE precision (fp32 NCHW), forward and
// backward (models/layers/norm_act.py:69-85, models/resnetv2.py:205-215).
//
// A (sample, group) of the fp32 stem is up to 401 KB (8 channels x 112^2 x 4 B): it does not fit the register file the way
// the bf16 kernel (groupnorm.hip) keeps it, so these kernels STREAM it twice -- the second pass re-reads what the first
// just pulled through L2 / Infinity Cache:
//   forward : pass 1 x -> mean, rstd (shifted sums: sum(x - x0), sum((x - x0)^2) with x0 = the group's first element, fp32,
//             no catastrophic cancellation); pass 2 x (+ resid) -> y.        2 reads (+1) + 1 write
//             (stock: row-moments + normalise + clamp kernels = 3 reads + 2 writes, plus the residual add's 2 reads + 1 write)
//   backward: pass 1 dy, x (+ resid for the ReLU mask) -> per-channel sum(dy), sum(dy * xhat) and the two group sums;
//             pass 2 dy, x (+ resid) -> dx (+ dresid).                        4 (6) reads + 1 (2) writes
//             (stock: threshold-backward, internal-gradients, elementwise backward, gamma/beta kernels, the add's backward)
// Deterministic: every sum is a fixed-order block reduction (no atomics); d(gamma), d(beta) per (sample, channel) are summed
// over samples in order by a second kernel.  HW must be a multiple of 4 (16-byte vectors inside a channel).
#include "acr_common.h"

#define GNF_GROUPS 32
// Cache policy of the second (last) pass: nontemporal loads and stores.  The activations are 100-400 MB per tensor -- nothing
// the next kernel could still find in the 4 MB L2 / 256 MB Infinity Cache -- and without the hint the pass evicts the lines
// its own first pass just pulled in for the neighbouring workgroups.  Measured in the fp32 bench (rocprofv3 sums per step):
// forward 3.37 -> 2.97 ms, backward 6.67 -> 6.02 ms.
#define GNF_LD2(p) __builtin_nontemporal_load(&(p))
#define GNF_ST(p, v) __builtin_nontemporal_store(v, &(p))
enum { GNF_NONE = 0, GNF_RELU = 1, GNF_ADD_RELU = 2 };

template <int NT>
__device__ __forceinline__ float gnf_block_sum(float v, float* sh) {
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
__global__ __launch_bounds__(NT) void gnf_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, float* __restrict__ stats, int C, int HW, int cg, float eps) {
    __shared__ float sh[NT / 64];
    const int g = blockIdx.x % GNF_GROUPS, n = blockIdx.x / GNF_GROUPS;
    const int64_t base = ((int64_t)n * C + (int64_t)g * cg) * HW;
    const int nvec = (cg * HW) >> 2, vpc = HW >> 2;
    const float inv_n = 1.f / (float)(cg * HW);
    const int tid = threadIdx.x;
    const f32x4* xv = reinterpret_cast<const f32x4*>(x + base);
    const float x0 = x[base];
    float s1 = 0.f, s2 = 0.f;
    for (int v = tid; v < nvec; v += NT) {
        const f32x4 a = xv[v];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = a[e] - x0; s1 += d; s2 = fmaf(d, d, s2); }
    }
    const float m1 = gnf_block_sum<NT>(s1, sh) * inv_n;
    const float m2 = gnf_block_sum<NT>(s2, sh) * inv_n;
    const float mean = x0 + m1;
    const float rstd = rsqrtf(fmaxf(m2 - m1 * m1, 0.f) + eps);
    const f32x4* rv = reinterpret_cast<const f32x4*>(res + (ACT == GNF_ADD_RELU ? base : 0));
    f32x4* yv = reinterpret_cast<f32x4*>(y + base);
    for (int v = tid; v < nvec; v += NT) {
        const int c = g * cg + v / vpc;
        const float ga = gamma[c] * rstd;
        const float be = beta[c] - mean * ga;
        const f32x4 a = GNF_LD2(xv[v]);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(a[e], ga, be);
        if (ACT == GNF_ADD_RELU) o += rv[v];
        if (ACT != GNF_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        GNF_ST(yv[v], o);
    }
    if (tid == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
}

template <int NT, int ACT>
__global__ __launch_bounds__(NT) void gnf_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ res, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ stats,
                                                     float* __restrict__ dx, float* __restrict__ dres,
                                                     float* __restrict__ dgamma_part, float* __restrict__ dbeta_part, int C, int HW, int cg) {
    __shared__ float sh[NT / 64];
    const int g = blockIdx.x % GNF_GROUPS, n = blockIdx.x / GNF_GROUPS;
    const int64_t base = ((int64_t)n * C + (int64_t)g * cg) * HW;
    const int vpc = HW >> 2;
    const float inv_n = 1.f / (float)(cg * HW);
    const int tid = threadIdx.x;
    const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
    // the forward's exact expression decides the ReLU mask, so it is the one forward applied
    auto masked = [&](const f32x4& gy, const f32x4& a, const f32x4& r, float ga, float be) {
        f32x4 o = gy;
        if (ACT != GNF_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pre = fmaf(a[e], ga, be);
                if (ACT == GNF_ADD_RELU) pre += r[e];
                if (!(pre > 0.f)) o[e] = 0.f;
            }
        }
        return o;
    };
    float s1 = 0.f, s2 = 0.f;                               // group sums of gamma * dy' and gamma * dy' * xhat
    for (int cl = 0; cl < cg; ++cl) {
        const int c = g * cg + cl;
        const float gam = gamma[c], ga = gam * rstd, be = beta[c] - mean * ga;
        const f32x4* xv = reinterpret_cast<const f32x4*>(x + base + (int64_t)cl * HW);
        const f32x4* gv = reinterpret_cast<const f32x4*>(dy + base + (int64_t)cl * HW);
        const f32x4* rv = reinterpret_cast<const f32x4*>(res + (ACT == GNF_ADD_RELU ? base + (int64_t)cl * HW : 0));
        float db = 0.f, dg = 0.f;
        for (int v = tid; v < vpc; v += NT) {
            const f32x4 a = xv[v];
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (ACT == GNF_ADD_RELU) r = rv[v];
            const f32x4 gy = masked(gv[v], a, r, ga, be);
#pragma unroll
            for (int e = 0; e < 4; ++e) { db += gy[e]; dg = fmaf(gy[e], (a[e] - mean) * rstd, dg); }
        }
        db = gnf_block_sum<NT>(db, sh);
        dg = gnf_block_sum<NT>(dg, sh);
        if (tid == 0) {
            dgamma_part[(int64_t)n * C + c] = dg;
            dbeta_part[(int64_t)n * C + c] = db;
        }
        s1 = fmaf(db, gam, s1);
        s2 = fmaf(dg, gam, s2);
    }
    const float c1 = s1 * inv_n, c2 = s2 * inv_n;
    for (int cl = 0; cl < cg; ++cl) {
        const int c = g * cg + cl;
        const float gam = gamma[c], ga = gam * rstd, be = beta[c] - mean * ga;
        const f32x4* xv = reinterpret_cast<const f32x4*>(x + base + (int64_t)cl * HW);
        const f32x4* gv = reinterpret_cast<const f32x4*>(dy + base + (int64_t)cl * HW);
        const f32x4* rv = reinterpret_cast<const f32x4*>(res + (ACT == GNF_ADD_RELU ? base + (int64_t)cl * HW : 0));
        f32x4* ov = reinterpret_cast<f32x4*>(dx + base + (int64_t)cl * HW);
        f32x4* dv = reinterpret_cast<f32x4*>(dres + (ACT == GNF_ADD_RELU ? base + (int64_t)cl * HW : 0));
        for (int v = tid; v < vpc; v += NT) {
            const f32x4 a = GNF_LD2(xv[v]);
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (ACT == GNF_ADD_RELU) r = GNF_LD2(rv[v]);
            const f32x4 gy = masked(GNF_LD2(gv[v]), a, r, ga, be);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rstd * (fmaf(gy[e], gam, -c1) - (a[e] - mean) * rstd * c2);
            GNF_ST(ov[v], o);
            if (ACT == GNF_ADD_RELU) GNF_ST(dv[v], gy);
        }
    }
}

// ---- small launches (CAM generation: two views of one image = 64 (sample, group) pairs on 256 CUs) ---------------------------------
// A (sample, group) is cut into P parts: gnf_part_kernel reduces each part to shifted sums (same shift x0 = the group's first
// element), gnf_apply_kernel combines the P partials in part order (deterministic) and normalises its own part.  Two launches
// of N * 32 * P workgroups instead of one of N * 32: 27 -> ~10 us for the 1024-channel maps of a 384^2 image.
__global__ __launch_bounds__(256) void gnf_part_kernel(const float* __restrict__ x, float* __restrict__ parts, int C, int HW, int cg, int P, int vper) {
    __shared__ float sh[4];
    const int part = blockIdx.x % P, ng = blockIdx.x / P;
    const int g = ng % GNF_GROUPS, n = ng / GNF_GROUPS;
    const int64_t base = ((int64_t)n * C + (int64_t)g * cg) * HW;
    const int nvec = (cg * HW) >> 2;
    const int v0 = part * vper, v1 = min(nvec, v0 + vper);
    const f32x4* xv = reinterpret_cast<const f32x4*>(x + base);
    const float x0 = x[base];
    float s1 = 0.f, s2 = 0.f;
    for (int v = v0 + threadIdx.x; v < v1; v += 256) {
        const f32x4 a = xv[v];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = a[e] - x0; s1 += d; s2 = fmaf(d, d, s2); }
    }
    s1 = gnf_block_sum<256>(s1, sh);
    s2 = gnf_block_sum<256>(s2, sh);
    if (threadIdx.x == 0) { parts[2 * blockIdx.x] = s1; parts[2 * blockIdx.x + 1] = s2; }
}
template <int ACT>
__global__ __launch_bounds__(256) void gnf_apply_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ parts, float* __restrict__ y,
                                                        float* __restrict__ stats, int C, int HW, int cg, int P, int vper, float eps) {
    const int part = blockIdx.x % P, ng = blockIdx.x / P;
    const int g = ng % GNF_GROUPS, n = ng / GNF_GROUPS;
    const int64_t base = ((int64_t)n * C + (int64_t)g * cg) * HW;
    const int nvec = (cg * HW) >> 2, vpc = HW >> 2;
    const float inv_n = 1.f / (float)(cg * HW);
    float s1 = 0.f, s2 = 0.f;
    for (int p = 0; p < P; ++p) { s1 += parts[2 * (ng * P + p)]; s2 += parts[2 * (ng * P + p) + 1]; }      // part order, every thread alike
    const float x0 = x[base];
    const float m1 = s1 * inv_n, m2 = s2 * inv_n;
    const float mean = x0 + m1;
    const float rstd = rsqrtf(fmaxf(m2 - m1 * m1, 0.f) + eps);
    const f32x4* xv = reinterpret_cast<const f32x4*>(x + base);
    const f32x4* rv = reinterpret_cast<const f32x4*>(res + (ACT == GNF_ADD_RELU ? base : 0));
    f32x4* yv = reinterpret_cast<f32x4*>(y + base);
    const int v0 = part * vper, v1 = min(nvec, v0 + vper);
    for (int v = v0 + threadIdx.x; v < v1; v += 256) {
        const int c = g * cg + v / vpc;
        const float ga = gamma[c] * rstd;
        const float be = beta[c] - mean * ga;
        const f32x4 a = xv[v];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(a[e], ga, be);
        if (ACT == GNF_ADD_RELU) o += rv[v];
        if (ACT != GNF_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        yv[v] = o;
    }
    if (part == 0 && threadIdx.x == 0) { stats[2 * ng] = mean; stats[2 * ng + 1] = rstd; }
}
static int gnf_fwd_parts(int N, int C, int HW) {
    const int groups = N * GNF_GROUPS, nvec = (C / GNF_GROUPS) * HW / 4;
    if (groups > 128 || nvec < 4096) return 1;
    int P = 512 / groups;
    if (P > 8) P = 8;
    while (P > 1 && nvec / P < 1024) --P;                    // at least 4 vectors per thread and part
    return P;
}
extern "C" size_t acr_groupnorm_fwd_ws_floats(int32_t N, int32_t C, int32_t HW) {
    if (N <= 0 || C <= 0 || HW <= 0 || (C % GNF_GROUPS) != 0) return 0;
    const int P = gnf_fwd_parts(N, C, HW);
    return P > 1 ? (size_t)N * GNF_GROUPS * P * 2 : 0;
}

// dgamma[c] = sum_n part[n][c] in sample order (8 interleaved chains combined in fixed order), fp32 out
__global__ __launch_bounds__(256) void gnf_param_reduce_kernel(const float* __restrict__ gpart, const float* __restrict__ bpart, int N,
                                                               int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
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
        if (i < C) dgamma[i] = t;
        else dbeta[i - C] = t;
    }
}

static int gnf_check(const char* who, int N, int C, int HW, int act) {
    ACR_CHECK_ARG(N > 0 && C > 0 && (C % GNF_GROUPS) == 0, "%s: C=%d must be a multiple of 32", who, C);
    ACR_CHECK_ARG(HW > 0 && (HW % 4) == 0, "%s: H*W=%d must be a multiple of 4 (16-byte vectors per channel)", who, HW);
    ACR_CHECK_ARG(act >= 0 && act <= 2, "%s: unknown act %d", who, act);
    return ACR_OK;
}

#define GNF_DISPATCH(KERNEL, ...)                                                                                \
    if (big) {                                                                                                    \
        if (act == 0) hipLaunchKernelGGL((KERNEL<1024, 0>), grid, dim3(1024), 0, st, __VA_ARGS__);               \
        else if (act == 1) hipLaunchKernelGGL((KERNEL<1024, 1>), grid, dim3(1024), 0, st, __VA_ARGS__);          \
        else hipLaunchKernelGGL((KERNEL<1024, 2>), grid, dim3(1024), 0, st, __VA_ARGS__);                        \
    } else {                                                                                                      \
        if (act == 0) hipLaunchKernelGGL((KERNEL<256, 0>), grid, dim3(256), 0, st, __VA_ARGS__);                 \
        else if (act == 1) hipLaunchKernelGGL((KERNEL<256, 1>), grid, dim3(256), 0, st, __VA_ARGS__);            \
        else hipLaunchKernelGGL((KERNEL<256, 2>), grid, dim3(256), 0, st, __VA_ARGS__);                          \
    }

extern "C" int acr_groupnorm_fwd_f32(const float* x, const float* resid, const float* gamma, const float* beta, float* y, float* stats,
                                     int32_t N, int32_t C, int32_t HW, float eps, int32_t act, float* ws, void* stream) {
    ACR_CHECK_ARG(x && gamma && beta && y && stats && (act != GNF_ADD_RELU || resid), "acr_groupnorm_fwd_f32: null pointer");
    int rc = gnf_check("acr_groupnorm_fwd_f32", N, C, HW, act);
    if (rc) return rc;
    ACR_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)resid & 15) == 0, "acr_groupnorm_fwd_f32: 16-byte alignment");
    const int cg = C / GNF_GROUPS;
    const bool big = (int64_t)cg * HW >= 32768;              // >= 8 vectors per thread at 1024 threads
    const dim3 grid(N * GNF_GROUPS);
    hipStream_t st = (hipStream_t)stream;
    const int P = ws ? gnf_fwd_parts(N, C, HW) : 1;
    if (P > 1) {                                            // small launch: every (sample, group) cut into P parts, two launches
        const int nvec = cg * HW / 4, vper = (nvec + P - 1) / P;
        const dim3 pgrid(N * GNF_GROUPS * P);
        hipLaunchKernelGGL(gnf_part_kernel, pgrid, dim3(256), 0, st, x, ws, C, HW, cg, P, vper);
        if (act == 0) hipLaunchKernelGGL((gnf_apply_kernel<0>), pgrid, dim3(256), 0, st, x, resid, gamma, beta, (const float*)ws, y, stats, C, HW, cg, P, vper, eps);
        else if (act == 1) hipLaunchKernelGGL((gnf_apply_kernel<1>), pgrid, dim3(256), 0, st, x, resid, gamma, beta, (const float*)ws, y, stats, C, HW, cg, P, vper, eps);
        else hipLaunchKernelGGL((gnf_apply_kernel<2>), pgrid, dim3(256), 0, st, x, resid, gamma, beta, (const float*)ws, y, stats, C, HW, cg, P, vper, eps);
        return acr_check_launch("acr_groupnorm_fwd_f32(parts)");
    }
    GNF_DISPATCH(gnf_fwd_kernel, x, resid, gamma, beta, y, stats, C, HW, cg, eps)
    return acr_check_launch("acr_groupnorm_fwd_f32");
}

extern "C" int acr_groupnorm_bwd_f32(const float* dy, const float* x, const float* resid, const float* gamma, const float* beta,
                                     const float* stats, float* dx, float* dresid, float* dgamma_part, float* dbeta_part,
                                     float* dgamma, float* dbeta, int32_t N, int32_t C, int32_t HW, int32_t act, void* stream) {
    ACR_CHECK_ARG(dy && x && gamma && beta && stats && dx && dgamma_part && dbeta_part && (act != GNF_ADD_RELU || (resid && dresid)),
                  "acr_groupnorm_bwd_f32: null pointer");
    int rc = gnf_check("acr_groupnorm_bwd_f32", N, C, HW, act);
    if (rc) return rc;
    ACR_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)resid & 15) == 0 &&
                      ((uintptr_t)dresid & 15) == 0, "acr_groupnorm_bwd_f32: 16-byte alignment");
    const int cg = C / GNF_GROUPS;
    const bool big = (int64_t)HW >= 4096;                    // per-channel loops: 1024 threads only when a channel feeds them
    const dim3 grid(N * GNF_GROUPS);
    hipStream_t st = (hipStream_t)stream;
    GNF_DISPATCH(gnf_bwd_kernel, dy, x, resid, gamma, beta, stats, dx, dresid, dgamma_part, dbeta_part, C, HW, cg)
    if (dgamma && dbeta)
        hipLaunchKernelGGL(gnf_param_reduce_kernel, dim3((2 * C + 31) / 32), dim3(256), 0, st, (const float*)dgamma_part,
                           (const float*)dbeta_part, N, C, dgamma, dbeta);
    return acr_check_launch("acr_groupnorm_bwd_f32");
}
