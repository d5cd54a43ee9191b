// K4/K5 -- CAM read-outs of infer_cam.py: patch-token class activation (DPT/ACR.py:133-134),
// torch-semantics bilinear resize with label mask / h-flip / accumulate fused (infer_cam.py:156-162,
// 186-196, 201, 208), and the affinity refinement patch_aff @ cam (infer_cam.py:164-165, 183-184).
// All small and HBM/L2-bound; one wave per output row, lanes along the contiguous axis.
#include "acr_common.h"

// out[n][c] = relu(x[n,:] . w[c,:] + bias[c]); one wave per patch n, x row kept in registers.
template <typename T>
__global__ __launch_bounds__(256) void patch_cam_kernel(const T* __restrict__ x, int64_t x_st, const T* __restrict__ w,
                                                        const T* __restrict__ bias, int N, int D, int C,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const T* xr = x + (int64_t)n * x_st;
    for (int c = 0; c < C; ++c) {
        const T* wr = w + (int64_t)c * D;
        float acc = 0.f;
        for (int d = lane; d < D; d += 64) acc = fmaf(acr_load1<T>(xr + d), acr_load1<T>(wr + d), acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) out[(int64_t)n * C + c] = fmaxf(acc + acr_load1<T>(bias + c), 0.f);
    }
}

// torch upsample_bilinear2d source index (aten/src/ATen/native/UpSample.h area_pixel_compute_source_index)
__device__ __forceinline__ float src_index(float scale, int dst, bool align_corners) {
    if (align_corners) return scale * (float)dst;
    const float s = scale * ((float)dst + 0.5f) - 0.5f;
    return s < 0.f ? 0.f : s;
}

__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ src, int64_t src_sc, int64_t src_sp,
                                                       int C, int ih, int iw, float* __restrict__ dst, int oh, int ow,
                                                       int align_corners, float sh, float sw,
                                                       const float* __restrict__ chan_mul, int hflip, int accumulate) {
    const int64_t total = (int64_t)C * oh * ow;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int x = (int)(idx % ow);
        const int y = (int)((idx / ow) % oh);
        const int c = (int)(idx / ((int64_t)ow * oh));
        const float fy = src_index(sh, y, align_corners), fx = src_index(sw, x, align_corners);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < ih - 1 ? 1 : 0), x1 = x0 + (x0 < iw - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* s = src + (int64_t)c * src_sc;
        const float v00 = s[((int64_t)y0 * iw + x0) * src_sp], v01 = s[((int64_t)y0 * iw + x1) * src_sp];
        const float v10 = s[((int64_t)y1 * iw + x0) * src_sp], v11 = s[((int64_t)y1 * iw + x1) * src_sp];
        float val = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
        if (chan_mul) val *= chan_mul[c];
        const int xo = hflip ? (ow - 1 - x) : x;
        float* d = dst + ((int64_t)c * oh + y) * ow + xo;
        *d = accumulate ? (*d + val) : val;
    }
}

// out[s][c][r] = sum_k (sum_l a[s][l][1+r][1+k]) * cam[s][c][k] for sample s = blockIdx.y; one wave per row r, up to 8 cams per
// pass.  HBM-bound: every element of the (L, T, T) stack is read once.  The layers are summed FIRST, in layer order, into one
// value per k (the reference sums the 12 maps before the product too, infer_cam.py:164-165), with 4 x 4 independent loads in
// flight per lane (the first version's single dependent load per iteration ran at 1.07 TB/s).
__global__ __launch_bounds__(256) void aff_refine_kernel(const float* __restrict__ a, int64_t a_sb, int L, int T,
                                                         const float* __restrict__ cam, int n_cam, int c0,
                                                         float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int N = T - 1;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;
    a += (int64_t)blockIdx.y * a_sb;
    cam += (int64_t)blockIdx.y * n_cam * N;
    out += (int64_t)blockIdx.y * n_cam * N;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    const int nc = min(8, n_cam - c0);
    const float* row0 = a + (int64_t)(1 + r) * T + 1;
    const int64_t lst = (int64_t)T * T;
    for (int k0 = 0; k0 < N; k0 += 256) {
        float av[4] = {0.f, 0.f, 0.f, 0.f};
        int kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) kk[u] = min(k0 + 64 * u + lane, N - 1);          // clamped: loads stay unconditional
        int l = 0;
        for (; l + 4 <= L; l += 4) {
            float t[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) t[j][u] = __builtin_nontemporal_load(row0 + (l + j) * lst + kk[u]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) av[u] += t[j][u];
        }
        for (; l < L; ++l)
#pragma unroll
            for (int u = 0; u < 4; ++u) av[u] += __builtin_nontemporal_load(row0 + l * lst + kk[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 + 64 * u + lane < N) {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c < nc) acc[c] = fmaf(av[u], cam[(int64_t)(c0 + c) * N + kk[u]], acc[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off);
        if (lane == 0 && c < nc) out[(int64_t)(c0 + c) * N + r] = acc[c];
    }
}

extern "C" int acr_patch_cam(const void* x, int64_t x_st, const void* w, const void* bias, int32_t N, int32_t D,
                             int32_t C, int32_t dtype, float* out, void* stream) {
    ACR_CHECK_ARG(x && w && bias && out, "acr_patch_cam: null pointer");
    ACR_CHECK_ARG(N > 0 && D > 0 && C > 0 && x_st >= D, "acr_patch_cam: bad geometry N=%d D=%d C=%d", N, D, C);
    const dim3 grid((N + 3) / 4);
    if (dtype == ACR_F32)
        hipLaunchKernelGGL((patch_cam_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, x_st,
                           (const float*)w, (const float*)bias, N, D, C, out);
    else if (dtype == ACR_BF16)
        hipLaunchKernelGGL((patch_cam_kernel<__bf16>), grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, x_st,
                           (const __bf16*)w, (const __bf16*)bias, N, D, C, out);
    else {
        acr_set_error("acr_patch_cam: unknown dtype %d", dtype);
        return ACR_ERR_UNSUPPORTED;
    }
    return acr_check_launch("acr_patch_cam");
}

extern "C" int acr_bilinear_resize(const float* src, int64_t src_sc, int64_t src_sp, int32_t C, int32_t ih,
                                   int32_t iw, float* dst, int32_t oh, int32_t ow, int32_t align_corners,
                                   const float* chan_mul, int32_t hflip, int32_t accumulate, void* stream) {
    ACR_CHECK_ARG(src && dst, "acr_bilinear_resize: null pointer");
    ACR_CHECK_ARG(C > 0 && ih > 0 && iw > 0 && oh > 0 && ow > 0, "acr_bilinear_resize: bad geometry");
    // torch: align_corners -> (in-1)/(out-1) (0 when out == 1); else in/out
    float sh, sw;
    if (align_corners) {
        sh = oh > 1 ? (float)(ih - 1) / (float)(oh - 1) : 0.f;
        sw = ow > 1 ? (float)(iw - 1) / (float)(ow - 1) : 0.f;
    } else {
        sh = (float)ih / (float)oh;
        sw = (float)iw / (float)ow;
    }
    const int64_t total = (int64_t)C * oh * ow;
    int64_t nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(bilinear_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, src, src_sc, src_sp, C, ih, iw,
                       dst, oh, ow, align_corners, sh, sw, chan_mul, hflip, accumulate);
    return acr_check_launch("acr_bilinear_resize");
}

extern "C" int acr_aff_refine_batch(const float* a, int64_t a_sb, int32_t L, int32_t T, const float* cam, int32_t n_cam,
                                    int32_t nbatch, float* out, void* stream) {
    ACR_CHECK_ARG(a && cam && out, "acr_aff_refine: null pointer");
    ACR_CHECK_ARG(L > 0 && T > 1 && n_cam > 0 && nbatch > 0 && nbatch < 65536, "acr_aff_refine: bad geometry L=%d T=%d n=%d batch=%d", L, T, n_cam, nbatch);
    ACR_CHECK_ARG(nbatch == 1 || a_sb >= (int64_t)L * T * T, "acr_aff_refine: batch stride smaller than one (L, T, T) stack");
    const dim3 grid((T - 1 + 3) / 4, nbatch);
    for (int c0 = 0; c0 < n_cam; c0 += 8)
        hipLaunchKernelGGL(aff_refine_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, a_sb, L, T, cam, n_cam, c0, out);
    return acr_check_launch("acr_aff_refine");
}

extern "C" int acr_aff_refine(const float* a, int32_t L, int32_t T, const float* cam, int32_t n_cam, float* out,
                              void* stream) {
    return acr_aff_refine_batch(a, 0, L, T, cam, n_cam, 1, out, stream);
}
