// Weight standardisation of every StdConv2dSame of the ResNetV2 stem in ONE launch each way
// (models/layers/std_conv.py:56-59: w_hat = (w - mean) / (std + eps) per output channel, biased std, recomputed
// every forward).  The stock path is ~10 tiny elementwise/reduction kernels per convolution and direction
// (~500 launches of ~5 us per step for the 52 convolutions); here one workgroup owns one output channel of one
// convolution, found through a small descriptor table, and does the two-pass statistics in fp32.
//   forward : p0 = w (cout, n) bf16, p1 = w_hat out, p3 (nullable, 1x1 convolutions) = w_hat^T (n, cout) out
//   backward: p0 = w, p1 = g = dL/dw_hat, p2 = dL/dw out:
//             dw = [g - mean(g) - w_hat * mean(g * w_hat) * (std + eps) / std] / (std + eps)
#include "acr_common.h"

typedef __bf16 bf16_t;

struct WStdDesc {
    uint64_t p0, p1, p2, p3;
    int32_t cout, n, ch_start, pad;
};

__device__ __forceinline__ float wstd_block_sum(float v, float* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

template <bool BWD, typename T>
__global__ __launch_bounds__(256) void wstd_kernel(const WStdDesc* __restrict__ desc, int n_conv, float eps) {
    __shared__ float sh[4];
    const int ch = blockIdx.x;
    int ci = 0;
    while (ci + 1 < n_conv && desc[ci + 1].ch_start <= ch) ++ci;
    const WStdDesc d = desc[ci];
    const int c = ch - d.ch_start, n = d.n, tid = threadIdx.x;
    const T* w = reinterpret_cast<const T*>(d.p0) + (int64_t)c * n;
    const float inv_n = 1.f / (float)n;
    float s = 0.f;
    for (int i = tid; i < n; i += 256) s += (float)w[i];
    const float mean = wstd_block_sum(s, sh) * inv_n;
    float ss = 0.f;
    for (int i = tid; i < n; i += 256) { const float t = (float)w[i] - mean; ss = fmaf(t, t, ss); }
    const float sd = sqrtf(wstd_block_sum(ss, sh) * inv_n);
    const float inv = 1.f / (sd + eps);
    if (!BWD) {
        T* out = reinterpret_cast<T*>(d.p1) + (int64_t)c * n;
        T* out_t = reinterpret_cast<T*>(d.p3);               // 1x1 convolutions: also the (cin, cout) copy the input-gradient
        for (int i = tid; i < n; i += 256) {                 // GEMM reads, written here instead of 33 transposes per step
            const T v = (T)(((float)w[i] - mean) * inv);
            out[i] = v;
            if (out_t) out_t[(int64_t)i * d.cout + c] = v;
        }
    } else {
        const T* g = reinterpret_cast<const T*>(d.p1) + (int64_t)c * n;
        T* dw = reinterpret_cast<T*>(d.p2) + (int64_t)c * n;
        float sg = 0.f, sgw = 0.f;
        for (int i = tid; i < n; i += 256) {
            const float gi = (float)g[i];
            sg += gi;
            sgw = fmaf(gi, ((float)w[i] - mean) * inv, sgw);
        }
        const float mg = wstd_block_sum(sg, sh) * inv_n;
        const float mgw = wstd_block_sum(sgw, sh) * inv_n;
        const float k = (sd > 0.f) ? mgw * (sd + eps) / sd : 0.f;
        for (int i = tid; i < n; i += 256)
            dw[i] = (T)(((float)g[i] - mg - ((float)w[i] - mean) * inv * k) * inv);
    }
}

extern "C" int acr_weight_std_bf16(const void* desc_dev, int32_t n_conv, int32_t total_channels, float eps,
                                   int32_t backward, void* stream) {
    ACR_CHECK_ARG(desc_dev && n_conv > 0 && total_channels > 0, "acr_weight_std_bf16: bad arguments");
    if (backward)
        hipLaunchKernelGGL((wstd_kernel<true, bf16_t>), dim3(total_channels), dim3(256), 0, (hipStream_t)stream,
                           (const WStdDesc*)desc_dev, n_conv, eps);
    else
        hipLaunchKernelGGL((wstd_kernel<false, bf16_t>), dim3(total_channels), dim3(256), 0, (hipStream_t)stream,
                           (const WStdDesc*)desc_dev, n_conv, eps);
    return acr_check_launch("acr_weight_std_bf16");
}

// the same for fp32 weights (reference precision; replaces ~800 tiny elementwise / reduction launches per fp32 step)
extern "C" int acr_weight_std_f32(const void* desc_dev, int32_t n_conv, int32_t total_channels, float eps, int32_t backward,
                                  void* stream) {
    ACR_CHECK_ARG(desc_dev && n_conv > 0 && total_channels > 0, "acr_weight_std_f32: bad arguments");
    if (backward)
        hipLaunchKernelGGL((wstd_kernel<true, float>), dim3(total_channels), dim3(256), 0, (hipStream_t)stream,
                           (const WStdDesc*)desc_dev, n_conv, eps);
    else
        hipLaunchKernelGGL((wstd_kernel<false, float>), dim3(total_channels), dim3(256), 0, (hipStream_t)stream,
                           (const WStdDesc*)desc_dev, n_conv, eps);
    return acr_check_launch("acr_weight_std_f32");
}
