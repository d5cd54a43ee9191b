// 3x3 stride-2 max-pool of the ResNetV2 stem with the "SAME" padding folded in (models/resnetv2.py:322-328 'same' stem:
// MaxPool2dSame(kernel_size=3, stride=2) = -inf pad (0,1,0,1) + max_pool2d).  HBM-bound byte work: one pass over the
// input, a 1-byte argmax per output for the backward (the stock path keeps int64 indices and a padded copy), and a
// gather backward (each input pixel looks at the <= 4 windows that contain it) -- no atomics, deterministic.
#include "acr_common.h"

typedef __bf16 bf16_t;

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                          uint8_t* __restrict__ amax, int64_t total, int H, int W, int Ho,
                                                          int Wo, int pt, int pl) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int wo = (int)(o % Wo);
    const int64_t t = o / Wo;
    const int ho = (int)(t % Ho);
    const int64_t nc = t / Ho;
    const bf16_t* xp = x + nc * H * W;
    const int h0 = 2 * ho - pt, w0 = 2 * wo - pl;
    float best = -INFINITY;
    int arg = 0;
    bool have = false;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int h = h0 + i, w = w0 + j;
            if (h >= 0 && h < H && w >= 0 && w < W) {
                const float v = (float)xp[(int64_t)h * W + w];
                if (!have || v > best || v != v) { best = v; arg = i * 3 + j; have = true; }   // first max, NaN wins (ATen)
            }
        }
    y[o] = (bf16_t)best;
    amax[o] = (uint8_t)arg;
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16_t* __restrict__ dy, const uint8_t* __restrict__ amax,
                                                          bf16_t* __restrict__ dx, int64_t total, int H, int W, int Ho,
                                                          int Wo, int pt, int pl) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const int w = (int)(p % W);
    const int64_t t = p / W;
    const int h = (int)(t % H);
    const int64_t nc = t / H;
    const bf16_t* dyp = dy + nc * Ho * Wo;
    const uint8_t* ap = amax + nc * Ho * Wo;
    // windows (ho, wo) with 2 ho - pt <= h <= 2 ho - pt + 2
    const int ho_lo = max((h + pt - 1) >> 1, 0), ho_hi = min((h + pt) >> 1, Ho - 1);
    const int wo_lo = max((w + pl - 1) >> 1, 0), wo_hi = min((w + pl) >> 1, Wo - 1);
    float g = 0.f;
    for (int ho = ho_lo; ho <= ho_hi; ++ho)
        for (int wo = wo_lo; wo <= wo_hi; ++wo) {
            const int i = h - (2 * ho - pt), j = w - (2 * wo - pl);
            if (ap[ho * Wo + wo] == i * 3 + j) g += (float)dyp[ho * Wo + wo];
        }
    dx[p] = (bf16_t)g;
}

// 8 consecutive pixels of one input row per thread (one 16-byte store): the windows touching them are <= 6 columns x 2
// rows of outputs, whose argmax bytes and gradients are read once into registers.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8p;
__global__ __launch_bounds__(256) void maxpool_bwd_vec_kernel(const bf16_t* __restrict__ dy, const uint8_t* __restrict__ amax,
                                                              bf16_t* __restrict__ dx, int64_t total8, int H, int W, int Ho,
                                                              int Wo, int pt, int pl) {
    const int64_t t8 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t8 >= total8) return;
    const int W8 = W >> 3;
    const int w0 = (int)(t8 % W8) * 8;
    const int64_t t = t8 / W8;
    const int h = (int)(t % H);
    const int64_t nc = t / H;
    const bf16_t* dyp = dy + nc * Ho * Wo;
    const uint8_t* ap = amax + nc * Ho * Wo;
    const int ho_lo = max((h + pt - 1) >> 1, 0), ho_hi = min((h + pt) >> 1, Ho - 1);
    const int wo_first = max((w0 + pl - 1) >> 1, 0), wo_last = min((w0 + 7 + pl) >> 1, Wo - 1);   // <= 6 columns
    float g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
        const int i = h - (2 * ho - pt);                     // row of the pixel inside window ho
        for (int wo = wo_first; wo <= wo_last; ++wo) {
            const int a = ap[ho * Wo + wo];
            if (a / 3 == i) {
                const int e = (2 * wo - pl) + (a - 3 * i) - w0;      // which of my 8 pixels this window's maximum is
                if (e >= 0 && e < 8) {
                    const float v = (float)dyp[ho * Wo + wo];
#pragma unroll
                    for (int k = 0; k < 8; ++k) g[k] += (k == e) ? v : 0.f;
                }
            }
        }
    }
    bf16x8p o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (bf16_t)g[k];
    *reinterpret_cast<bf16x8p*>(dx + (nc * H + h) * W + w0) = o;
}

extern "C" int acr_maxpool3x3s2_fwd_bf16(const void* x, void* y, uint8_t* amax, int64_t nc, int32_t h, int32_t w, int32_t ho,
                                         int32_t wo, int32_t pad_top, int32_t pad_left, void* stream) {
    ACR_CHECK_ARG(x && y && amax, "acr_maxpool3x3s2_fwd_bf16: null pointer");
    ACR_CHECK_ARG(nc > 0 && h > 0 && w > 0 && ho > 0 && wo > 0 && pad_top >= 0 && pad_left >= 0 && pad_top < 3 && pad_left < 3 &&
                      2 * (ho - 1) - pad_top < h && 2 * (wo - 1) - pad_left < w,
                  "acr_maxpool3x3s2_fwd_bf16: every window must contain at least one input pixel");
    const int64_t total = nc * ho * wo;
    ACR_CHECK_ARG((total + 255) / 256 < (1ll << 31), "acr_maxpool3x3s2_fwd_bf16: too large");
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (bf16_t*)y, amax, total, h, w, ho, wo, pad_top, pad_left);
    return acr_check_launch("acr_maxpool3x3s2_fwd_bf16");
}

extern "C" int acr_maxpool3x3s2_bwd_bf16(const void* dy, const uint8_t* amax, void* dx, int64_t nc, int32_t h, int32_t w,
                                         int32_t ho, int32_t wo, int32_t pad_top, int32_t pad_left, void* stream) {
    ACR_CHECK_ARG(dy && dx && amax, "acr_maxpool3x3s2_bwd_bf16: null pointer");
    ACR_CHECK_ARG(nc > 0 && h > 0 && w > 0 && ho > 0 && wo > 0 && pad_top >= 0 && pad_left >= 0 && pad_top < 3 && pad_left < 3,
                  "acr_maxpool3x3s2_bwd_bf16: bad geometry");
    const int64_t total = nc * h * w;
    ACR_CHECK_ARG((total + 255) / 256 < (1ll << 31), "acr_maxpool3x3s2_bwd_bf16: too large");
    if ((w % 8) == 0 && ((uintptr_t)dx & 15) == 0) {
        const int64_t total8 = total / 8;
        hipLaunchKernelGGL(maxpool_bwd_vec_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dy, amax, (bf16_t*)dx, total8, h, w, ho, wo, pad_top, pad_left);
        return acr_check_launch("acr_maxpool3x3s2_bwd_bf16");
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dy, amax, (bf16_t*)dx, total, h, w, ho, wo, pad_top, pad_left);
    return acr_check_launch("acr_maxpool3x3s2_bwd_bf16");
}
