// 3x3 stride-2 max-pool of the ResNetV2 stem with the "SAME" padding folded in (models/resnetv2.py:322-328 'same' stem:
// MaxPool2dSame(kernel_size=3, stride=2) = -inf pad (0,1,0,1) + max_pool2d).  HBM-bound byte work: one pass over the
// input, a 1-byte argmax per output for the backward (the stock path keeps int64 indices and a padded copy), and a
// gather backward (each input pixel looks at the <= 4 windows that contain it) -- no atomics, deterministic.
#include "acr_common.h"

typedef __bf16 bf16_t;

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ amax, int64_t total, int H, int W, int Ho,
                                                          int Wo, int pt, int pl) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int wo = (int)(o % Wo);
    const int64_t t = o / Wo;
    const int ho = (int)(t % Ho);
    const int64_t nc = t / Ho;
    const T* xp = x + nc * H * W;
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
    y[o] = (T)best;
    amax[o] = (uint8_t)arg;
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ amax,
                                                          T* __restrict__ dx, int64_t total, int H, int W, int Ho,
                                                          int Wo, int pt, int pl) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const int w = (int)(p % W);
    const int64_t t = p / W;
    const int h = (int)(t % H);
    const int64_t nc = t / H;
    const T* dyp = dy + nc * Ho * Wo;
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
    dx[p] = (T)g;
}

// 8 consecutive pixels of one input row per thread (16-byte stores): the windows touching them are <= 6 columns x 2
// rows of outputs, whose argmax bytes and gradients are read once into registers.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8p;
__device__ __forceinline__ void store8(bf16_t* p, const float* g) {
    bf16x8p o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (bf16_t)g[k];
    *reinterpret_cast<bf16x8p*>(p) = o;
}
__device__ __forceinline__ void store8(float* p, const float* g) {
    *reinterpret_cast<f32x4*>(p) = f32x4{g[0], g[1], g[2], g[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{g[4], g[5], g[6], g[7]};
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_vec_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ amax,
                                                              T* __restrict__ dx, int64_t total8, int H, int W, int Ho,
                                                              int Wo, int pt, int pl) {
    const int64_t t8 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t8 >= total8) return;
    const int W8 = W >> 3;
    const int w0 = (int)(t8 % W8) * 8;
    const int64_t t = t8 / W8;
    const int h = (int)(t % H);
    const int64_t nc = t / H;
    const T* dyp = dy + nc * Ho * Wo;
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
    store8(dx + (nc * H + h) * W + w0, g);
}

// fp32 maps with no left / top padding (the stem's SAME pool of an even-sized map: pad (0,1,0,1)) and W % 8 == 0: FOUR outputs of a row
// per thread -- the 9 input columns they cover are two 16-byte loads + one float per input row, the outputs one 16-byte store and
// their argmax bytes one 4-byte store (one output per thread with nine 4-byte loads ran at 2.2 TB/s).  Windows are walked in the
// scalar kernel's (i, j) order with its comparison, so maxima and argmax bytes are the same.
__global__ __launch_bounds__(256) void maxpool_fwd_vec4_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ amax,
                                                               int64_t total4, int H, int W, int Ho, int Wo) {
    const int64_t t4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t4 >= total4) return;
    const int Wq = Wo >> 2;
    const int wq = (int)(t4 % Wq);
    const int64_t t = t4 / Wq;
    const int ho = (int)(t % Ho);
    const int64_t nc = t / Ho;
    const float* xp = x + nc * H * W + 8 * wq;
    const bool last = 8 * wq + 8 < W;                        // the ninth column exists
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int arg[4] = {0, 0, 0, 0};
    bool have[4] = {false, false, false, false};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int h = 2 * ho + i;
        if (h < H) {
            const float* rp = xp + (int64_t)h * W;
            const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp)), b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp + 4));
            const float v[9] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], last ? rp[8] : 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (2 * k + j < 8 || last) {
                        const float u = v[2 * k + j];
                        if (!have[k] || u > best[k] || u != u) { best[k] = u; arg[k] = i * 3 + j; have[k] = true; }
                    }
                }
        }
    }
    const int64_t o = (nc * Ho + ho) * Wo + 4 * wq;
    *reinterpret_cast<f32x4*>(y + o) = f32x4{best[0], best[1], best[2], best[3]};
    *reinterpret_cast<uint32_t*>(amax + o) = (uint32_t)arg[0] | ((uint32_t)arg[1] << 8) | ((uint32_t)arg[2] << 16) | ((uint32_t)arg[3] << 24);
}
// Backward of the same geometry (H even): a thread owns input rows 2r, 2r+1 x 8 columns.  The windows that can put their maximum
// there are rows r-1 (its third row) and r (its first two) x columns 4q-1 .. 4q+3: per window row one 4-byte + one 1-byte argmax load
// and one 16-byte + one 4-byte gradient load, every contribution added in the scalar kernel's (ho, wo) order -- the same sums bit
// for bit, with each window row read by two threads instead of by every pixel it touches.
__global__ __launch_bounds__(256) void maxpool_bwd_vec2x8_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ amax, float* __restrict__ dx,
                                                                 int64_t total, int H, int W, int Ho, int Wo) {
    const int64_t tix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (tix >= total) return;
    const int W8 = W >> 3;
    const int q = (int)(tix % W8);
    const int64_t t = tix / W8;
    const int r = (int)(t % (H >> 1));
    const int64_t nc = t / (H >> 1);
    const float* dyp = dy + nc * Ho * Wo;
    const uint8_t* ap = amax + nc * Ho * Wo;
    float g[2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int k = 0; k < 8; ++k) g[a][k] = 0.f;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int ho = r - 1 + d;
        if (ho < 0 || ho >= Ho) continue;
        const int64_t ro = (int64_t)ho * Wo + 4 * q;
        const uint32_t a4 = *reinterpret_cast<const uint32_t*>(ap + ro);
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(dyp + ro);
        int am[5];
        float vv[5];
        am[0] = q > 0 ? (int)ap[ro - 1] : 255;               // window column 4q-1 (none left of the map)
        vv[0] = q > 0 ? dyp[ro - 1] : 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) { am[c + 1] = (int)((a4 >> (8 * c)) & 255u); vv[c + 1] = v4[c]; }
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            if (am[c] == 255) continue;
            const int i = am[c] / 3, j = am[c] - 3 * i;
            const int row = 2 * ho + i - 2 * r;              // 0, 1: mine
            const int e = 2 * (4 * q - 1 + c) + j - 8 * q;   // column inside my 8
            if (row >= 0 && row < 2 && e >= 0 && e < 8) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int k = 0; k < 8; ++k) g[a][k] += (a == row && k == e) ? vv[c] : 0.f;
            }
        }
    }
    float* o = dx + (nc * H + 2 * r) * W + 8 * q;
    store8(o, g[0]);
    store8(o + W, g[1]);
}

template <typename T>
static int maxpool_fwd(const char* what, const void* x, void* y, uint8_t* amax, int64_t nc, int32_t h, int32_t w, int32_t ho, int32_t wo,
                       int32_t pad_top, int32_t pad_left, void* stream) {
    ACR_CHECK_ARG(x && y && amax, "%s: null pointer", what);
    ACR_CHECK_ARG(nc > 0 && h > 0 && w > 0 && ho > 0 && wo > 0 && pad_top >= 0 && pad_left >= 0 && pad_top < 3 && pad_left < 3 &&
                      2 * (ho - 1) - pad_top < h && 2 * (wo - 1) - pad_left < w,
                  "%s: every window must contain at least one input pixel", what);
    const int64_t total = nc * ho * wo;
    ACR_CHECK_ARG((total + 255) / 256 < (1ll << 31), "%s: too large", what);
    if (sizeof(T) == 4 && pad_top == 0 && pad_left == 0 && (w % 8) == 0 && wo * 2 == w && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
        ((uintptr_t)amax & 3) == 0) {
        const int64_t total4 = total / 4;
        hipLaunchKernelGGL(maxpool_fwd_vec4_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y,
                           amax, total4, h, w, ho, wo);
        return acr_check_launch(what);
    }
    hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y,
                       amax, total, h, w, ho, wo, pad_top, pad_left);
    return acr_check_launch(what);
}

template <typename T>
static int maxpool_bwd(const char* what, const void* dy, const uint8_t* amax, void* dx, int64_t nc, int32_t h, int32_t w, int32_t ho,
                       int32_t wo, int32_t pad_top, int32_t pad_left, void* stream) {
    ACR_CHECK_ARG(dy && dx && amax, "%s: null pointer", what);
    ACR_CHECK_ARG(nc > 0 && h > 0 && w > 0 && ho > 0 && wo > 0 && pad_top >= 0 && pad_left >= 0 && pad_top < 3 && pad_left < 3,
                  "%s: bad geometry", what);
    const int64_t total = nc * h * w;
    ACR_CHECK_ARG((total + 255) / 256 < (1ll << 31), "%s: too large", what);
    if (sizeof(T) == 4 && pad_top == 0 && pad_left == 0 && (w % 8) == 0 && (h % 2) == 0 && wo * 2 == w && ho * 2 == h && ((uintptr_t)dx & 15) == 0 &&
        ((uintptr_t)dy & 15) == 0 && ((uintptr_t)amax & 3) == 0) {
        const int64_t nthr = total / 16;
        hipLaunchKernelGGL(maxpool_bwd_vec2x8_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, amax,
                           (float*)dx, nthr, h, w, ho, wo);
        return acr_check_launch(what);
    }
    if ((w % 8) == 0 && ((uintptr_t)dx & 15) == 0) {
        const int64_t total8 = total / 8;
        hipLaunchKernelGGL(maxpool_bwd_vec_kernel<T>, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const T*)dy,
                           amax, (T*)dx, total8, h, w, ho, wo, pad_top, pad_left);
        return acr_check_launch(what);
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, amax,
                       (T*)dx, total, h, w, ho, wo, pad_top, pad_left);
    return acr_check_launch(what);
}

extern "C" int acr_maxpool3x3s2_fwd_bf16(const void* x, void* y, uint8_t* amax, int64_t nc, int32_t h, int32_t w, int32_t ho,
                                         int32_t wo, int32_t pad_top, int32_t pad_left, void* stream) {
    return maxpool_fwd<bf16_t>("acr_maxpool3x3s2_fwd_bf16", x, y, amax, nc, h, w, ho, wo, pad_top, pad_left, stream);
}
extern "C" int acr_maxpool3x3s2_bwd_bf16(const void* dy, const uint8_t* amax, void* dx, int64_t nc, int32_t h, int32_t w,
                                         int32_t ho, int32_t wo, int32_t pad_top, int32_t pad_left, void* stream) {
    return maxpool_bwd<bf16_t>("acr_maxpool3x3s2_bwd_bf16", dy, amax, dx, nc, h, w, ho, wo, pad_top, pad_left, stream);
}
// the same for fp32 maps (reference precision; the stock backward scatters with atomics: 0.8 ms per step vs 0.2 ms here)
extern "C" int acr_maxpool3x3s2_fwd_f32(const void* x, void* y, uint8_t* amax, int64_t nc, int32_t h, int32_t w, int32_t ho,
                                        int32_t wo, int32_t pad_top, int32_t pad_left, void* stream) {
    return maxpool_fwd<float>("acr_maxpool3x3s2_fwd_f32", x, y, amax, nc, h, w, ho, wo, pad_top, pad_left, stream);
}
extern "C" int acr_maxpool3x3s2_bwd_f32(const void* dy, const uint8_t* amax, void* dx, int64_t nc, int32_t h, int32_t w,
                                        int32_t ho, int32_t wo, int32_t pad_top, int32_t pad_left, void* stream) {
    return maxpool_bwd<float>("acr_maxpool3x3s2_bwd_f32", dy, amax, dx, nc, h, w, ho, wo, pad_top, pad_left, stream);
}

// ---- y[nc][i][j] = x[nc][2i][2j]: what a stride-2 1x1 convolution reads (the shortcuts of stages 1 and 2, models/resnetv2.py:232-249
// DownsampleConv; SAME padding is empty for a 1x1 kernel), and its backward dx = dy scattered onto the even pixels, zero elsewhere --
// one pass each way (autograd's nested slice backward made two zero fills and two strided copies of it).  fp32, Ho = ceil(H/2).
template <bool VEC> __global__ __launch_bounds__(256) void subsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t nthr, int H, int W,
                                                                                  int Ho, int Wo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nthr) return;
    const int wq = (Wo + 3) >> 2;
    const int q = (int)(i % wq);
    int64_t t = i / wq;
    const int r = (int)(t % Ho);
    const int64_t nc = t / Ho;
    const float* src = x + (nc * H + 2 * r) * W + 8 * q;
    float* dst = y + (nc * Ho + r) * Wo + 4 * q;
    if (VEC) {                                               // W % 8 == 0
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
        *reinterpret_cast<f32x4*>(dst) = f32x4{a[0], a[2], b[0], b[2]};
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * q + e < Wo) dst[e] = src[2 * e];
    }
}
template <bool VEC> __global__ __launch_bounds__(256) void subsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t nthr, int H, int W,
                                                                                  int Ho, int Wo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nthr) return;
    const int wq = (W + 7) >> 3;
    const int q = (int)(i % wq);
    int64_t t = i / wq;
    const int r = (int)(t % H);
    const int64_t nc = t / H;
    float* dst = dx + (nc * H + r) * W + 8 * q;
    const float* src = dy + (nc * Ho + (r >> 1)) * Wo + 4 * q;
    const bool even = (r & 1) == 0;
    if (VEC) {                                               // W % 8 == 0
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (even) v = *reinterpret_cast<const f32x4*>(src);
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], 0.f, v[1], 0.f};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[2], 0.f, v[3], 0.f};
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (8 * q + e < W) dst[e] = (even && (e & 1) == 0) ? src[e >> 1] : 0.f;
    }
}
extern "C" int acr_subsample2_fwd_f32(const float* x, float* y, int64_t nc, int32_t h, int32_t w, void* stream) {
    ACR_CHECK_ARG(x && y && nc > 0 && h > 0 && w > 0, "acr_subsample2_fwd_f32: bad argument");
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    const int64_t nthr = nc * ho * ((wo + 3) / 4);
    ACR_CHECK_ARG((nthr + 255) / 256 < (1ll << 31), "acr_subsample2_fwd_f32: too large");
    const bool vec = (w % 8) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0;
    if (vec) hipLaunchKernelGGL(subsample2_fwd_kernel<true>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, nthr, h, w, ho, wo);
    else hipLaunchKernelGGL(subsample2_fwd_kernel<false>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, nthr, h, w, ho, wo);
    return acr_check_launch("acr_subsample2_fwd_f32");
}
extern "C" int acr_subsample2_bwd_f32(const float* dy, float* dx, int64_t nc, int32_t h, int32_t w, void* stream) {
    ACR_CHECK_ARG(dy && dx && nc > 0 && h > 0 && w > 0, "acr_subsample2_bwd_f32: bad argument");
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    const int64_t nthr = nc * h * ((w + 7) / 8);
    ACR_CHECK_ARG((nthr + 255) / 256 < (1ll << 31), "acr_subsample2_bwd_f32: too large");
    const bool vec = (w % 8) == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)dy & 15) == 0;
    if (vec) hipLaunchKernelGGL(subsample2_bwd_kernel<true>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, nthr, h, w, ho, wo);
    else hipLaunchKernelGGL(subsample2_bwd_kernel<false>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, nthr, h, w, ho, wo);
    return acr_check_launch("acr_subsample2_bwd_f32");
}
