// Dense-CRF refinement of CAMs: the permutohedral-lattice Gaussian filter and the mean-field update of
// tool/imutils.py:345-362 (pydensecrf: densecrf v2, Kraehenbuehl & Koltun 2011), called by infer_cam.py:27-40,218-225.
// The lattice follows the reference tree's own copy of that code, wrapper/bilateralfilter/permutohedral.cpp:112-283 (init)
// and :441-520 (splat / blur / slice), operation by operation in fp32 -- but is built the GPU way:
//
//   CPU reference                                   here
//   ---------------------------------------------   ------------------------------------------------------------------
//   hash table, ids in insertion order              every pixel emits its d+1 vertex keys packed into 64 bits; ONE stable
//                                                   radix sort of the (key, pixel-vertex) pairs + a scan of the run heads
//                                                   numbers the lattice points (sorted-key order: deterministic) -- no hash
//   blur neighbours by hash look-up                 binary search in the sorted unique keys
//   splat: values[o] += w * in, pixel after pixel   the sort is stable, so the run of a lattice point lists its contributions
//                                                   in exactly the (pixel, vertex) order of the CPU loop: one thread sums its
//                                                   run sequentially in fp32 -> the SAME rounding sequence, bit for bit, and
//                                                   no float atomics (run-to-run deterministic)
//   blur / slice                                    element-wise kernels, same expression order
//
// The filter is HBM / latency bound integer-and-gather work (no GEMM in it): ~(d+1) N K multiply-adds per splat and slice and
// (d+1) M K per blur, against tables of (d+1) N entries.  Lattice point numbering differs from the hash table's (sorted vs
// first-come), which no result depends on.
#include <hipcub/hipcub.hpp>

#include "acr_common.h"

namespace {

constexpr int KEY_BITS = 12;                 // per coordinate, biased by 2048: |key| < 2048 checked in the kernel
constexpr int KEY_BIAS = 1 << (KEY_BITS - 1);

struct LatticeHeader {                       // first 64 bytes of the workspace (device memory)
    int32_t n_points;                        // M
    int32_t key_overflow;                    // != 0: a key left the packable range (the build is then invalid)
    int32_t n_pixels, d;
    int32_t n_long;                          // lattice points whose splat run is longer than LONG_RUN (listed in long_list)
    int32_t pad[11];
};
constexpr int LONG_RUN = 96;                 // runs above this are summed by a whole wave (lattice_splat_long_kernel)
constexpr int LONG_WAVES = 2048;             // waves of that kernel per label plane (they stride over the list)

struct LatticeLayout {                       // byte offsets into the workspace, a pure function of (n_pixels, d)
    int64_t keys, keys_sorted, ids, ids_sorted, flags, weights, offsets, sorted_w, sorted_pix, seg_start, ukeys, nbr, long_list, temp, total;
    int64_t temp_bytes;
};

inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// The reference's SSE init pads the pixel list to a multiple of 4 with ZERO-feature lanes and inserts their vertex keys too
// (permutohedral.cpp:161-163,232-240): when n % 4 != 0 the lattice owns the simplex around the origin even if no pixel touches
// it; those points get no splat but take part in the blur.  One phantom pixel (weight 0) reproduces that.
inline int lattice_padded(int n) { return n + ((n & 3) ? 1 : 0); }

int lattice_layout(int32_t n, int32_t d, LatticeLayout* L) {
    const int64_t P = (int64_t)lattice_padded(n) * (d + 1);
    size_t sort_bytes = 0, scan_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                           (uint32_t*)nullptr, (int)P, 0, KEY_BITS * d, (hipStream_t)0) != hipSuccess)
        return ACR_ERR_LAUNCH;
    if (hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (int)P, (hipStream_t)0) != hipSuccess)
        return ACR_ERR_LAUNCH;
    int64_t o = align256(sizeof(LatticeHeader));
    auto take = [&](int64_t bytes) { const int64_t at = o; o = align256(o + bytes); return at; };
    L->keys = take(P * 8);
    L->keys_sorted = take(P * 8);
    L->ids = take(P * 4);
    L->ids_sorted = take(P * 4);
    L->flags = take(P * 4);                  // run-head flags, scanned in place into 1-based point numbers
    L->weights = take(P * 4);
    L->offsets = take(P * 4);
    L->sorted_w = take(P * 4);
    L->sorted_pix = take(P * 4);
    L->seg_start = take((P + 1) * 4);
    L->ukeys = take(P * 8);
    L->nbr = take(P * (d + 1) * 8);          // (d+1, M, 2) int32, worst case M = P
    L->long_list = take((P / LONG_RUN + 1) * 4);
    L->temp_bytes = (int64_t)(sort_bytes > scan_bytes ? sort_bytes : scan_bytes);
    L->temp = take(L->temp_bytes);
    L->total = o;
    return ACR_OK;
}

template <typename T>
__host__ __device__ inline T* at(void* ws, int64_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(ws) + off); }
template <typename T>
__host__ __device__ inline const T* at(const void* ws, int64_t off) { return reinterpret_cast<const T*>(reinterpret_cast<const char*>(ws) + off); }

// ---- per pixel: the enclosing simplex, its d+1 vertex keys and barycentric weights (permutohedral.cpp:158-243) --------------
template <int D>
__global__ __launch_bounds__(256) void lattice_pairs_kernel(const uint8_t* __restrict__ rgb, int H, int W, float sxy, float srgb,
                                                            uint64_t* __restrict__ keys, uint32_t* __restrict__ ids,
                                                            float* __restrict__ weights, LatticeHeader* hdr, int n_padded) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_padded) return;
    const bool phantom = p >= H * W;                              // zero features, weight 0 (see lattice_padded)
    float f[D];
    f[0] = phantom ? 0.f : (float)(p % W) / sxy;
    f[1] = phantom ? 0.f : (float)(p / W) / sxy;
    if constexpr (D == 5) {
#pragma unroll
        for (int c = 0; c < 3; ++c) f[2 + c] = phantom ? 0.f : (float)rgb[(int64_t)p * 3 + c] / srgb;
    }
    // scale factors: 1 / sqrt((i+2)(i+1)) * inv_std_dev evaluated in double, then rounded to float (:146-149)
    const float inv_std_dev = (float)(sqrt(2.0 / 3.0) * (D + 1));
    float elevated[D + 1], rem0[D + 1], rank[D + 1], bary[D + 2];
    float sm = 0.f;
#pragma unroll
    for (int j = D; j > 0; --j) {
        const float scale = (float)(1.0 / sqrt((double)((j + 1) * j)) * (double)inv_std_dev);
        const float cf = __fmul_rn(f[j - 1], scale);
        elevated[j] = __fsub_rn(sm, __fmul_rn((float)j, cf));
        sm = __fadd_rn(sm, cf);
    }
    elevated[0] = sm;
    const float invdplus1 = 1.0f / (float)(D + 1), dplus1 = (float)(D + 1);
    float total = 0.f;
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        const float v = rintf(__fmul_rn(invdplus1, elevated[i]));      // round half to even, as cvtps_epi32 under ROUND_NEAREST
        rem0[i] = __fmul_rn(v, dplus1);
        total = __fadd_rn(total, v);
        rank[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float di = __fsub_rn(elevated[i], rem0[i]);
#pragma unroll
        for (int j = i + 1; j <= D; ++j) {
            const float dj = __fsub_rn(elevated[j], rem0[j]);
            const float c = di < dj ? 1.f : 0.f;
            rank[i] += c;
            rank[j] += 1.f - c;
        }
    }
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        rank[i] += total;
        const float add = rank[i] < 0.f ? dplus1 : 0.f;
        const float sub = rank[i] >= dplus1 ? dplus1 : 0.f;
        rank[i] += add - sub;
        rem0[i] += add - sub;
    }
#pragma unroll
    for (int i = 0; i <= D + 1; ++i) bary[i] = 0.f;
#pragma unroll
    for (int i = 0; i <= D; ++i) {
        const float v = __fmul_rn(__fsub_rn(elevated[i], rem0[i]), invdplus1);
        const int q = D - (int)rank[i];
        // bary[q] += v; bary[q + 1] -= v with q a runtime index: unrolled selects keep the array in registers
#pragma unroll
        for (int s = 0; s <= D + 1; ++s) {
            if (s == q) bary[s] = __fadd_rn(bary[s], v);
            if (s == q + 1) bary[s] = __fsub_rn(bary[s], v);
        }
    }
    bary[0] = __fadd_rn(bary[0], __fadd_rn(1.0f, bary[D + 1]));
    bool overflow = false;
#pragma unroll
    for (int r = 0; r <= D; ++r) {
        uint64_t packed = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const int rk = (int)rank[i];
            const int canonical = rk <= D - r ? r : r - (D + 1);        // canonical[r][rank] (:137-142)
            const int key = (int)rem0[i] + canonical;
            overflow |= (key < -KEY_BIAS + 1) || (key > KEY_BIAS - 2);  // one spare step each way for the neighbour keys
            packed = (packed << KEY_BITS) | (uint64_t)((key + KEY_BIAS) & ((1 << KEY_BITS) - 1));
        }
        const int64_t e = (int64_t)p * (D + 1) + r;
        keys[e] = packed;
        ids[e] = (uint32_t)e;
        weights[e] = phantom ? 0.f : bary[r];
    }
    if (overflow) atomicOr(&hdr->key_overflow, 1);
}

__global__ __launch_bounds__(256) void lattice_heads_kernel(const uint64_t* __restrict__ ks, int32_t* __restrict__ flags, int P) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < P) flags[i] = (i == 0 || ks[i] != ks[i - 1]) ? 1 : 0;
}

// after the scan flags[i] = 1-based lattice point of sorted pair i
__global__ __launch_bounds__(256) void lattice_scatter_kernel(const uint64_t* __restrict__ ks, const uint32_t* __restrict__ ids_sorted,
                                                              const int32_t* __restrict__ point1, const float* __restrict__ weights,
                                                              int32_t* __restrict__ offsets, float* __restrict__ sorted_w,
                                                              int32_t* __restrict__ sorted_pix, int32_t* __restrict__ seg_start,
                                                              uint64_t* __restrict__ ukeys, LatticeHeader* hdr, int P, int dplus1,
                                                              int n_pixels) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int pt = point1[i] - 1;
    const uint32_t e = ids_sorted[i];
    offsets[e] = pt;
    sorted_w[i] = weights[e];
    const int32_t px = (int32_t)(e / (uint32_t)dplus1);
    sorted_pix[i] = px < n_pixels ? px : 0;                       // the phantom pixel: weight 0, any readable pixel
    if (i == 0 || ks[i] != ks[i - 1]) {
        seg_start[pt] = i;
        ukeys[pt] = ks[i];
    }
    if (i == P - 1) {
        seg_start[pt + 1] = P;
        hdr->n_points = pt + 1;
    }
}

__device__ __forceinline__ int lattice_find(const uint64_t* __restrict__ ukeys, int M, uint64_t key) {
    int lo = 0, hi = M - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const uint64_t k = ukeys[mid];
        if (k == key) return mid;
        if (k < key) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

// blur neighbours along axis j (permutohedral.cpp:268-281): n1 = key - 1 (all coordinates), n1[j] = key[j] + d; n2 mirrored.
// Axis j == d touches the implied (d+1)-th coordinate only, so n1 = key - 1, n2 = key + 1 there.
template <int D>
__global__ __launch_bounds__(256) void lattice_neighbors_kernel(const uint64_t* __restrict__ ukeys, LatticeHeader* hdr,
                                                                int32_t* __restrict__ nbr, int P, const int32_t* __restrict__ seg_start,
                                                                int32_t* __restrict__ long_list) {
    const int M = hdr->n_points;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= M) return;
    if (j == 0 && seg_start[i + 1] - seg_start[i] > LONG_RUN) long_list[atomicAdd(&hdr->n_long, 1)] = i;   // order irrelevant
    const uint64_t key = ukeys[i];
    uint64_t k1 = 0, k2 = 0;
    bool ok1 = true, ok2 = true;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const int v = (int)((key >> (KEY_BITS * (D - 1 - c))) & ((1 << KEY_BITS) - 1)) - KEY_BIAS;
        const int a = c == j ? v + D : v - 1;
        const int b = c == j ? v - D : v + 1;
        ok1 &= a >= -KEY_BIAS && a < KEY_BIAS;
        ok2 &= b >= -KEY_BIAS && b < KEY_BIAS;
        k1 = (k1 << KEY_BITS) | (uint64_t)((a + KEY_BIAS) & ((1 << KEY_BITS) - 1));
        k2 = (k2 << KEY_BITS) | (uint64_t)((b + KEY_BIAS) & ((1 << KEY_BITS) - 1));
    }
    int32_t* out = nbr + ((int64_t)j * P + i) * 2;               // stride P (worst case), only the first M entries are live
    out[0] = ok1 ? lattice_find(ukeys, M, k1) : -1;
    out[1] = ok2 ? lattice_find(ukeys, M, k2) : -1;
}

// ---- filter ---------------------------------------------------------------------------------------------------------------
// values[k][pt + 1] = sum over the run of pt, in (pixel, vertex) order, of w * (in[k][pixel] * pre[pixel])
__global__ __launch_bounds__(256) void lattice_splat_kernel(const LatticeHeader* hdr, const int32_t* __restrict__ seg_start,
                                                            const float* __restrict__ sorted_w, const int32_t* __restrict__ sorted_pix,
                                                            const float* __restrict__ in, const float* __restrict__ pre,
                                                            float* __restrict__ vals, int n_pixels, int64_t vstride) {
    const int M = hdr->n_points;
    const int pt = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (pt > M) return;                                           // slot 0 (= "no neighbour") and slot M + 1 are zero
    float* v = vals + (int64_t)k * vstride;
    if (pt == M) { v[0] = 0.f; v[M + 1] = 0.f; return; }
    const float* x = in + (int64_t)k * n_pixels;
    const int s = seg_start[pt], e = seg_start[pt + 1];
    if (e - s > LONG_RUN) return;                                 // lattice_splat_long_kernel's
    float acc = 0.f;
    for (int i = s; i < e; ++i) {
        const int px = sorted_pix[i];
        float xv = x[px];
        if (pre) xv = __fmul_rn(xv, pre[px]);
        acc = __fadd_rn(acc, __fmul_rn(sorted_w[i], xv));
    }
    v[pt + 1] = acc;
}

// Long runs (a flat image region puts 10^4 pixels on one bilateral lattice point): one thread walking such a run pays a
// dependent gather per element (~80 cycles each, 1.2 ms per launch at VOC size).  Here a wave takes the run 64 entries at a
// time: the lanes fetch weight, pixel and value and form the 64 products in parallel (each product is rounded exactly as in the
// serial loop), then the wave adds them IN ORDER -- acc += readlane(product, i), i = 0..63 -- so the rounding sequence, and
// with it every bit of the result, stays that of the CPU loop.
__global__ __launch_bounds__(256) void lattice_splat_long_kernel(const LatticeHeader* hdr, const int32_t* __restrict__ long_list,
                                                                 const int32_t* __restrict__ seg_start, const float* __restrict__ sorted_w,
                                                                 const int32_t* __restrict__ sorted_pix, const float* __restrict__ in,
                                                                 const float* __restrict__ pre, float* __restrict__ vals, int n_pixels,
                                                                 int64_t vstride) {
    const int n_long = hdr->n_long;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float* x = in + (int64_t)blockIdx.y * n_pixels;
    float* v = vals + (int64_t)blockIdx.y * vstride;
    for (int r = wave; r < n_long; r += gridDim.x * 4) {
        const int pt = long_list[r];
        const int s = seg_start[pt], e = seg_start[pt + 1];
        float acc = 0.f;
        for (int c = s; c < e; c += 64) {
            const int i = c + lane;
            float prod = 0.f;
            if (i < e) {
                const int px = sorted_pix[i];
                float xv = x[px];
                if (pre) xv = __fmul_rn(xv, pre[px]);
                prod = __fmul_rn(sorted_w[i], xv);
            }
            const int cnt = min(64, e - c);                      // wave-uniform
            if (cnt == 64) {
#pragma unroll
                for (int t = 0; t < 64; ++t) acc = __fadd_rn(acc, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prod), t)));
            } else {
                for (int t = 0; t < cnt; ++t) acc = __fadd_rn(acc, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prod), t)));
            }
        }
        if (lane == 0) v[pt + 1] = acc;
    }
}

__global__ __launch_bounds__(256) void lattice_blur_kernel(const LatticeHeader* hdr, const int32_t* __restrict__ nbr,
                                                           const float* __restrict__ src, float* __restrict__ dst, int64_t vstride) {
    const int M = hdr->n_points;
    const int pt = blockIdx.x * 256 + threadIdx.x;
    if (pt > M) return;
    const float* s = src + (int64_t)blockIdx.y * vstride;
    float* d = dst + (int64_t)blockIdx.y * vstride;
    if (pt == M) { d[0] = 0.f; d[M + 1] = 0.f; return; }
    const int n1 = nbr[2 * pt] + 1, n2 = nbr[2 * pt + 1] + 1;
    d[pt + 1] = __fadd_rn(s[pt + 1], __fmul_rn(0.5f, __fadd_rn(s[n1], s[n2])));
}

// out[k][p] = (sum_j (w_j * alpha) * values[k][offset_j + 1]) (* post[p]) (* post_scale)
template <int D>
__global__ __launch_bounds__(256) void lattice_slice_kernel(const int32_t* __restrict__ offsets, const float* __restrict__ weights,
                                                            const float* __restrict__ vals, const float* __restrict__ post,
                                                            float post_scale, int apply_scale, float* __restrict__ out, int n_pixels,
                                                            int64_t vstride) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pixels) return;
    const float* v = vals + (int64_t)blockIdx.y * vstride;
    const float alpha = 1.0f / (1.f + exp2f(-(float)D));          // exact: 1 / (1 + 2^-d)
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j <= D; ++j) {
        const float w = __fmul_rn(weights[(int64_t)p * (D + 1) + j], alpha);
        acc = __fadd_rn(acc, __fmul_rn(w, v[offsets[(int64_t)p * (D + 1) + j] + 1]));
    }
    if (post) acc = __fmul_rn(acc, post[p]);
    if (apply_scale) acc = __fmul_rn(post_scale, acc);
    out[(int64_t)blockIdx.y * n_pixels + p] = acc;
}

// ---- mean field -------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void crf_unary_kernel(const float* __restrict__ probs, float* __restrict__ unary, int64_t n, float clip) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) unary[i] = -logf(fminf(fmaxf(probs[i], clip), 1.0f));
}

// norm[p] = 1 / sqrt(filter(1)[p] + 1e-20)   (DenseKernel::initLattice, NORMALIZE_SYMMETRIC)
__global__ __launch_bounds__(256) void crf_norm_kernel(float* __restrict__ norm, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) norm[i] = 1.0f / sqrtf(norm[i] + 1e-20f);
}

// q[:, p] = softmax_k(-unary[:, p] + msg0[:, p] + msg1[:, p])   (DenseCRF::inference + expAndNormalize)
__global__ __launch_bounds__(256) void crf_update_kernel(const float* __restrict__ unary, const float* __restrict__ msg0,
                                                         const float* __restrict__ msg1, float* __restrict__ q, int n, int K) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) {
        float t = -unary[(int64_t)k * n + p];
        if (msg0) t = __fadd_rn(t, msg0[(int64_t)k * n + p]);
        if (msg1) t = __fadd_rn(t, msg1[(int64_t)k * n + p]);
        q[(int64_t)k * n + p] = t;
        mx = fmaxf(mx, t);
    }
    float sum = 0.f;
    for (int k = 0; k < K; ++k) {
        const float e = expf(q[(int64_t)k * n + p] - mx);
        q[(int64_t)k * n + p] = e;
        sum += e;
    }
    for (int k = 0; k < K; ++k) q[(int64_t)k * n + p] = q[(int64_t)k * n + p] / sum;
}

}  // namespace

extern "C" int64_t acr_lattice_ws_bytes(int32_t n_pixels, int32_t d) {
    if (n_pixels <= 0 || (d != 2 && d != 5) || ((int64_t)n_pixels + 1) * (d + 1) >= (1ll << 31)) {
        acr_set_error("acr_lattice_ws_bytes: n_pixels = %d, d = %d (d must be 2 or 5)", n_pixels, d);
        return -1;
    }
    LatticeLayout L;
    if (lattice_layout(n_pixels, d, &L) != ACR_OK) {
        acr_set_error("acr_lattice_ws_bytes: hipcub temporary-storage query failed (no HIP device?)");
        return -1;
    }
    return L.total;
}

extern "C" int acr_lattice_build(const void* rgb, int32_t H, int32_t W, float sxy, float srgb, void* ws, int64_t ws_bytes, void* stream) {
    const int d = rgb ? 5 : 2;
    ACR_CHECK_ARG(ws && H > 0 && W > 0 && sxy > 0.f && (!rgb || srgb > 0.f), "acr_lattice_build: bad arguments");
    ACR_CHECK_ARG(((int64_t)H * W + 1) * (d + 1) < (1ll << 31), "acr_lattice_build: image too large (%d x %d)", H, W);
    const int n = H * W, n_pad = lattice_padded(n), P = n_pad * (d + 1);
    LatticeLayout L;
    if (lattice_layout(n, d, &L) != ACR_OK) { acr_set_error("acr_lattice_build: hipcub temporary-storage query failed"); return ACR_ERR_LAUNCH; }
    ACR_CHECK_ARG(ws_bytes >= L.total, "acr_lattice_build: workspace of %lld bytes, %lld needed", (long long)ws_bytes, (long long)L.total);
    hipStream_t st = (hipStream_t)stream;
    LatticeHeader* hdr = at<LatticeHeader>(ws, 0);
    LatticeHeader h0 = {};
    h0.n_pixels = n;
    h0.d = d;
    if (hipMemcpyAsync(hdr, &h0, sizeof(h0), hipMemcpyHostToDevice, st) != hipSuccess) return acr_check_launch("acr_lattice_build(header)");
    const dim3 gp((n_pad + 255) / 256), gP((P + 255) / 256);
    if (d == 5)
        hipLaunchKernelGGL(lattice_pairs_kernel<5>, gp, dim3(256), 0, st, (const uint8_t*)rgb, H, W, sxy, srgb, at<uint64_t>(ws, L.keys),
                           at<uint32_t>(ws, L.ids), at<float>(ws, L.weights), hdr, n_pad);
    else
        hipLaunchKernelGGL(lattice_pairs_kernel<2>, gp, dim3(256), 0, st, (const uint8_t*)nullptr, H, W, sxy, 1.f, at<uint64_t>(ws, L.keys),
                           at<uint32_t>(ws, L.ids), at<float>(ws, L.weights), hdr, n_pad);
    size_t tb = (size_t)L.temp_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(at<void>(ws, L.temp), tb, at<const uint64_t>(ws, L.keys), at<uint64_t>(ws, L.keys_sorted),
                                           at<const uint32_t>(ws, L.ids), at<uint32_t>(ws, L.ids_sorted), P, 0, KEY_BITS * d, st) != hipSuccess)
        return acr_check_launch("acr_lattice_build(sort)");
    hipLaunchKernelGGL(lattice_heads_kernel, gP, dim3(256), 0, st, at<const uint64_t>(ws, L.keys_sorted), at<int32_t>(ws, L.flags), P);
    tb = (size_t)L.temp_bytes;
    if (hipcub::DeviceScan::InclusiveSum(at<void>(ws, L.temp), tb, at<const int32_t>(ws, L.flags), at<int32_t>(ws, L.flags), P, st) != hipSuccess)
        return acr_check_launch("acr_lattice_build(scan)");
    hipLaunchKernelGGL(lattice_scatter_kernel, gP, dim3(256), 0, st, at<const uint64_t>(ws, L.keys_sorted), at<const uint32_t>(ws, L.ids_sorted),
                       at<const int32_t>(ws, L.flags), at<const float>(ws, L.weights), at<int32_t>(ws, L.offsets), at<float>(ws, L.sorted_w),
                       at<int32_t>(ws, L.sorted_pix), at<int32_t>(ws, L.seg_start), at<uint64_t>(ws, L.ukeys), hdr, P, d + 1, n);
    const dim3 gn((P + 255) / 256, d + 1);                      // M is only known on the device: cover the worst case, exit early
    if (d == 5)
        hipLaunchKernelGGL(lattice_neighbors_kernel<5>, gn, dim3(256), 0, st, at<const uint64_t>(ws, L.ukeys), hdr, at<int32_t>(ws, L.nbr), P,
                           at<const int32_t>(ws, L.seg_start), at<int32_t>(ws, L.long_list));
    else
        hipLaunchKernelGGL(lattice_neighbors_kernel<2>, gn, dim3(256), 0, st, at<const uint64_t>(ws, L.ukeys), hdr, at<int32_t>(ws, L.nbr), P,
                           at<const int32_t>(ws, L.seg_start), at<int32_t>(ws, L.long_list));
    return acr_check_launch("acr_lattice_build");
}

// Synchronises the stream: the caller sizes the value scratch of acr_lattice_filter from n_points.
extern "C" int acr_lattice_info(const void* ws, int32_t* n_points, int32_t* key_overflow, void* stream) {
    ACR_CHECK_ARG(ws && n_points && key_overflow, "acr_lattice_info: null argument");
    LatticeHeader h;
    if (hipMemcpyAsync(&h, ws, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        return acr_check_launch("acr_lattice_info");
    *n_points = h.n_points;
    *key_overflow = h.key_overflow;
    if (h.key_overflow) {
        acr_set_error("acr_lattice_info: a lattice key left the +-%d range the 64-bit packing covers (features too large)", KEY_BIAS - 2);
        return ACR_ERR_INVALID;
    }
    return ACR_OK;
}

// Device pointers of the per-pixel tables inside the workspace (tests compare them with the reference's tables).
extern "C" int acr_lattice_tables(void* ws, int32_t n_pixels, int32_t d, void** offsets, void** weights, void** point_keys) {
    ACR_CHECK_ARG(ws && offsets && weights && point_keys && (d == 2 || d == 5), "acr_lattice_tables: bad arguments");
    LatticeLayout L;
    if (lattice_layout(n_pixels, d, &L) != ACR_OK) { acr_set_error("acr_lattice_tables: layout query failed"); return ACR_ERR_LAUNCH; }
    *offsets = at<void>(ws, L.offsets);
    *weights = at<void>(ws, L.weights);
    *point_keys = at<void>(ws, L.ukeys);
    return ACR_OK;
}

extern "C" int acr_lattice_filter(const void* ws, int32_t n_pixels, int32_t d, int32_t n_points, const void* in, const void* pre, void* out,
                                  const void* post, float post_scale, int32_t apply_scale, int32_t K, void* vals, void* stream) {
    ACR_CHECK_ARG(ws && in && out && vals && K > 0 && n_points > 0 && (d == 2 || d == 5), "acr_lattice_filter: bad arguments");
    LatticeLayout L;
    if (lattice_layout(n_pixels, d, &L) != ACR_OK) { acr_set_error("acr_lattice_filter: layout query failed"); return ACR_ERR_LAUNCH; }
    hipStream_t st = (hipStream_t)stream;
    const LatticeHeader* hdr = at<LatticeHeader>(ws, 0);
    const int64_t vstride = (int64_t)n_points + 2;
    const int64_t P = (int64_t)lattice_padded(n_pixels) * (d + 1);
    float* va = (float*)vals;
    float* vb = va + (int64_t)K * vstride;
    const dim3 gm((n_points + 1 + 255) / 256, K);
    hipLaunchKernelGGL(lattice_splat_kernel, gm, dim3(256), 0, st, hdr, at<const int32_t>(ws, L.seg_start), at<const float>(ws, L.sorted_w),
                       at<const int32_t>(ws, L.sorted_pix), (const float*)in, (const float*)pre, va, n_pixels, vstride);
    hipLaunchKernelGGL(lattice_splat_long_kernel, dim3(LONG_WAVES / 4, K), dim3(256), 0, st, hdr, at<const int32_t>(ws, L.long_list),
                       at<const int32_t>(ws, L.seg_start), at<const float>(ws, L.sorted_w), at<const int32_t>(ws, L.sorted_pix),
                       (const float*)in, (const float*)pre, va, n_pixels, vstride);
    for (int j = 0; j <= d; ++j) {
        hipLaunchKernelGGL(lattice_blur_kernel, gm, dim3(256), 0, st, hdr, at<const int32_t>(ws, L.nbr) + (int64_t)j * P * 2, (const float*)va,
                           vb, vstride);
        float* t = va; va = vb; vb = t;
    }
    const dim3 gs((n_pixels + 255) / 256, K);
    if (d == 5)
        hipLaunchKernelGGL(lattice_slice_kernel<5>, gs, dim3(256), 0, st, at<const int32_t>(ws, L.offsets), at<const float>(ws, L.weights),
                           (const float*)va, (const float*)post, post_scale, apply_scale, (float*)out, n_pixels, vstride);
    else
        hipLaunchKernelGGL(lattice_slice_kernel<2>, gs, dim3(256), 0, st, at<const int32_t>(ws, L.offsets), at<const float>(ws, L.weights),
                           (const float*)va, (const float*)post, post_scale, apply_scale, (float*)out, n_pixels, vstride);
    return acr_check_launch("acr_lattice_filter");
}

extern "C" int acr_crf_unary(const void* probs, void* unary, int64_t n, float clip, void* stream) {
    ACR_CHECK_ARG(probs && unary && n > 0, "acr_crf_unary: bad arguments");
    hipLaunchKernelGGL(crf_unary_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)probs, (float*)unary, n, clip);
    return acr_check_launch("acr_crf_unary");
}

extern "C" int acr_crf_norm(void* norm, int32_t n_pixels, void* stream) {
    ACR_CHECK_ARG(norm && n_pixels > 0, "acr_crf_norm: bad arguments");
    hipLaunchKernelGGL(crf_norm_kernel, dim3((n_pixels + 255) / 256), dim3(256), 0, (hipStream_t)stream, (float*)norm, n_pixels);
    return acr_check_launch("acr_crf_norm");
}

extern "C" int acr_crf_update(const void* unary, const void* msg0, const void* msg1, void* q, int32_t n_pixels, int32_t K, void* stream) {
    ACR_CHECK_ARG(unary && q && n_pixels > 0 && K > 0, "acr_crf_update: bad arguments");
    hipLaunchKernelGGL(crf_update_kernel, dim3((n_pixels + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)unary, (const float*)msg0,
                       (const float*)msg1, (float*)q, n_pixels, K);
    return acr_check_launch("acr_crf_update");
}
