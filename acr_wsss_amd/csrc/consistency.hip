// K3 -- attention-consistency regulariser (train_acr.py:143-161) forward and backward.
//
// The reference un-flips the view-2 head-mean maps with 3*p in-place slice assignments (84 launches
// at p = 28, each backward node cloning the full (B,L,T,T) gradient) and then takes two L1 means.
// Here the flip is the index permutation pi(i*p+j) = i*p + (p-1-j) folded into the address of the
// second operand, so forward is one streaming pass over the two stacks and backward one pass that
// writes both gradients.  HBM-bound: forward reads 2*B*L*T*T*4 bytes, backward reads the same and
// writes the same.  One wave per attention row; lanes walk the row, so view-1 loads are
// fully coalesced and view-2 loads are coalesced p-element segments read right-to-left.
// Deterministic: per-workgroup partials + a fixed-order second stage in double.
#include "acr_common.h"

#define CONS_WAVES 4
#define CONS_THREADS (CONS_WAVES * 64)
#define CONS_MAX_BLOCKS 4096

static inline int cons_blocks(int64_t rows) {
    int64_t nb = (rows + CONS_WAVES - 1) / CONS_WAVES;
    return (int)(nb < CONS_MAX_BLOCKS ? nb : CONS_MAX_BLOCKS);
}

extern "C" size_t acr_consistency_ws_floats(int32_t B, int32_t L, int32_t T) {
    return (size_t)2 * (size_t)cons_blocks((int64_t)B * L * T);
}

// pi applied to a patch index c in [0, p*p): exact for p*p < 2^22 via float reciprocal.
__device__ __forceinline__ int flip_index(int c, int p, float inv_p) {
    int i = (int)(((float)c + 0.5f) * inv_p);
    int j = c - i * p;
    return i * p + (p - 1 - j);
}

__global__ __launch_bounds__(CONS_THREADS) void cons_fwd_kernel(
        const float* __restrict__ a1, const float* __restrict__ a2, int64_t a_sb,
        int L, int T, int p, int64_t rows, float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = T - 1;
    const float inv_p = 1.0f / (float)p;
    const int64_t TT = (int64_t)T * T;
    float cls = 0.f, aff = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * CONS_WAVES + wave; row < rows;
         row += (int64_t)gridDim.x * CONS_WAVES) {
        const int i = (int)(row % T);
        const int64_t bl = row / T;
        const int l = (int)(bl % L);
        const int64_t b = bl / L;
        const int64_t base = b * a_sb + (int64_t)l * TT;
        const int i2 = (i == 0) ? 0 : 1 + flip_index(i - 1, p, inv_p);
        const float* r1 = a1 + base + (int64_t)i * T + 1;
        const float* r2 = a2 + base + (int64_t)i2 * T + 1;
        float acc = 0.f;
        for (int c = lane; c < N; c += 64) acc += fabsf(r1[c] - r2[flip_index(c, p, inv_p)]);
        if (i == 0) cls += acc; else aff += acc;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cls += __shfl_xor(cls, off);
        aff += __shfl_xor(aff, off);
    }
    __shared__ float sh[2 * CONS_WAVES];
    if (lane == 0) { sh[2 * wave] = cls; sh[2 * wave + 1] = aff; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float c = 0.f, a = 0.f;
        for (int w = 0; w < CONS_WAVES; ++w) { c += sh[2 * w]; a += sh[2 * w + 1]; }
        partial[2 * blockIdx.x] = c;
        partial[2 * blockIdx.x + 1] = a;
    }
}

__global__ __launch_bounds__(256) void cons_reduce_kernel(const float* __restrict__ partial, int nblocks,
                                                         double inv_cls, double inv_aff,
                                                         float* __restrict__ out2) {
    __shared__ double sh[2 * 256];
    double c = 0.0, a = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { c += partial[2 * i]; a += partial[2 * i + 1]; }
    sh[threadIdx.x] = c;
    sh[256 + threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh[threadIdx.x] += sh[threadIdx.x + s];
            sh[256 + threadIdx.x] += sh[256 + threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out2[0] = (float)(sh[0] * inv_cls);
        out2[1] = (float)(sh[256] * inv_aff);
    }
}

__global__ __launch_bounds__(CONS_THREADS) void cons_bwd_kernel(
        const float* __restrict__ a1, const float* __restrict__ a2, int64_t a_sb,
        int L, int T, int p, int64_t rows, const float* __restrict__ gout2,
        float w_cls, float w_aff, float* __restrict__ g1, float* __restrict__ g2, int64_t g_sb, int64_t g_st) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = T - 1;
    const float inv_p = 1.0f / (float)p;
    const int64_t TT = (int64_t)T * T;
    const float gc = gout2[0] * w_cls, ga = gout2[1] * w_aff;
    for (int64_t row = (int64_t)blockIdx.x * CONS_WAVES + wave; row < rows;
         row += (int64_t)gridDim.x * CONS_WAVES) {
        const int i = (int)(row % T);
        const int64_t bl = row / T;
        const int l = (int)(bl % L);
        const int64_t b = bl / L;
        const int64_t abase = b * a_sb + (int64_t)l * TT;
        const int64_t gbase = b * g_sb + (int64_t)l * T * g_st;
        const int i2 = (i == 0) ? 0 : 1 + flip_index(i - 1, p, inv_p);
        const float* r1 = a1 + abase + (int64_t)i * T + 1;
        const float* r2 = a2 + abase + (int64_t)i2 * T + 1;
        float* o1 = g1 + gbase + (int64_t)i * g_st;
        float* o2 = g2 + gbase + (int64_t)i2 * g_st;
        const float w = (i == 0) ? gc : ga;
        if (lane == 0) { o1[0] = 0.f; o2[0] = 0.f; }            // column 0 is never looked at
        if (lane < (int)g_st - T) { o1[T + lane] = 0.f; o2[T + lane] = 0.f; }   // pad columns of the pitch (<= 3 here)
        for (int c = lane; c < N; c += 64) {
            const int c2 = flip_index(c, p, inv_p);
            const float d = r1[c] - r2[c2];
            const float s = (d > 0.f) ? w : ((d < 0.f) ? -w : 0.f);
            o1[1 + c] = s;
            o2[1 + c2] = -s;
        }
    }
}

// ---- p % 4 == 0 fast path: each lane owns 4 consecutive patches of one p-block ------------------------------
// view 1: cols 1 + i*p + 4j .. +3 (one 16-byte load, 4-byte aligned); view 2: the mirrored group of the same block,
// cols 1 + i*p + (p-4-4j) .. +3, read with one load and reversed in registers.  4x fewer load/store instructions
// than the lane-per-element form; unaligned dwordx4 is legal on gfx950 (addresses are dword aligned).
typedef float cf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ cf4 ld4(const float* p) { cf4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st4(float* p, cf4 v) { __builtin_memcpy(p, &v, 16); }

__global__ __launch_bounds__(CONS_THREADS) void cons_fwd_vec_kernel(
        const float* __restrict__ a1, const float* __restrict__ a2, int64_t a_sb,
        int L, int T, int p, int64_t rows, float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gpb = p >> 2, ngroups = p * gpb;               // groups per block, per row
    const float inv_p = 1.0f / (float)p, inv_g = 1.0f / (float)gpb;
    const int64_t TT = (int64_t)T * T;
    float cls = 0.f, aff = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * CONS_WAVES + wave; row < rows;
         row += (int64_t)gridDim.x * CONS_WAVES) {
        const int i = (int)(row % T);
        const int64_t bl = row / T;
        const int l = (int)(bl % L);
        const int64_t b = bl / L;
        const int64_t base = b * a_sb + (int64_t)l * TT;
        const int i2 = (i == 0) ? 0 : 1 + flip_index(i - 1, p, inv_p);
        const float* r1 = a1 + base + (int64_t)i * T + 1;
        const float* r2 = a2 + base + (int64_t)i2 * T + 1;
        float acc = 0.f;
        for (int gidx = lane; gidx < ngroups; gidx += 64) {
            const int blk = (int)(((float)gidx + 0.5f) * inv_g);
            const int j = gidx - blk * gpb;
            const cf4 x = ld4(r1 + blk * p + 4 * j);
            const cf4 y = ld4(r2 + blk * p + (p - 4 - 4 * j));
            acc += fabsf(x[0] - y[3]) + fabsf(x[1] - y[2]) + fabsf(x[2] - y[1]) + fabsf(x[3] - y[0]);
        }
        if (i == 0) cls += acc; else aff += acc;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cls += __shfl_xor(cls, off);
        aff += __shfl_xor(aff, off);
    }
    __shared__ float sh[2 * CONS_WAVES];
    if (lane == 0) { sh[2 * wave] = cls; sh[2 * wave + 1] = aff; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float c = 0.f, a = 0.f;
        for (int w = 0; w < CONS_WAVES; ++w) { c += sh[2 * w]; a += sh[2 * w + 1]; }
        partial[2 * blockIdx.x] = c;
        partial[2 * blockIdx.x + 1] = a;
    }
}

__global__ __launch_bounds__(CONS_THREADS) void cons_bwd_vec_kernel(
        const float* __restrict__ a1, const float* __restrict__ a2, int64_t a_sb,
        int L, int T, int p, int64_t rows, const float* __restrict__ gout2,
        float w_cls, float w_aff, float* __restrict__ g1, float* __restrict__ g2, int64_t g_sb, int64_t g_st) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gpb = p >> 2, ngroups = p * gpb;
    const float inv_p = 1.0f / (float)p, inv_g = 1.0f / (float)gpb;
    const int64_t TT = (int64_t)T * T;
    const float gc = gout2[0] * w_cls, ga = gout2[1] * w_aff;
    for (int64_t row = (int64_t)blockIdx.x * CONS_WAVES + wave; row < rows;
         row += (int64_t)gridDim.x * CONS_WAVES) {
        const int i = (int)(row % T);
        const int64_t bl = row / T;
        const int l = (int)(bl % L);
        const int64_t b = bl / L;
        const int64_t abase = b * a_sb + (int64_t)l * TT;
        const int64_t gbase = b * g_sb + (int64_t)l * T * g_st;
        const int i2 = (i == 0) ? 0 : 1 + flip_index(i - 1, p, inv_p);
        const float* r1 = a1 + abase + (int64_t)i * T + 1;
        const float* r2 = a2 + abase + (int64_t)i2 * T + 1;
        float* o1 = g1 + gbase + (int64_t)i * g_st;
        float* o2 = g2 + gbase + (int64_t)i2 * g_st;
        const float w = (i == 0) ? gc : ga;
        if (lane == 0) { o1[0] = 0.f; o2[0] = 0.f; }
        if (lane < (int)g_st - T) { o1[T + lane] = 0.f; o2[T + lane] = 0.f; }
        for (int gidx = lane; gidx < ngroups; gidx += 64) {
            const int blk = (int)(((float)gidx + 0.5f) * inv_g);
            const int j = gidx - blk * gpb;
            const int c1 = blk * p + 4 * j, c2 = blk * p + (p - 4 - 4 * j);
            const cf4 x = ld4(r1 + c1);
            const cf4 y = ld4(r2 + c2);
            cf4 s, t;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = x[e] - y[3 - e];
                s[e] = (d > 0.f) ? w : ((d < 0.f) ? -w : 0.f);
                t[3 - e] = -s[e];
            }
            st4(o1 + 1 + c1, s);
            st4(o2 + 1 + c2, t);
        }
    }
}

extern "C" int acr_consistency_fwd(const float* a1, const float* a2, int64_t a_sb, int32_t B, int32_t L,
                                   int32_t T, int32_t p, float* partial_ws, float* out2, void* stream) {
    ACR_CHECK_ARG(a1 && a2 && partial_ws && out2, "acr_consistency_fwd: null pointer");
    ACR_CHECK_ARG(B > 0 && L > 0 && p > 0 && T == p * p + 1, "acr_consistency_fwd: need T == p*p+1 (T=%d p=%d)", T, p);
    ACR_CHECK_ARG(a_sb >= (int64_t)L * T * T, "acr_consistency_fwd: batch stride %lld < L*T*T", (long long)a_sb);
    const int64_t rows = (int64_t)B * L * T;
    const int nb = cons_blocks(rows);
    hipStream_t st = (hipStream_t)stream;
    if ((p & 3) == 0)
        hipLaunchKernelGGL(cons_fwd_vec_kernel, dim3(nb), dim3(CONS_THREADS), 0, st, a1, a2, a_sb, L, T, p, rows, partial_ws);
    else
        hipLaunchKernelGGL(cons_fwd_kernel, dim3(nb), dim3(CONS_THREADS), 0, st, a1, a2, a_sb, L, T, p, rows, partial_ws);
    const double n = (double)(T - 1);
    hipLaunchKernelGGL(cons_reduce_kernel, dim3(1), dim3(256), 0, st, partial_ws, nb,
                       1.0 / ((double)B * L * n), 1.0 / ((double)B * L * n * n), out2);
    return acr_check_launch("acr_consistency_fwd");
}

extern "C" int acr_consistency_bwd(const float* a1, const float* a2, int64_t a_sb, int32_t B, int32_t L,
                                   int32_t T, int32_t p, const float* gout2, float* g1, float* g2,
                                   int64_t g_sb, int64_t g_st, void* stream) {
    ACR_CHECK_ARG(a1 && a2 && gout2 && g1 && g2, "acr_consistency_bwd: null pointer");
    ACR_CHECK_ARG(B > 0 && L > 0 && p > 0 && T == p * p + 1, "acr_consistency_bwd: need T == p*p+1 (T=%d p=%d)", T, p);
    ACR_CHECK_ARG(a_sb >= (int64_t)L * T * T && g_st >= T && g_st < T + 64 && g_sb >= (int64_t)L * T * g_st,
                  "acr_consistency_bwd: batch stride / row pitch too small");
    const int64_t rows = (int64_t)B * L * T;
    const int nb = cons_blocks(rows);
    const double n = (double)(T - 1);
    if ((p & 3) == 0)
        hipLaunchKernelGGL(cons_bwd_vec_kernel, dim3(nb), dim3(CONS_THREADS), 0, (hipStream_t)stream, a1, a2, a_sb, L, T,
                           p, rows, gout2, (float)(1.0 / ((double)B * L * n)), (float)(1.0 / ((double)B * L * n * n)),
                           g1, g2, g_sb, g_st);
    else
        hipLaunchKernelGGL(cons_bwd_kernel, dim3(nb), dim3(CONS_THREADS), 0, (hipStream_t)stream, a1, a2, a_sb, L, T, p,
                           rows, gout2, (float)(1.0 / ((double)B * L * n)), (float)(1.0 / ((double)B * L * n * n)),
                           g1, g2, g_sb, g_st);
    return acr_check_launch("acr_consistency_bwd");
}
