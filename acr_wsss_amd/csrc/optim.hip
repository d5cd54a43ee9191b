// Fused optimizer step of the bf16-weights / fp32-masters training mode: one launch over every parameter tensor.
// Reference semantics: tool/torchutils.py:10-31 PolyOptimizer = torch SGD whose momentum slot holds wt_dec (5e-4),
// weight decay 0, dampening 0, no Nesterov:   buf = mu * buf + g ;  w = w - lr * buf   (fp32), then the bf16 working
// copy of w is refreshed.  The stock path is 5 multi-tensor launches (cast g, mul, add, add, cast w) moving 32 B per
// parameter; this is one pass: 2 (g) + 4+4 (buf) + 4+4 (w) + 2 (bf16 w) = 20 B per parameter, HBM-bound.
#include "acr_common.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;


#define SGD_CHUNK 8192

__global__ __launch_bounds__(256) void sgd_step_kernel(const acr_sgd_tensor* __restrict__ tab, const int32_t* __restrict__ blk_tensor,
                                                       const int32_t* __restrict__ blk_chunk, float lr, float mu) {
    const acr_sgd_tensor t = tab[blk_tensor[blockIdx.x]];
    if (!t.grad) return;
    const int64_t c0 = (int64_t)blk_chunk[blockIdx.x] * SGD_CHUNK;
    const int64_t c1 = min(c0 + SGD_CHUNK, t.n);
    const bf16_t* g = (const bf16_t*)t.grad;
    bf16_t* p = (bf16_t*)t.param;
    const bool vec = ((((uintptr_t)g | (uintptr_t)p) & 15) == 0) && ((((uintptr_t)t.master | (uintptr_t)t.mom) & 15) == 0);
    if (vec) {
        for (int64_t i = c0 + threadIdx.x * 8; i + 8 <= c1; i += 256 * 8) {
            const bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + i);
            f32x4 b0 = *reinterpret_cast<const f32x4*>(t.mom + i), b1 = *reinterpret_cast<const f32x4*>(t.mom + i + 4);
            f32x4 w0 = *reinterpret_cast<const f32x4*>(t.master + i), w1 = *reinterpret_cast<const f32x4*>(t.master + i + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                b0[e] = mu * b0[e] + (float)gv[e];
                b1[e] = mu * b1[e] + (float)gv[4 + e];
                w0[e] = fmaf(-lr, b0[e], w0[e]);
                w1[e] = fmaf(-lr, b1[e], w1[e]);
                o[e] = (bf16_t)w0[e];
                o[4 + e] = (bf16_t)w1[e];
            }
            *reinterpret_cast<f32x4*>(t.mom + i) = b0;
            *reinterpret_cast<f32x4*>(t.mom + i + 4) = b1;
            *reinterpret_cast<f32x4*>(t.master + i) = w0;
            *reinterpret_cast<f32x4*>(t.master + i + 4) = w1;
            *reinterpret_cast<bf16x8*>(p + i) = o;
        }
        const int64_t tail = c0 + ((c1 - c0) & ~(int64_t)7);
        for (int64_t i = tail + threadIdx.x; i < c1; i += 256) {
            const float b = mu * t.mom[i] + (float)g[i];
            const float w = fmaf(-lr, b, t.master[i]);
            t.mom[i] = b; t.master[i] = w; p[i] = (bf16_t)w;
        }
    } else {
        for (int64_t i = c0 + threadIdx.x; i < c1; i += 256) {
            const float b = mu * t.mom[i] + (float)g[i];
            const float w = fmaf(-lr, b, t.master[i]);
            t.mom[i] = b; t.master[i] = w; p[i] = (bf16_t)w;
        }
    }
}

// the same step for an all-fp32 model (the reference's precision): gradient fp32, the parameter is its own master (no working
// copy).  Replaces torch's 13-launch multi-tensor SGD (0.95 ms per step) by one pass: 4 (g) + 4+4 (buf) + 4+4 (w) bytes.
__global__ __launch_bounds__(256) void sgd_step_f32_kernel(const acr_sgd_tensor* __restrict__ tab, const int32_t* __restrict__ blk_tensor,
                                                           const int32_t* __restrict__ blk_chunk, float lr, float mu) {
    const acr_sgd_tensor t = tab[blk_tensor[blockIdx.x]];
    if (!t.grad) return;
    const int64_t c0 = (int64_t)blk_chunk[blockIdx.x] * SGD_CHUNK;
    const int64_t c1 = min(c0 + SGD_CHUNK, t.n);
    const float* g = (const float*)t.grad;
    const bool vec = ((((uintptr_t)g | (uintptr_t)t.master | (uintptr_t)t.mom) & 15) == 0);
    int64_t i = c0 + threadIdx.x * 4;
    if (vec) {
        for (; i + 4 <= c1; i += 256 * 4) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
            f32x4 b = *reinterpret_cast<const f32x4*>(t.mom + i), w = *reinterpret_cast<const f32x4*>(t.master + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                b[e] = mu * b[e] + gv[e];
                w[e] = fmaf(-lr, b[e], w[e]);
            }
            *reinterpret_cast<f32x4*>(t.mom + i) = b;
            *reinterpret_cast<f32x4*>(t.master + i) = w;
        }
    }
    const int64_t tail = vec ? c0 + ((c1 - c0) & ~(int64_t)3) : c0;
    for (int64_t j = tail + threadIdx.x; j < c1; j += 256) {
        const float b = mu * t.mom[j] + g[j];
        t.mom[j] = b;
        t.master[j] = fmaf(-lr, b, t.master[j]);
    }
}

extern "C" int acr_sgd_step_f32(const void* table, const int32_t* blk_tensor, const int32_t* blk_chunk, int32_t nblocks, float lr,
                                float momentum, void* stream) {
    ACR_CHECK_ARG(table && blk_tensor && blk_chunk && nblocks > 0, "acr_sgd_step_f32: null pointer / empty launch");
    hipLaunchKernelGGL(sgd_step_f32_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const acr_sgd_tensor*)table,
                       blk_tensor, blk_chunk, lr, momentum);
    return acr_check_launch("acr_sgd_step_f32");
}

extern "C" int32_t acr_sgd_chunk_elems(void) { return SGD_CHUNK; }

extern "C" int acr_sgd_step_bf16(const void* table, const int32_t* blk_tensor, const int32_t* blk_chunk, int32_t nblocks,
                                 float lr, float momentum, void* stream) {
    ACR_CHECK_ARG(table && blk_tensor && blk_chunk && nblocks > 0, "acr_sgd_step_bf16: null pointer / empty launch");
    hipLaunchKernelGGL(sgd_step_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const acr_sgd_tensor*)table,
                       blk_tensor, blk_chunk, lr, momentum);
    return acr_check_launch("acr_sgd_step_bf16");
}

// ---------------------------------------------------------------------------------------------------------------
// W^T of every block Linear in ONE launch, right after the optimizer step.  The input-gradient GEMMs want the weight
// as (in, out); the per-layer `weight.t().contiguous()` they used ran at 0.7 TB/s and cost 0.6 ms per step (48
// launches).  Table-driven like the optimizer step; 64x64 tiles through LDS (padded rows), 16-byte accesses both ways.
// ---------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void transpose_many_kernel(const acr_tr_tensor* __restrict__ tab, const int32_t* __restrict__ blk_tensor) {
    __shared__ bf16_t tile[64][72];
    const acr_tr_tensor t = tab[blk_tensor[blockIdx.x]];
    const int lt = blockIdx.x - t.tile0;
    const int r0 = (lt / t.tiles_c) * 64, c0 = (lt % t.tiles_c) * 64;
    const bf16_t* s = (const bf16_t*)t.src;
    bf16_t* d = (bf16_t*)t.dst;
    const int tid = threadIdx.x;
    // load 64 rows x 64 cols: 8 threads x 16 B per row, 32 rows per pass
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = p * 32 + (tid >> 3), c = (tid & 7) * 8;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r0 + r < t.rows && c0 + c < t.cols) v = *reinterpret_cast<const bf16x8*>(s + (int64_t)(r0 + r) * t.cols + c0 + c);
        *reinterpret_cast<bf16x8*>(&tile[r][c]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int c = p * 32 + (tid >> 3), r = (tid & 7) * 8;      // output row = source column c, 8 source rows r..r+7
        if (c0 + c < t.cols && r0 + r < t.rows) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[r + e][c];
            *reinterpret_cast<bf16x8*>(d + (int64_t)(c0 + c) * t.rows + r0 + r) = v;
        }
    }
}

// the same for fp32 weights (rows, cols multiples of 4): 64x64 tile through LDS, 16-byte accesses both ways.  In fp32 the
// input-gradient GEMM on W as stored (NN: the B tile is read one float at a time) runs at 105-110 TFLOP/s, on the
// transposed copy (NT: 16-byte fragment reads) at 121-125 (scripts/lab/gemm_nn_vs_nt.py), bit-identical results.
__global__ __launch_bounds__(256) void transpose_many_f32_kernel(const acr_tr_tensor* __restrict__ tab, const int32_t* __restrict__ blk_tensor) {
    __shared__ float tile[64][68];
    const acr_tr_tensor t = tab[blk_tensor[blockIdx.x]];
    const int lt = blockIdx.x - t.tile0;
    const int r0 = (lt / t.tiles_c) * 64, c0 = (lt % t.tiles_c) * 64;
    const float* s = (const float*)t.src;
    float* d = (float*)t.dst;
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {                            // 16 threads x 16 B per row, 16 rows per pass
        const int r = p * 16 + (tid >> 4), c = (tid & 15) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < t.rows && c0 + c < t.cols) v = *reinterpret_cast<const f32x4*>(s + (int64_t)(r0 + r) * t.cols + c0 + c);
        *reinterpret_cast<f32x4*>(&tile[r][c]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = p * 16 + (tid >> 4), r = (tid & 15) * 4;     // output row = source column c, 4 source rows r..r+3
        if (c0 + c < t.cols && r0 + r < t.rows) {
            const f32x4 v = {tile[r][c], tile[r + 1][c], tile[r + 2][c], tile[r + 3][c]};
            *reinterpret_cast<f32x4*>(d + (int64_t)(c0 + c) * t.rows + r0 + r) = v;
        }
    }
}

extern "C" int acr_transpose_many_f32(const void* table, const int32_t* blk_tensor, int32_t nblocks, void* stream) {
    ACR_CHECK_ARG(table && blk_tensor && nblocks > 0, "acr_transpose_many_f32: null pointer / empty launch");
    hipLaunchKernelGGL(transpose_many_f32_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const acr_tr_tensor*)table,
                       blk_tensor);
    return acr_check_launch("acr_transpose_many_f32");
}

extern "C" int acr_transpose_many_bf16(const void* table, const int32_t* blk_tensor, int32_t nblocks, void* stream) {
    ACR_CHECK_ARG(table && blk_tensor && nblocks > 0, "acr_transpose_many_bf16: null pointer / empty launch");
    hipLaunchKernelGGL(transpose_many_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const acr_tr_tensor*)table,
                       blk_tensor);
    return acr_check_launch("acr_transpose_many_bf16");
}
