// Shared device/host helpers for libacr_hip.so (gfx950 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/acr_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define ACR_LOG2E 1.4426950408889634f

// ---- host side error plumbing (thread-local message, never throws) ---------------------------
void acr_set_error(const char* fmt, ...);
#define ACR_CHECK_ARG(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            acr_set_error(__VA_ARGS__);     \
            return ACR_ERR_INVALID;         \
        }                                   \
    } while (0)
int acr_check_launch(const char* what);
int32_t acr_opt(int option);          // explicit option table (acr_set_option), api.hip

// ---- barrier that publishes LDS-DMA data ---------------------------------------------------------
// `global_load_lds` writes LDS asynchronously and is tracked by the issuing wave's vmcnt only.  hipcc does NOT always put
// the s_waitcnt vmcnt(0) in front of a __syncthreads() that follows (seen missing when the barrier sits at a loop head
// reached over `continue` edges: waves passed the barrier with their DMA still in flight and a partner read stale LDS,
// once in ~50 launches).  Every barrier that hands DMA'd tiles to other waves therefore goes through this.
__device__ __forceinline__ void acr_dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ---- XCD-aware block remap --------------------------------------------------------------------
// Workgroup barrier WITHOUT __syncthreads()'s fences, for loops that keep LDS-DMA (`global_load_lds`) in flight across it.
// __syncthreads() = release fence + s_barrier + acquire fence, and hipcc implements the workgroup-scope release of LDS as "every
// outstanding LDS-DMA has landed": it puts `s_waitcnt vmcnt(0)` in front of the barrier whenever a DMA may be pending (found in
// the ISA of every ring kernel, round 4) -- a ring that is N stages ahead degenerates to "wait for the newest stage every step".
// Contract of this one: the caller has waited (counted `s_waitcnt vmcnt(n)`) for the DMA pieces the OTHER waves are about to read,
// and `s_waitcnt lgkmcnt(0)` for its own ds_write / ds_read where another wave overwrites or reads those addresses next.
// The "memory" clobber keeps the compiler from moving memory accesses across it.
__device__ __forceinline__ void acr_barrier_nofence() { asm volatile("s_barrier" ::: "memory"); }

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Remap the
// linear block id so that each XCD walks one contiguous chunk of the work list: neighbouring
// tiles (same batch/head -> same K/V panels) then hit the same 4 MiB L2.  Bijective for any n.
__device__ __forceinline__ int acr_xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// C/D fragment row of a 32x32 MFMA accumulator register (dtype independent on gfx950):
// element `reg` of lane l sits at row krow(reg, l>>5), column l&31.
__device__ __forceinline__ int acr_krow(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// ---- dtype-generic 4-element loads / stores (fp32 math everywhere) ---------------------------
template <typename T>
__device__ __forceinline__ f32x4 acr_load4(const T* p);
template <>
__device__ __forceinline__ f32x4 acr_load4<float>(const float* p) {
    return *reinterpret_cast<const f32x4*>(p);
}
template <>
__device__ __forceinline__ f32x4 acr_load4<__bf16>(const __bf16* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
template <typename T>
__device__ __forceinline__ void acr_store1(T* p, float v);
template <>
__device__ __forceinline__ void acr_store1<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void acr_store1<__bf16>(__bf16* p, float v) { *p = (__bf16)v; }
template <typename T>
__device__ __forceinline__ void acr_store4(T* p, f32x4 v);
template <>
__device__ __forceinline__ void acr_store4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <>
__device__ __forceinline__ void acr_store4<__bf16>(__bf16* p, f32x4 v) {
    bf16x4 r = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = r;
}
template <typename T>
__device__ __forceinline__ float acr_load1(const T* p) { return (float)*p; }

// ---- slab sums of a K-split product whose OUTPUT is small --------------------------------------------------------------------------
// out[i] = sum_k slab[k][i] (float4 per column index i < n4).  One thread per output float4 walking all slabs (the products' own
// reduce kernels) leaves a 64 x 64 weight gradient with 4 workgroups reading 256 slabs one after the other: 123 us for 4 MB
// (scripts/lab/conv_wgrad_trace.py; the stem's 1x1 / 3x3 weight gradients spent 1.4 ms per step there).  Here a workgroup is
// 32 columns x G slab groups: thread (tx, ty) sums slabs ty, ty + G, ... in ascending order with four loads in flight, and the G
// partial sums meet in LDS, where they are added in group order.  Deterministic: the grouping depends on (n4, nslab) only.
template <int G>
__global__ __launch_bounds__(32 * G) void acr_slab_sum_wide_kernel(const float* __restrict__ ws, int nslab, int64_t n4, float* __restrict__ out) {
    __shared__ f32x4 part[G][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + tx;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        const f32x4* p = reinterpret_cast<const f32x4*>(ws) + i;
        int k = ty;
        for (; k + 3 * G < nslab; k += 4 * G) {
            const f32x4 v0 = p[(int64_t)k * n4], v1 = p[(int64_t)(k + G) * n4], v2 = p[(int64_t)(k + 2 * G) * n4], v3 = p[(int64_t)(k + 3 * G) * n4];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < nslab; k += G) s += p[(int64_t)k * n4];
    }
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < n4) {
        f32x4 t = part[0][tx];
#pragma unroll
        for (int g = 1; g < G; ++g) t += part[g][tx];
        reinterpret_cast<f32x4*>(out)[i] = t;
    }
}
// slab groups for an output of n4 float4: enough threads to keep ~128k loads in flight, 0 = the one-thread-per-output kernel is fine
static inline int acr_slab_sum_groups(int nslab, int64_t n4) {
    if (n4 >= 65536 || nslab < 8) return 0;
    int g = 32;
    while (g > 4 && n4 * (g / 2) >= 131072) g >>= 1;
    while (g > 4 && g > nslab) g >>= 1;
    return g;
}
// true when the wide kernel took the sum
static inline bool acr_slab_sum_wide(const float* ws, int nslab, int64_t n4, float* out, hipStream_t st) {
    const int g = acr_slab_sum_groups(nslab, n4);
    if (g == 0) return false;
    const dim3 grid((unsigned)((n4 + 31) / 32));
    if (g == 32) hipLaunchKernelGGL((acr_slab_sum_wide_kernel<32>), grid, dim3(1024), 0, st, ws, nslab, n4, out);
    else if (g == 16) hipLaunchKernelGGL((acr_slab_sum_wide_kernel<16>), grid, dim3(512), 0, st, ws, nslab, n4, out);
    else if (g == 8) hipLaunchKernelGGL((acr_slab_sum_wide_kernel<8>), grid, dim3(256), 0, st, ws, nslab, n4, out);
    else hipLaunchKernelGGL((acr_slab_sum_wide_kernel<4>), grid, dim3(128), 0, st, ws, nslab, n4, out);
    return true;
}
