// Lab micro-benchmark (gfx950): does the ORDER of the six-term split-product MFMAs matter?  v_mfma_f32_32x32x16_bf16 with operands
// in registers, 24 MFMAs per iteration on 4 accumulators (a stage of the split-product GEMM):
//   order 0: six back-to-back on one accumulator, then the next accumulator (dependent chains of 6: what C3_MFMA6 / the GEMM emit)
//   order 1: round-robin over the 4 accumulators (every MFMA depends on the one 4 back)
//   order 2: pairs (2 on one accumulator, then the next)
// at 1 and 2 waves per SIMD.   build: hipcc -O3 --offload-arch=gfx950 mfma_bf16_chain.hip -o mfma_bf16_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) * (1.0f / 65536.0f) - 0.5f; }
#define MF(ACC, X, Y) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X, Y, ACC, 0, 0, 0)
template <int ORDER>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    unsigned seed = blockIdx.x * 977u + threadIdx.x * 131u + 7u;
    bf16x8 x[6], y[6];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 8; ++e) { x[i][e] = (__bf16)rnd(seed); y[i][e] = (__bf16)rnd(seed); }
    f32x16 a[4] = {{0}, {0}, {0}, {0}};
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 6; ++u) MF(a[t], x[u], y[(u + t) % 6]);
        } else if (ORDER == 1) {
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) MF(a[t], x[u], y[(u + t) % 6]);
        } else {
#pragma unroll
            for (int u = 0; u < 6; u += 2)
#pragma unroll
                for (int t = 0; t < 4; ++t) { MF(a[t], x[u], y[(u + t) % 6]); MF(a[t], x[u + 1], y[(u + 1 + t) % 6]); }
        }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) s += a[t][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ORDER>
static double run(float* out, int threads, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<ORDER>, dim3(256), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<ORDER>, dim3(256), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 3.0 * 256 * (threads / 64) * (double)iters * 24 * 32768.0;
    return flop / (ms * 1e-3) * 1e-12;
}
int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    const int iters = 200000;
    for (int rep = 0; rep < 2; ++rep) {
        printf("1 wave / SIMD : chains of 6 %.0f TF   round robin %.0f TF   pairs %.0f TF\n", run<0>(out, 256, iters), run<1>(out, 256, iters), run<2>(out, 256, iters));
        printf("2 waves / SIMD: chains of 6 %.0f TF   round robin %.0f TF   pairs %.0f TF\n", run<0>(out, 512, iters / 2), run<1>(out, 512, iters / 2), run<2>(out, 512, iters / 2));
    }
    return 0;
}
