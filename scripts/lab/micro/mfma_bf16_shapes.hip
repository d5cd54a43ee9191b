// Lab micro-benchmark (gfx950): sustained FLOP/s of v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16 on RANDOM operands held
// in registers (the guide reports the 16x16x32 loop holding a higher clock), 1 or 2 waves per SIMD, every CU busy.
// build: hipcc -O3 --offload-arch=gfx950 mfma_bf16_shapes.hip -o mfma_bf16_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) * (1.0f / 65536.0f) - 0.5f; }
template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    unsigned seed = blockIdx.x * 977u + threadIdx.x * 131u + 7u;
    bf16x8 x[4], y[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { x[i][e] = (__bf16)rnd(seed); y[i][e] = (__bf16)rnd(seed); }
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 a[4] = {{0}, {0}, {0}, {0}};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) a[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[u], y[(u + t) & 3], a[t], 0, 0, 0);
        for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) s += a[t][e];
    } else {
        f32x4 a[16];
        for (int t = 0; t < 16; ++t) a[t] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 2; ++u)                      // 32 MFMA of 16x16x32 = the FLOP of 16 MFMA of 32x32x16
#pragma unroll
                for (int t = 0; t < 16; ++t) a[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[(u + t) & 3], y[(2 * u + t) & 3], a[t], 0, 0, 0);
        for (int t = 0; t < 16; ++t) for (int e = 0; e < 4; ++e) s += a[t][e];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int SHAPE>
static double run(float* out, int threads, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 5.0 * 256 * (threads / 64) * (double)iters * 16 * 32768.0;      // per wave and iteration: 16 x (32x32x16x2)
    return flop / (ms * 1e-3) * 1e-12;
}
int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    const int iters = 400000;
    for (int rep = 0; rep < 2; ++rep) {
        printf("1 wave / SIMD : 32x32x16 %.0f TF   16x16x32 %.0f TF\n", run<32>(out, 256, iters), run<16>(out, 256, iters));
        printf("2 waves / SIMD: 32x32x16 %.0f TF   16x16x32 %.0f TF\n", run<32>(out, 512, iters / 2), run<16>(out, 512, iters / 2));
    }
    return 0;
}
