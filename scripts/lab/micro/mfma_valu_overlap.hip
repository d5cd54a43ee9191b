// Lab micro-benchmark (gfx950): do bf16 MFMAs and plain VALU instructions overlap on one SIMD?
// Per iteration a wave issues 4 MFMA (32x32x16 bf16, independent accumulators) and / or 32 independent v_fma_f32.
//   role 0: MFMA only   role 1: VALU only   role 2: interleaved in one wave (1 MFMA + 8 VALU, four times)
// Blocks of 4 waves (one per SIMD) or 8 waves (two per SIMD); with 8 waves the two halves can take different roles -- both
// possible wave -> SIMD mappings (w & 3 and w >> 1) are tried.
// build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define NV 8
#define VALU8() for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c))
// sel: 0 = all waves role A; 1 = role by (wave & 1); 2 = role by (wave >> 2) & 1; 3 = role by (wave >> 1) & 1
__global__ __launch_bounds__(512) void k(float* out, int iters, int roleA, int roleB, int sel) {
    const int wave = threadIdx.x >> 6;
    const int pick = sel == 0 ? 0 : sel == 1 ? (wave & 1) : sel == 2 ? ((wave >> 2) & 1) : ((wave >> 1) & 1);
    const int role = pick ? roleB : roleA;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(i * 0.5f); }
    float v[NV];
    for (int i = 0; i < NV; ++i) v[i] = threadIdx.x * 0.01f + i;
    const float m = 1.0001f, c = 0.5f;
    if (role == 0) {
        for (int it = 0; it < iters; ++it) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
        }
    } else if (role == 1) {
        for (int it = 0; it < iters; ++it) { VALU8(); VALU8(); VALU8(); VALU8(); }
    } else {
        for (int it = 0; it < iters; ++it) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            VALU8();
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
            VALU8();
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
            VALU8();
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
            VALU8();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    for (int i = 0; i < NV; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static float run(float* out, int iters, int threads, int a, int b, int sel) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, iters, a, b, sel);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, iters, a, b, sel);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e6f / iters;
}
int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    printf("ns per iteration (4 MFMA 32x32x16 bf16 and / or 32 v_fma_f32 per wave), 256 blocks = one per CU\n");
    printf("1 wave / SIMD :  MFMA %.1f   VALU %.1f   interleaved in the wave %.1f\n", run(out, iters, 256, 0, 0, 0), run(out, iters, 256, 1, 1, 0),
           run(out, iters, 256, 2, 2, 0));
    printf("2 waves / SIMD:  MFMA %.1f   VALU %.1f   interleaved in the wave %.1f\n", run(out, iters, 512, 0, 0, 0), run(out, iters, 512, 1, 1, 0),
           run(out, iters, 512, 2, 2, 0));
    printf("2 waves / SIMD, half the waves MFMA, half VALU; split by wave&1: %.1f   by wave>>2: %.1f   by (wave>>1)&1: %.1f\n",
           run(out, iters, 512, 0, 1, 1), run(out, iters, 512, 0, 1, 2), run(out, iters, 512, 0, 1, 3));
    return 0;
}
