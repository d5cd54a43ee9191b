// Does a DEPENDENT chain of v_mfma_f32_32x32x2_f32 (same accumulator back to back) issue at the 64-cycle rate?
// hipcc -O3 --offload-arch=gfx950 mfma_chain_lab.hip -o /tmp/chain && /tmp/chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void chain(float* out, int iters, float x, float y) {
    f32x16 a[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) a[i][e] = 0.f;
    float xx = x + threadIdx.x * 1e-6f, yy = y;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 32 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) a[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(xx, yy, a[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += a[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// chain with a VALU phase between two chains (attention-like): 32 dependent MFMAs, ~NV VALU ops on the result, repeat
// MODE 0: nothing; 1: s_setprio 1 for odd hardware wave slots; 2: s_setprio 1 for odd blocks; 3: odd hardware slots start
// half a phase late (s_sleep); 4: prio by slot + setprio dropped during the VALU phase
template <int NV, int MODE>
__global__ __launch_bounds__(256) void chain_valu(float* out, int iters, float x, float y) {
    f32x16 a = {0}, b = {0};
    float xx = x + threadIdx.x * 1e-6f, yy = y;
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hwid));
    const bool odd = (MODE == 2) ? (blockIdx.x & 1) : (hwid & 1);
    if ((MODE == 1 || MODE == 2 || MODE == 4) && odd) __builtin_amdgcn_s_setprio(2);
    if (MODE == 3 && odd) { for (int i = 0; i < 16; ++i) __builtin_amdgcn_s_sleep(2); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 32; ++rep) a = __builtin_amdgcn_mfma_f32_32x32x2f32(xx, yy, a, 0, 0, 0);
        f32x16 p;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float t = a[e];
#pragma unroll
            for (int v = 0; v < NV; ++v) t = fmaf(t, 0.999f, 0.001f);
            p[e] = __builtin_amdgcn_exp2f(t * 1e-3f);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) b = __builtin_amdgcn_mfma_f32_32x32x2f32(p[e], yy, b, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) b = __builtin_amdgcn_mfma_f32_32x32x2f32(p[e], xx, b, 0, 0, 0);
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += a[e] + b[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    auto run = [&](const char* name, auto launch, double mfma_per_wave) {
        for (int r = 0; r < 2; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); }
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.3f ms\n", name, ms);
        (void)mfma_per_wave;
    };
    const int it = 4000;
    // waves per SIMD = blocks*4 waves / (256 CUs * 4 SIMDs): 256 blocks -> 1 wave/SIMD, 512 -> 2, 1024 -> 4
    for (int blocks : {256, 512, 1024}) {
        printf("-- %d blocks (%d waves/SIMD): ideal %.3f ms\n", blocks, blocks / 256, blocks / 256 * it * 32.0 * 64 / 2.4e9 * 1e3);
        run("1 accumulator (dependent chain)", [&] { hipLaunchKernelGGL(chain<1>, dim3(blocks), dim3(256), 0, 0, out, it, 0.5f, 0.25f); }, 0);
        run("2 accumulators", [&] { hipLaunchKernelGGL(chain<2>, dim3(blocks), dim3(256), 0, 0, out, it, 0.5f, 0.25f); }, 0);
        run("4 accumulators", [&] { hipLaunchKernelGGL(chain<4>, dim3(blocks), dim3(256), 0, 0, out, it, 0.5f, 0.25f); }, 0);
    }
    for (int blocks : {256, 512, 1024}) {
        printf("-- %d blocks, 64 MFMA + VALU phase per iter: MFMA-only ideal %.3f ms\n", blocks, blocks / 256 * (it / 2) * 64.0 * 64 / 2.4e9 * 1e3);
        run("chain + 2 VALU/elem + exp", [&] { hipLaunchKernelGGL((chain_valu<2, 0>), dim3(blocks), dim3(256), 0, 0, out, it / 2, 0.5f, 0.25f); }, 0);
        run("chain + 8 VALU/elem + exp", [&] { hipLaunchKernelGGL((chain_valu<8, 0>), dim3(blocks), dim3(256), 0, 0, out, it / 2, 0.5f, 0.25f); }, 0);
        run("  8 VALU, prio by hw slot parity", [&] { hipLaunchKernelGGL((chain_valu<8, 1>), dim3(blocks), dim3(256), 0, 0, out, it / 2, 0.5f, 0.25f); }, 0);
        run("  8 VALU, prio by block parity", [&] { hipLaunchKernelGGL((chain_valu<8, 2>), dim3(blocks), dim3(256), 0, 0, out, it / 2, 0.5f, 0.25f); }, 0);
        run("  8 VALU, odd slots start late", [&] { hipLaunchKernelGGL((chain_valu<8, 3>), dim3(blocks), dim3(256), 0, 0, out, it / 2, 0.5f, 0.25f); }, 0);
        run("  24 VALU", [&] { hipLaunchKernelGGL((chain_valu<24, 0>), dim3(blocks), dim3(256), 0, 0, out, it / 2, 0.5f, 0.25f); }, 0);
        run("  24 VALU, prio by hw slot parity", [&] { hipLaunchKernelGGL((chain_valu<24, 1>), dim3(blocks), dim3(256), 0, 0, out, it / 2, 0.5f, 0.25f); }, 0);
        run("  24 VALU, odd slots start late", [&] { hipLaunchKernelGGL((chain_valu<24, 3>), dim3(blocks), dim3(256), 0, 0, out, it / 2, 0.5f, 0.25f); }, 0);
    }
    return 0;
}
