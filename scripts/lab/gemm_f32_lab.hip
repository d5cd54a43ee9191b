// Lab harness for acr_gemm_f32: times the kernel on synthetic operands under -D variants (compiled ON the GPU box).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../acr_wsss_amd/csrc [-DLAB_...] gemm_f32_lab.hip -o lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <stdarg.h>
#ifdef LAB_STAMP
unsigned long long* g_lab_stamp = nullptr;
#endif
#include "../../acr_wsss_amd/csrc/gemm_f32.hip"
#include <algorithm>

static thread_local char g_err[512] = "";
void acr_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); }
int acr_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); return -2; } return 0; }
int32_t acr_opt(int) { return getenv("LAB_REG") ? 1 : 0; }

// bare MFMA loop: peak of this box
__global__ __launch_bounds__(256) void mfma_peak(float* out, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f + 0.5f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 25120, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
    float *a, *b, *c, *bias, *ws, *dy, *dw;
    size_t na = (size_t)M * K, nb = (size_t)N * K, nc = (size_t)M * N;
    hipMalloc(&a, na * 4); hipMalloc(&b, nb * 4); hipMalloc(&c, nc * 4); hipMalloc(&bias, N * 4); hipMalloc(&dy, nc * 4); hipMalloc(&dw, nb * 4);
    size_t nws = acr_gemm_f32_ws_floats(ACR_GEMM_TN, ACR_MATH_F32, N, K, M);
    hipMalloc(&ws, nws * 4 + 16);
    std::vector<float> h(std::max(std::max(na, nb), nc));
    srand(1);
    for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    hipMemcpy(a, h.data(), na * 4, hipMemcpyHostToDevice); hipMemcpy(b, h.data(), nb * 4, hipMemcpyHostToDevice);
    hipMemcpy(dy, h.data(), nc * 4, hipMemcpyHostToDevice); hipMemcpy(bias, h.data(), N * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    // peak
    float* po; hipMalloc(&po, 1024 * 256 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(mfma_peak, dim3(1024), dim3(256), 0, 0, po, 20000); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("bare MFMA loop: %.1f TF\n", 1024.0 * 4 * 4 * 20000 * 4096.0 / (ms * 1e-3) / 1e12);
    const double fl = 2.0 * M * N * K;
    auto run = [&](const char* name, auto fn) {
        for (int i = 0; i < 2; ++i) fn();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) fn();
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.3f ms  %6.1f TF", name, ms / 5, fl / (ms / 5 * 1e-3) / 1e12);
#ifdef LAB_STAMP
        {
            std::vector<unsigned long long> hs(2 * 8192);
            hipMemcpy(hs.data(), g_lab_stamp, hs.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> clk, cyc;
            for (int i = 0; i < 4096; ++i) if (hs[2 * i + 1]) { clk.push_back((double)hs[2 * i] / hs[2 * i + 1] * 0.1); cyc.push_back((double)hs[2 * i]); }
            std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
            if (!clk.empty()) printf("   clock median %.2f GHz, WG main-loop cycles median %.0f (min %.0f max %.0f)", clk[clk.size() / 2], cyc[cyc.size() / 2], cyc.front(), cyc.back());
            hipMemset(g_lab_stamp, 0, 2 * 8192 * 8);
        }
#endif
        printf("\n");
    };
#ifdef LAB_STAMP
    hipMalloc(&g_lab_stamp, 2 * 8192 * 8); hipMemset(g_lab_stamp, 0, 2 * 8192 * 8);
#endif
    run("NT x W^T + b", [&] { acr_gemm_f32(ACR_GEMM_NT, ACR_MATH_F32, 0, a, K, b, K, bias, nullptr, 0, c, N, nullptr, nullptr, M, N, K, nullptr, 0); });
    run("NN dy W", [&] { acr_gemm_f32(ACR_GEMM_NN, ACR_MATH_F32, 0, dy, N, b, K, nullptr, nullptr, 0, a, K, nullptr, nullptr, M, K, N, nullptr, 0); });
    run("TN dy^T x", [&] { acr_gemm_f32(ACR_GEMM_TN, ACR_MATH_F32, 0, dy, N, a, K, nullptr, nullptr, 0, dw, K, nullptr, nullptr, N, K, M, ws, 0); });
    printf("last error: %s\n", g_err);
    return 0;
}
