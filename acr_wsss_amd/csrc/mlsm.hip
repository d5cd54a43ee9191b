// Multilabel soft-margin classification loss of the ACR step (train_acr.py:160-161: F.multilabel_soft_margin_loss on the class logits
// of each view) and its backward, one launch each way.  The stock op is ~20 tiny kernels per view and direction (log-sigmoid twice,
// products, negations, row sums, divisions, the mean): 40 launches of one workgroup each per training step for 640 numbers.
//   loss = mean_n [ (1/C) sum_c -( y ls(x) + (1 - y) ls(-x) ) ],   ls(x) = min(x, 0) - log1p(exp(-|x|))   (ATen's log_sigmoid)
//   dx[n][c] = g (sigmoid(x) - y) / (N C)
// One workgroup: thread t sums rows t, t + 256, ... (classes in order), the row means meet in LDS and are added in row order by a
// fixed tree -- deterministic.  g is read from device memory (the upstream gradient of a scalar loss: no host round trip).
#include "acr_common.h"

__device__ __forceinline__ float mlsm_ls(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

__global__ __launch_bounds__(256) void mlsm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int N, int C, int64_t ldx,
                                                       int64_t ldy, float* __restrict__ out) {
    __shared__ float sh[256];
    float acc = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float* xr = x + (int64_t)n * ldx;
        const float* yr = y + (int64_t)n * ldy;
        float s = 0.f;
        for (int c = 0; c < C; ++c) {
            const float xv = xr[c], yv = yr[c];
            s += -(yv * mlsm_ls(xv) + (1.f - yv) * mlsm_ls(-xv));
        }
        acc += s / (float)C;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0] / (float)N;
}

__global__ __launch_bounds__(256) void mlsm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ g, int N,
                                                       int C, int64_t ldx, int64_t ldy, float* __restrict__ dx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    const float xv = x[(int64_t)n * ldx + c], yv = y[(int64_t)n * ldy + c];
    const float e = expf(-fabsf(xv));
    const float sig = xv >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    dx[i] = g[0] * (sig - yv) / ((float)N * (float)C);
}

extern "C" int acr_mlsm_fwd_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, int32_t N, int32_t C, float* loss, void* stream) {
    ACR_CHECK_ARG(x && y && loss, "acr_mlsm_fwd_f32: null pointer");
    ACR_CHECK_ARG(N > 0 && C > 0 && ldx >= C && ldy >= C, "acr_mlsm_fwd_f32: bad shape (N=%d C=%d)", N, C);
    hipLaunchKernelGGL(mlsm_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, y, N, C, ldx, ldy, loss);
    return acr_check_launch("acr_mlsm_fwd_f32");
}
extern "C" int acr_mlsm_bwd_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* g, int32_t N, int32_t C, float* dx,
                                void* stream) {
    ACR_CHECK_ARG(x && y && g && dx, "acr_mlsm_bwd_f32: null pointer");
    ACR_CHECK_ARG(N > 0 && C > 0 && ldx >= C && ldy >= C && (int64_t)N * C < (1ll << 31), "acr_mlsm_bwd_f32: bad shape (N=%d C=%d)", N, C);
    hipLaunchKernelGGL(mlsm_bwd_kernel, dim3((unsigned)(((int64_t)N * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, g, N, C, ldx, ldy, dx);
    return acr_check_launch("acr_mlsm_bwd_f32");
}
