// Token assembly of the hybrid ViT (models/vision_transformer.py:449-467: x = patch_embed(x) -> flatten(2).transpose(1, 2) ->
// cat(cls_token[, dist_token], x) -> + pos_embed) as ONE pass each way.  The reference's chain is four elementwise / copy kernels
// forward (bias add in NCHW, transposing cat, position add) and, backward, a transposing copy for the projection's gradient, a
// (0, 2, 3) reduction for its bias gradient and a batch reduction for the position embedding's -- about 0.9 ms of a 126 ms step.
//
//   forward   tok[b][P + t][d] = y[b][d][t] + bias[d] + pos[P + t][d]        tok[b][p][d] = prefix[p][d] + pos[p][d]   (p < P)
//   backward  dy[b][d][t] = dtok[b][P + t][d]      dpos[r][d] = sum_b dtok[b][r][d]
//             (dbias = sum_t dpos[P + t], dprefix = dpos[:P]: two tiny reductions the caller does on dpos)
// 32 x 32 tiles through LDS (128-byte rows on both sides); the backward workgroup walks the batch with its tile so that the batch sum
// stays in registers, in batch order (deterministic).
#include "acr_common.h"
#include "../../include/acr_hip.h"

__global__ __launch_bounds__(256) void tokens_fwd_kernel(const float* __restrict__ y, const float* __restrict__ bias, const float* __restrict__ prefix,
                                                         const float* __restrict__ pos, float* __restrict__ tok, int D, int T, int P) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int t0 = blockIdx.x * 32, d0 = blockIdx.y * 32, b = blockIdx.z;
    const float* yb = y + (int64_t)b * D * T;
    float* tb = tok + (int64_t)b * (P + T) * D;
#pragma unroll
    for (int j = ty; j < 32; j += 8)
        tile[j][tx] = (d0 + j < D && t0 + tx < T) ? yb[(int64_t)(d0 + j) * T + t0 + tx] : 0.f;
    __syncthreads();
    const int d = d0 + tx;
    if (d < D) {
        const float bd = bias[d];
#pragma unroll
        for (int j = ty; j < 32; j += 8) {
            const int r = P + t0 + j;
            if (t0 + j < T) tb[(int64_t)r * D + d] = (tile[tx][j] + bd) + pos[(int64_t)r * D + d];
        }
        if (blockIdx.x == 0 && ty < P) tb[(int64_t)ty * D + d] = prefix[ty * D + d] + pos[ty * D + d];
    }
}

__global__ __launch_bounds__(256) void tokens_bwd_kernel(const float* __restrict__ dtok, float* __restrict__ dy, float* __restrict__ dpos, int B, int D,
                                                         int T, int P) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int t0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const int d = d0 + tx;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, accp = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* gb = dtok + (int64_t)b * (P + T) * D;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = ty + 8 * q;
            const float v = (d < D && t0 + j < T) ? gb[(int64_t)(P + t0 + j) * D + d] : 0.f;
            acc[q] += v;
            tile[j][tx] = v;
        }
        if (blockIdx.x == 0 && ty < P && d < D) accp += gb[(int64_t)ty * D + d];
        __syncthreads();
        float* yb = dy + (int64_t)b * D * T;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = ty + 8 * q;                       // channel d0 + j, token t0 + tx
            if (d0 + j < D && t0 + tx < T) yb[(int64_t)(d0 + j) * T + t0 + tx] = tile[tx][j];
        }
        __syncthreads();
    }
    if (d < D) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = ty + 8 * q;
            if (t0 + j < T) dpos[(int64_t)(P + t0 + j) * D + d] = acc[q];
        }
        if (blockIdx.x == 0 && ty < P) dpos[(int64_t)ty * D + d] = accp;
    }
}

extern "C" int acr_tokens_fwd_f32(const float* y, const float* bias, const float* prefix, const float* pos, float* tok, int32_t B, int32_t D,
                                  int32_t T, int32_t P, void* stream) {
    ACR_CHECK_ARG(y && bias && prefix && pos && tok, "acr_tokens_fwd_f32: null pointer");
    ACR_CHECK_ARG(B > 0 && D > 0 && T > 0 && P >= 0 && P <= 8 && B < 65536 && (D + 31) / 32 < 65536, "acr_tokens_fwd_f32: bad shape (B=%d D=%d T=%d P=%d)", B, D, T, P);
    hipLaunchKernelGGL(tokens_fwd_kernel, dim3((T + 31) / 32, (D + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, y, bias, prefix, pos, tok, D, T, P);
    return acr_check_launch("acr_tokens_fwd_f32");
}

extern "C" int acr_tokens_bwd_f32(const float* dtok, float* dy, float* dpos, int32_t B, int32_t D, int32_t T, int32_t P, void* stream) {
    ACR_CHECK_ARG(dtok && dy && dpos, "acr_tokens_bwd_f32: null pointer");
    ACR_CHECK_ARG(B > 0 && D > 0 && T > 0 && P >= 0 && P <= 8 && (D + 31) / 32 < 65536, "acr_tokens_bwd_f32: bad shape (B=%d D=%d T=%d P=%d)", B, D, T, P);
    hipLaunchKernelGGL(tokens_bwd_kernel, dim3((T + 31) / 32, (D + 31) / 32), dim3(256), 0, (hipStream_t)stream, dtok, dy, dpos, B, D, T, P);
    return acr_check_launch("acr_tokens_bwd_f32");
}
