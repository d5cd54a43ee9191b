// bf16 MFMA GEMM for the attention block's projections (models/vision_transformer.py:200 `self.qkv(x)`, :212
// `self.proj(x)`): Y[M,N] = A[M,K] . B[N,K]^T (+ bias[N]) (+ R[M,N]), both operands K-contiguous -- the natural
// nn.Linear layout ("NT").  The same kernel serves the input gradient (dX = dY . W with B = W^T, a small per-step
// transpose of the weight) so the qkv/proj matmuls of forward and backward-data run on hand-written MFMA code.
//
// Structure (gfx950, wave64): 128x128x64 block tile, 4 waves as 2x2, each wave 64x64 = 2x2 accumulators of
// v_mfma_f32_32x32x16_bf16; A/B tiles staged global -> registers -> LDS with the next K-step's loads in flight under
// the current step's 64 MFMAs (two LDS buffers, one barrier per step); LDS rows padded to 144 B so every
// ds_read_b128 lane group is conflict-free; XCD-aware tile order keeps the 128-row A panel of consecutive N tiles
// in one L2.  fp32 accumulate, bias / residual fused in the epilogue, bf16 out.
#include "acr_common.h"

typedef __bf16 bf16_t;
#define GBP 72
#define GEMM_MEMBAR() asm volatile("" ::: "memory")

struct GemmRegs { bf16x8 v[4]; };          // 128 rows x 64 cols = 1024 16-byte chunks / 256 threads

__device__ __forceinline__ void gemm_gload(GemmRegs& t, const bf16_t* g, int64_t ld, int row0, int nrows, int k0, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = tid + i * 256;
        const int rc = min(row0 + (slot >> 3), nrows - 1);
        t.v[i] = *reinterpret_cast<const bf16x8*>(g + (int64_t)rc * ld + k0 + (slot & 7) * 8);
    }
}
__device__ __forceinline__ void gemm_lstore(bf16_t* lds, const GemmRegs& t, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = tid + i * 256;
        *reinterpret_cast<bf16x8*>(lds + (slot >> 3) * GBP + (slot & 7) * 8) = t.v[i];
    }
}

template <bool BIAS, bool RESID>
__global__ __launch_bounds__(256) void gemm_nt_bf16_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                           const bf16_t* __restrict__ B, int64_t ldb,
                                                           const bf16_t* __restrict__ bias,
                                                           const bf16_t* __restrict__ R, int64_t ldr,
                                                           bf16_t* __restrict__ Y, int64_t ldy, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) bf16_t As[2][128 * GBP];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[2][128 * GBP];
    const int ntn = (N + 127) >> 7;
    const int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id / ntn, tn = id % ntn;
    const int m0 = tm * 128, n0 = tn * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    // One K-step of global loads in flight under the current step's MFMAs.  (A second register set in flight was
    // measured: it pushes the kernel over 256 VGPR+AGPR -> 1 wave/SIMD, and the LDS store path -- 32 ds_write_b128
    // per step and CU -- becomes the limiter either way; see DESIGN.md for the LDS-DMA follow-up.)
    GemmRegs ar, br;
    const int nk = K >> 6;
    gemm_gload(ar, A, lda, m0, M, 0, tid);
    gemm_gload(br, B, ldb, n0, N, 0, tid);
    GEMM_MEMBAR();
    gemm_lstore(As[0], ar, tid);
    gemm_lstore(Bs[0], br, tid);
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt, cur ^= 1) {
        const int kn = min(kt + 1, nk - 1) * 64;            // last step re-loads a valid tile that is never used
        gemm_gload(ar, A, lda, m0, M, kn, tid);
        gemm_gload(br, B, ldb, n0, N, kn, tid);
        GEMM_MEMBAR();
        const bf16_t* ap = As[cur] + (wm * 64 + r) * GBP + 8 * hh;
        const bf16_t* bp = Bs[cur] + (wn * 64 + r) * GBP + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ap + 16 * ks);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ap + 32 * GBP + 16 * ks);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bp + 16 * ks);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bp + 32 * GBP + 16 * ks);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        GEMM_MEMBAR();
        gemm_lstore(As[cur ^ 1], ar, tid);
        gemm_lstore(Bs[cur ^ 1], br, tid);
        __syncthreads();
    }
    // acc[mt][nt][reg] = Y[m0 + 64 wm + 32 mt + krow(reg,hh)][n0 + 64 wn + 32 nt + r]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = n0 + wn * 64 + nt * 32 + r;
        if (col < N) {
            const float bv = BIAS ? (float)bias[col] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = m0 + wm * 64 + mt * 32 + acr_krow(reg, hh);
                    if (row < M) {
                        float y = acc[mt][nt][reg] + bv;
                        if (RESID) y += (float)R[(int64_t)row * ldr + col];
                        Y[(int64_t)row * ldy + col] = (bf16_t)y;
                    }
                }
            }
        }
    }
}

extern "C" int acr_linear_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, const void* bias,
                               const void* resid, int64_t ldr, void* y, int64_t ldy, int32_t M, int32_t N, int32_t K,
                               void* stream) {
    ACR_CHECK_ARG(a && b && y, "acr_linear_bf16: null pointer");
    ACR_CHECK_ARG(M > 0 && N > 0 && K >= 64 && (K % 64) == 0, "acr_linear_bf16: need K %% 64 == 0 (M=%d N=%d K=%d)", M, N, K);
    ACR_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && lda >= K && ldb >= K && ldy >= N && (!resid || ldr >= N),
                  "acr_linear_bf16: leading dimensions must cover the rows and be multiples of 8 elements");
    ACR_CHECK_ARG(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0, "acr_linear_bf16: operands must be 16-byte aligned");
    const int64_t tiles = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
    ACR_CHECK_ARG(tiles < (1ll << 31), "acr_linear_bf16: grid too large");
    const dim3 grid((unsigned)tiles);
    hipStream_t st = (hipStream_t)stream;
#define ACR_GEMM_LAUNCH(BI, RE)                                                                                       \
    hipLaunchKernelGGL((gemm_nt_bf16_kernel<BI, RE>), grid, dim3(256), 0, st, (const bf16_t*)a, lda, (const bf16_t*)b, \
                       ldb, (const bf16_t*)bias, (const bf16_t*)resid, ldr, (bf16_t*)y, ldy, M, N, K)
    if (bias && resid) ACR_GEMM_LAUNCH(true, true);
    else if (bias) ACR_GEMM_LAUNCH(true, false);
    else if (resid) ACR_GEMM_LAUNCH(false, true);
    else ACR_GEMM_LAUNCH(false, false);
#undef ACR_GEMM_LAUNCH
    return acr_check_launch("acr_linear_bf16");
}
