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
#include <stdlib.h>

#include "acr_common.h"
#include <type_traits>

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


// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant: tiles go global -> LDS directly (global_load_lds_dwordx4, no staging VGPRs, no ds_write path).
// A DMA wave-instruction writes 64 lanes x 16 B = 1 KiB linearly, i.e. 8 unpadded 128-byte rows; bank conflicts of
// the ds_read_b128 fragment reads are removed by an XOR swizzle applied on the SOURCE address (lane fetches logical
// chunk  p ^ ((row >> 1) & 7)  into physical chunk p) and mirrored on the read -- rows r and r+1 sit in different
// halves of the 256-byte bank row, so the 16 rows of any ds_read_b128 lane group land on 16 distinct 16-byte slots.
// ---------------------------------------------------------------------------------------------------------------
#define GEMM_GROUP_M 8
#define GL_TILE (128 * 64)            // elements per operand tile (unpadded)

__device__ __forceinline__ void glds_stage(bf16_t* ldsbuf, const bf16_t* g, int64_t ld, int row0, int nrows, int k0,
                                           int wave, int lane) {
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rbase = (wave * 4 + i) * 8;
        const int row = rbase + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        const bf16_t* src = g + (int64_t)min(row0 + row, nrows - 1) * ld + k0 + lc * 8;
        __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(ldsbuf + rbase * 64), 16, 0, 0);
    }
}
__device__ __forceinline__ bf16x8 glds_frag(const bf16_t* ldsbuf, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(ldsbuf + row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3));
}

template <bool BIAS, bool RESID>
__global__ __launch_bounds__(256) void gemm_nt_bf16_dma_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                               const bf16_t* __restrict__ B, int64_t ldb,
                                                               const bf16_t* __restrict__ bias,
                                                               const bf16_t* __restrict__ R, int64_t ldr,
                                                               bf16_t* __restrict__ Y, int64_t ldy, int M, int N, int K) {
    __shared__ __attribute__((aligned(1024))) bf16_t smem[4 * GL_TILE];      // [A0 | B0 | A1 | B1], ONE array
    // Tile order: XCD-contiguous chunks, and inside a chunk "grouped" rasterisation (bands of GEMM_GROUP_M row panels
    // walked column by column) so a band's A panels (8 x 196 KB at K = 768) stay L2-resident while each 128-column
    // weight tile is fetched once per band instead of once per row panel (N = 3072: the weight alone is 4.7 MB > L2).
    const int ntn = (N + 127) >> 7, ntm = (M + 127) >> 7;
    const int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id / ntn, tn = id % ntn;
    (void)ntm;
    const int m0 = tm * 128, n0 = tn * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int nk = K >> 6;
    glds_stage(smem, A, lda, m0, M, 0, wave, lane);
    glds_stage(smem + GL_TILE, B, ldb, n0, N, 0, wave, lane);
    acr_dma_barrier();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt, cur ^= 1) {
        if (kt + 1 < nk) {
            glds_stage(smem + (cur ^ 1) * 2 * GL_TILE, A, lda, m0, M, (kt + 1) * 64, wave, lane);
            glds_stage(smem + (cur ^ 1) * 2 * GL_TILE + GL_TILE, B, ldb, n0, N, (kt + 1) * 64, wave, lane);
        }
        const bf16_t* as = smem + cur * 2 * GL_TILE;
        const bf16_t* bs = as + GL_TILE;
        const int ra = wm * 64 + r, rb = wn * 64 + r;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = 2 * ks + hh;
            const bf16x8 a0 = glds_frag(as, ra, c), a1 = glds_frag(as, ra + 32, c);
            const bf16x8 b0 = glds_frag(bs, rb, c), b1 = glds_frag(bs, rb + 32, c);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        acr_dma_barrier();                                    // drains the DMA (vmcnt) and frees buffer `cur`
    }
    // Epilogue through LDS (the operand buffers are free after the last barrier): each wave parks its 64x64 fp32
    // tile in its own 16 KiB, then re-reads it row-major so that 8 lanes store one full 128-byte row segment with
    // 16-byte stores (a 2-byte-per-lane store tail is store-issue bound; row-scattered 8-byte stores measured worse).
    float* stile = reinterpret_cast<float*>(smem) + wave * 4096;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                stile[(mt * 32 + acr_krow(reg, hh)) * 64 + nt * 32 + r] = acc[mt][nt][reg];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = lane + 64 * i;
        const int lrow = idx >> 3, c8 = (idx & 7) * 8;
        const int row = m0 + wm * 64 + lrow, col = n0 + wn * 64 + c8;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stile + lrow * 64 + c8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(stile + lrow * 64 + c8 + 4);
        if (row < M && col < N) {
            float y[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if (BIAS) {
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(bias + col);
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] += (float)bv[e];
            }
            if (RESID) {
                const bf16x8 rv = *reinterpret_cast<const bf16x8*>(R + (int64_t)row * ldr + col);
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] += (float)rv[e];
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)y[e];
            *reinterpret_cast<bf16x8*>(Y + (int64_t)row * ldy + col) = o;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// 256x256 tile variant: 4 waves (2 x 2), each owning a 128x128 accumulator (16 MFMA tiles, 256 accumulator registers).
// The 128x128 kernel above gives every wave a 64x64 tile (4 fragment reads per 4 MFMAs) and relies on a second
// workgroup per CU to hide its DMA waits.  Here: 8 fragment reads per 16 MFMAs, ONE workgroup per CU, and everything is
// hidden inside the wave's own instruction stream (same recipe as gemm_tn_bf16_big_kernel below, where it is derived):
//  * ring of FOUR 32-deep K stages (32 KiB each), filled three stages ahead, counted vmcnt waits;
//  * the 8 DMA instructions of a stage are spread one per 4 MFMAs (a burst parks the in-order wave in the memory
//    pipe's issue queue while the matrix pipe drains);
//  * fragment reads are inline asm with counted lgkmcnt waits, one 16-deep slice ahead of the MFMAs; the asm outputs
//    are early-clobber: the reads land asynchronously, so an output tuple must never share a register with the address
//    operand of a later read in the same statement;
//  * the barrier that publishes stage kt+1 sits between the two slices of stage kt.
// LDS rows are 64 bytes (4 x 16-byte chunks); chunk c of row r sits at physical chunk c ^ ((r >> 2) & 3), so the 16
// rows of a ds_read_b128 lane group cover 16 distinct 16-byte slots of the 256-byte bank row.
// ---------------------------------------------------------------------------------------------------------------
#define GB_BN 256
#define GB_BK 32

// MT = 32-row A fragments per wave: the workgroup tile is (64 MT) x 256, MT = 4 -> 256 rows, MT = 5 -> 320 rows.
// Why 320: with M = 25 120 tokens and N = 768 (proj, fc2 forward, qkv / proj input gradients) a 256-row tile gives
// 99 x 3 = 297 workgroups -- 1.16 waves of the 256 CUs, i.e. two rounds with the second 16 % full -- while 320 rows give
// 79 x 3 = 237 workgroups: ONE round on 93 % of the CUs.  The host picks MT so that the tile count fits one round when
// it can (gb_pick_mt).  LDS: 4 stages x (64 MT + 256) x 64 B = 128 KiB (MT 4) / 144 KiB (MT 5).
template <bool BIAS, bool RESID, int MT>
__global__ __launch_bounds__(256) void gemm_nt_bf16_big_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                               const bf16_t* __restrict__ B, int64_t ldb,
                                                               const bf16_t* __restrict__ bias,
                                                               const bf16_t* __restrict__ R, int64_t ldr,
                                                               bf16_t* __restrict__ Y, int64_t ldy, int M, int N, int K) {
    constexpr int BM = 64 * MT;
    constexpr int STAGE = (BM + GB_BN) * GB_BK;            // elements per stage: A tile (BM x 32) then B tile (256 x 32)
    constexpr int ND = MT + 4;                             // DMA instructions per stage and thread: MT for A, 4 for B
    __shared__ __attribute__((aligned(1024))) bf16_t smem[4 * STAGE];
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    const int ntn = (N + GB_BN - 1) / GB_BN;
    const int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id / ntn, tn = id % ntn;
    const int m0 = tm * BM, n0 = tn * GB_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int nk = K / GB_BK;
    // DMA instruction q of a stage: q < MT: A rows (wave * MT + q) * 16 .. +15; q >= MT: B rows (wave * 4 + q - MT) * 16 ..
    uint32_t doff[ND];
#pragma unroll
    for (int q = 0; q < ND; ++q) {
        const int row = (q < MT ? wave * MT + q : wave * 4 + q - MT) * 16 + (lane >> 2);
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        doff[q] = (q < MT) ? (uint32_t)(((int64_t)min(m0 + row, M - 1) * lda + lc * 8) * 2)
                           : (uint32_t)(((int64_t)min(n0 + row, N - 1) * ldb + lc * 8) * 2);
    }
    auto dma = [&](int kt, int q) {
        const char* base = (q < MT) ? reinterpret_cast<const char*>(A + kt * GB_BK) : reinterpret_cast<const char*>(B + kt * GB_BK);
        bf16_t* dst = smem + (kt & 3) * STAGE +
                      ((q < MT) ? (wave * MT + q) * 16 * GB_BK : BM * GB_BK + (wave * 4 + q - MT) * 16 * GB_BK);
        __builtin_amdgcn_global_load_lds((glb_vp)(base + doff[q]), (lds_vp)dst, 16, 0, 0);
    };
    f32x16 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 a0[MT], b0[4], a1[MT], b1[4];
    const int ra = wm * (32 * MT) + r, rb = wn * 128 + r;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)smem;
    const uint32_t swa = (ra >> 2) & 3, swb = (rb >> 2) & 3;
    // byte address of this lane's fragment row for K slice ks: row * 64 + ((2 ks + hh) ^ sw) * 16; +32 rows = +2048 B
    const uint32_t oa0 = lds0 + ra * 64 + ((hh ^ swa) << 4), oa1 = lds0 + ra * 64 + (((2 + hh) ^ swa) << 4);
    const uint32_t ob0 = lds0 + BM * GB_BK * 2 + rb * 64 + ((hh ^ swb) << 4);
    const uint32_t ob1 = lds0 + BM * GB_BK * 2 + rb * 64 + (((2 + hh) ^ swb) << 4);
    // fragment reads and their waits are inline asm (see the header of this section); outputs early-clobber
#define GB_RD1(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "i"(OFF))
#define GB_READA(dst, addr)                                                                                          \
    { GB_RD1(dst[0], addr, 0); GB_RD1(dst[1], addr, 2048); GB_RD1(dst[2], addr, 4096); GB_RD1(dst[3], addr, 6144);    \
      if (MT == 5) GB_RD1(dst[MT - 1], addr, 8192); }
#define GB_READB(dst, addr)                                                                                          \
    { GB_RD1(dst[0], addr, 0); GB_RD1(dst[1], addr, 2048); GB_RD1(dst[2], addr, 4096); GB_RD1(dst[3], addr, 6144); }
    // wait until at most `cnt` LDS reads are outstanding, tied to the fragments that must have landed
#define GB_WAITF(cnt, x, y)                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                       \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[MT - 1]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]))
#define GB_MFMAS(x, y)                                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[i_], y[j_], acc[i_][j_], 0, 0, 0)
    // MODE 0: steady state (issues stage kt+3); 1: a later stage exists but nothing left to issue; 2: last stage.  The
    // loop is peeled so that each body is straight-line code: the scheduler groups can interleave it, and the
    // accumulators never meet at a control-flow merge (a merge made the register allocator spill them).
    auto body = [&](int kt, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const uint32_t so = (uint32_t)(kt & 3) * (STAGE * 2), so2 = (uint32_t)((kt + 1) & 3) * (STAGE * 2);
        GB_READA(a1, oa1 + so);                             // slice 1 of stage kt
        GB_READB(b1, ob1 + so);
        if (MT == 4) { GB_WAITF(8, a0, b0); } else { GB_WAITF(9, a0, b0); }
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < MT; ++q) dma(kt + 3, q);    // A part of stage kt+3, spread over this slice's MFMAs
        }
        GB_MFMAS(a0, b0);
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < MT; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        GB_WAITF(0, a1, b1);                                // this wave holds every fragment of stage kt
        if (MODE <= 1) {
            // stage kt+1 must have landed; stage kt+2 (ND) and the A part of stage kt+3 (MT) may stay in flight
            if (MODE == 0) {
                if (MT == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_barrier" ::: "memory");
            GB_READA(a0, oa0 + so2);                        // slice 0 of stage kt+1
            GB_READB(b0, ob0 + so2);
        }
        if (MODE == 0) {
#pragma unroll
            for (int q = MT; q < ND; ++q) dma(kt + 3, q);   // B part
        }
        GB_MFMAS(a1, b1);
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
    };
#pragma unroll
    for (int q = 0; q < ND; ++q) dma(0, q);
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < ND; ++q) dma(1, q);
    }
    if (nk > 2) {
#pragma unroll
        for (int q = 0; q < ND; ++q) dma(2, q);
    }
    if (nk > 2) { if (MT == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); }
    else if (nk > 1) { if (MT == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    GB_READA(a0, oa0);
    GB_READB(b0, ob0);
    {
        int kt = 0;
        for (; kt + 3 < nk; ++kt) body(kt, std::integral_constant<int, 0>{});
        for (; kt + 1 < nk; ++kt) body(kt, std::integral_constant<int, 1>{});
        body(kt, std::integral_constant<int, 2>{});
    }
#undef GB_RD1
#undef GB_READA
#undef GB_READB
#undef GB_WAITF
#undef GB_MFMAS
    __syncthreads();                                        // all fragment reads done before LDS is reused below
    // Epilogue through LDS, 32 rows x 64 columns of the wave's tile at a time (wave-private 8 KiB), 16-byte stores
    float* stile = reinterpret_cast<float*>(smem) + wave * 2048;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int qn = 0; qn < 2; ++qn) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) stile[acr_krow(reg, hh) * 64 + nt * 32 + r] = acc[t][qn * 2 + nt][reg];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = lane + 64 * i;
                const int lrow = idx >> 3, c8 = (idx & 7) * 8;
                const int row = m0 + wm * (32 * MT) + t * 32 + lrow, col = n0 + wn * 128 + qn * 64 + c8;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stile + lrow * 64 + c8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stile + lrow * 64 + c8 + 4);
                if (row < M && col < N) {
                    float y[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if (BIAS) {
                        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(bias + col);
#pragma unroll
                        for (int e = 0; e < 8; ++e) y[e] += (float)bv[e];
                    }
                    if (RESID) {
                        const bf16x8 rv = *reinterpret_cast<const bf16x8*>(R + (int64_t)row * ldr + col);
#pragma unroll
                        for (int e = 0; e < 8; ++e) y[e] += (float)rv[e];
                    }
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)y[e];
                    *reinterpret_cast<bf16x8*>(Y + (int64_t)row * ldy + col) = o;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// 320x256 tile, EIGHT waves (2 x 4, each 160x64: 10 accumulators = 160 registers, two waves per SIMD).  Made for the
// N = 768 GEMMs of the block (proj / fc2 forward, qkv / proj input gradients) at M = 25 120 tokens: 79 x 3 = 237
// workgroups = ONE round on 93 % of the CUs, where 128x128 tiles need 2.3 rounds of two-per-CU slots and 256x256 tiles
// 1.16 rounds.  Same ring recipe as the 256x256 kernel (four 32-deep stages, counted vmcnt, asm fragment reads one
// slice ahead, DMA spread over the MFMAs, barrier between the slices); two waves per SIMD additionally overlap each
// other's waits.  A stage is 36 DMA instructions (20 A + 16 B row groups); every wave issues 5, the last four slots
// re-load B groups 12..15 (same bytes to the same LDS addresses) so that the counted waits are uniform.
// ---------------------------------------------------------------------------------------------------------------
#define GW_BM 320
#define GW_STAGE ((GW_BM + GB_BN) * GB_BK)         // elements per stage = 36 KiB

// EPI 0: y = acc (+bias) (+resid).  EPI 1 (fc1 forward, models/vision_transformer.py:160-161): with h = bf16(acc + bias):
// AUX = GELU(h) (exact erf form, computed from the bf16-rounded h like the unfused pair of kernels) and Y = GELU'(h) -- the
// derivative fc2's backward needs, stored in place of h (which nothing else reads; Phi and the density are shared).  EPI 2
// (input gradient of fc2 through the GELU): y = acc * AUX with AUX = the saved GELU'(h).
// Phi(x) = 0.5 (1 + erf(x / sqrt 2)) by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far inside bf16's 2^-9), sharing
// exp(-x^2/2) with the density term of GELU': one exp, one rcp and five FMAs per element.  With libm's erff + expf the
// epilogues cost 70-90 us per GEMM -- as much as the separate GELU kernels they replace.
__device__ __forceinline__ void gw_phi(float x, float& cdf, float& e) {
    const float ax = fabsf(x);
    e = __expf(-0.5f * x * x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erf_abs = 1.f - p * t * e;                  // erf(|x| / sqrt 2)
    cdf = 0.5f * (1.f + copysignf(erf_abs, x));
}

template <bool BIAS, bool RESID, int EPI = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16_wide_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                   const bf16_t* __restrict__ B, int64_t ldb,
                                                                   const bf16_t* __restrict__ bias,
                                                                   const bf16_t* __restrict__ R, int64_t ldr,
                                                                   bf16_t* __restrict__ Y, int64_t ldy, int M, int N, int K,
                                                                   bf16_t* __restrict__ AUX = nullptr, int64_t ldaux = 0) {
    __shared__ __attribute__((aligned(1024))) bf16_t smem[4 * GW_STAGE];
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    const int ntn = (N + GB_BN - 1) / GB_BN;
    const int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id / ntn, tn = id % ntn;
    const int m0 = tm * GW_BM, n0 = tn * GB_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, hh = lane >> 5;
    const int nk = K / GB_BK;
    // DMA slot j = wave * 5 + q: j < 20 -> A row group j; 20 <= j < 36 -> B row group j - 20; j >= 36 -> B group j - 24 again
    uint32_t doff[5];
    int dgrp[5];                                            // LDS row group inside the stage (A groups 0..19, B groups 20..35)
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        int j = wave * 5 + q;
        if (j >= 36) j -= 4;
        dgrp[q] = j;
        const bool isa = j < 20;
        const int row = (isa ? j : j - 20) * 16 + (lane >> 2);
        const int lc = (lane & 3) ^ ((row >> 2) & 3);
        doff[q] = isa ? (uint32_t)(((int64_t)min(m0 + row, M - 1) * lda + lc * 8) * 2)
                      : (uint32_t)(((int64_t)min(n0 + row, N - 1) * ldb + lc * 8) * 2);
    }
    auto dma = [&](int kt, int q) {
        const char* base = (dgrp[q] < 20) ? reinterpret_cast<const char*>(A + kt * GB_BK) : reinterpret_cast<const char*>(B + kt * GB_BK);
        bf16_t* dst = smem + (kt & 3) * GW_STAGE + dgrp[q] * 16 * GB_BK;         // A tile then B tile: group g at g * 16 rows
        __builtin_amdgcn_global_load_lds((glb_vp)(base + doff[q]), (lds_vp)dst, 16, 0, 0);
    };
    f32x16 acc[5][2];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 a0[5], b0[2], a1[5], b1[2];
    const int ra = wm * 160 + r, rb = wn * 64 + r;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)smem;
    const uint32_t swa = (ra >> 2) & 3, swb = (rb >> 2) & 3;
    const uint32_t oa0 = lds0 + ra * 64 + ((hh ^ swa) << 4), oa1 = lds0 + ra * 64 + (((2 + hh) ^ swa) << 4);
    const uint32_t ob0 = lds0 + GW_BM * GB_BK * 2 + rb * 64 + ((hh ^ swb) << 4);
    const uint32_t ob1 = lds0 + GW_BM * GB_BK * 2 + rb * 64 + (((2 + hh) ^ swb) << 4);
#define GW_RD1(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "i"(OFF))
#define GW_READ7(xa, xb, aa, ab)                                                                                     \
    { GW_RD1(xa[0], aa, 0); GW_RD1(xa[1], aa, 2048); GW_RD1(xa[2], aa, 4096); GW_RD1(xa[3], aa, 6144);                \
      GW_RD1(xa[4], aa, 8192); GW_RD1(xb[0], ab, 0); GW_RD1(xb[1], ab, 2048); }
#define GW_WAIT(cnt, x, y)                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                       \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(y[0]), "+v"(y[1]))
#define GW_MFMA10(x, y)                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 5; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                 \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[i_], y[j_], acc[i_][j_], 0, 0, 0)
    auto body = [&](int kt, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const uint32_t so = (uint32_t)(kt & 3) * (GW_STAGE * 2), so2 = (uint32_t)((kt + 1) & 3) * (GW_STAGE * 2);
        GW_READ7(a1, b1, oa1 + so, ob1 + so);               // slice 1 of stage kt
        GW_WAIT(7, a0, b0);
        if (MODE == 0) { dma(kt + 3, 0); dma(kt + 3, 1); dma(kt + 3, 2); }
        GW_MFMA10(a0, b0);
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        GW_WAIT(0, a1, b1);                                 // this wave holds every fragment of stage kt
        if (MODE <= 1) {
            // stage kt+1 must have landed; stage kt+2 (5) and the first three DMAs of stage kt+3 may stay in flight
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            GW_READ7(a0, b0, oa0 + so2, ob0 + so2);         // slice 0 of stage kt+1
        }
        if (MODE == 0) { dma(kt + 3, 3); dma(kt + 3, 4); }
        GW_MFMA10(a1, b1);
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
    };
#pragma unroll
    for (int q = 0; q < 5; ++q) dma(0, q);
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < 5; ++q) dma(1, q);
    }
    if (nk > 2) {
#pragma unroll
        for (int q = 0; q < 5; ++q) dma(2, q);
    }
    if (nk > 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    GW_READ7(a0, b0, oa0, ob0);
    {
        int kt = 0;
        for (; kt + 3 < nk; ++kt) body(kt, std::integral_constant<int, 0>{});
        for (; kt + 1 < nk; ++kt) body(kt, std::integral_constant<int, 1>{});
        body(kt, std::integral_constant<int, 2>{});
    }
#undef GW_RD1
#undef GW_READ7
#undef GW_WAIT
#undef GW_MFMA10
    __syncthreads();                                        // all fragment reads done before LDS is reused below
    // Epilogue through LDS, 32 rows x 64 columns at a time (wave-private 8 KiB), 16-byte stores
    float* stile = reinterpret_cast<float*>(smem) + wave * 2048;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) stile[acr_krow(reg, hh) * 64 + nt * 32 + r] = acc[t][nt][reg];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i;
            const int lrow = idx >> 3, c8 = (idx & 7) * 8;
            const int row = m0 + wm * 160 + t * 32 + lrow, col = n0 + wn * 64 + c8;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(stile + lrow * 64 + c8);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(stile + lrow * 64 + c8 + 4);
            if (row < M && col < N) {
                float y[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (BIAS) {
                    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(bias + col);
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] += (float)bv[e];
                }
                if (RESID) {
                    const bf16x8 rv = *reinterpret_cast<const bf16x8*>(R + (int64_t)row * ldr + col);
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] += (float)rv[e];
                }
                bf16x8 o;
                if (EPI == 2) {
                    const bf16x8 hv = *reinterpret_cast<const bf16x8*>(AUX + (int64_t)row * ldaux + col);
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] *= (float)hv[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)y[e];
                if (EPI == 1) {
                    bf16x8 a8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float hx = (float)o[e];
                        float cdf, ex;
                        gw_phi(hx, cdf, ex);
                        a8[e] = (bf16_t)(hx * cdf);
                        o[e] = (bf16_t)fmaf(hx * ex, 0.39894228040143267794f, cdf);
                    }
                    *reinterpret_cast<bf16x8*>(AUX + (int64_t)row * ldaux + col) = a8;
                }
                *reinterpret_cast<bf16x8*>(Y + (int64_t)row * ldy + col) = o;
            }
        }
    }
}

// MT = 5 (320-row tiles, one round of the CUs for the N = 768 shapes) is not instantiated: its 160x128 wave tile needs
// 320 accumulator registers, more than the 256 AGPRs, and hipcc spills ~450 registers instead of keeping the rest of the
// accumulators in arch VGPRs (measured: 512 VGPR, 464 spills).  Fixing the quantisation of these shapes needs stream-K.
static int gb_pick_mt(int M, int N) { (void)M; (void)N; return 4; }

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
    const int mt = gb_pick_mt(M, N);
    // 320x256 tiles (8 waves) whenever they give the chip enough workgroups: at M = 25 120 tokens they take 36 / 86 / 105
    // / 120 / 142 us on proj, qkv-dX, qkv, fc2, fc1 (820-1030 TF) against 43 / 112 / 126 / 156 / 189 us for the 128x128
    // kernel; the N = 768 shapes fit ONE round of the CUs (79 x 3 = 237 workgroups).  Small problems keep 128x128 tiles.
    const int64_t tilesw = (int64_t)((M + GW_BM - 1) / GW_BM) * ((N + GB_BN - 1) / GB_BN);
    const bool env_nowide = acr_opt(ACR_OPT_GEMM_NOWIDE) != 0;
    const bool wide_ok = !env_nowide && tilesw >= 200;
    const dim3 gridw((unsigned)tilesw);
    const dim3 grid3((unsigned)(((M + 64 * mt - 1) / (64 * mt)) * ((N + GB_BN - 1) / GB_BN)));
    hipStream_t st = (hipStream_t)stream;
    const int env_variant = acr_opt(ACR_OPT_GEMM_VARIANT);   // 2: 128x128x64 2-stage, 3: 256x256x32 4-stage
    // LDS-DMA kernel stores 8-column (16-byte) groups: needs N, ldy, ldr multiples of 8 and 16-byte aligned y/bias/resid
    const bool vec_ok = (N % 8) == 0 && (ldy % 8) == 0 && (ldr % 8) == 0 && ((uintptr_t)y & 15) == 0 &&
                        ((uintptr_t)bias & 15) == 0 && ((uintptr_t)resid & 15) == 0;
    const bool env_regstage = acr_opt(ACR_OPT_GEMM_REGSTAGE) != 0;       // A/B switch for the older variant
    const bool use_regstage = env_regstage || !vec_ok;
#define ACR_GEMM_LAUNCH(BI, RE)                                                                                       \
    if (use_regstage)                                                                                                 \
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<BI, RE>), grid, dim3(256), 0, st, (const bf16_t*)a, lda,               \
                           (const bf16_t*)b, ldb, (const bf16_t*)bias, (const bf16_t*)resid, ldr, (bf16_t*)y, ldy, M, N, K); \
    else if (env_variant == 4 || (env_variant == 2 && wide_ok))                                                       \
        hipLaunchKernelGGL((gemm_nt_bf16_wide_kernel<BI, RE>), gridw, dim3(512), 0, st, (const bf16_t*)a, lda,         \
                           (const bf16_t*)b, ldb, (const bf16_t*)bias, (const bf16_t*)resid, ldr, (bf16_t*)y, ldy, M, N, K); \
    else if (env_variant == 3) {                                                                                      \
            hipLaunchKernelGGL((gemm_nt_bf16_big_kernel<BI, RE, 4>), grid3, dim3(256), 0, st, (const bf16_t*)a, lda,   \
                               (const bf16_t*)b, ldb, (const bf16_t*)bias, (const bf16_t*)resid, ldr, (bf16_t*)y, ldy, M, N, K); \
    } else                                                                                                            \
        hipLaunchKernelGGL((gemm_nt_bf16_dma_kernel<BI, RE>), grid, dim3(256), 0, st, (const bf16_t*)a, lda,           \
                           (const bf16_t*)b, ldb, (const bf16_t*)bias, (const bf16_t*)resid, ldr, (bf16_t*)y, ldy, M, N, K)
    if (bias && resid) ACR_GEMM_LAUNCH(true, true);
    else if (bias) ACR_GEMM_LAUNCH(true, false);
    else if (resid) ACR_GEMM_LAUNCH(false, true);
    else ACR_GEMM_LAUNCH(false, false);
#undef ACR_GEMM_LAUNCH
    return acr_check_launch("acr_linear_bf16");
}

static int gw_check(const char* who, const void* a, int64_t lda, const void* b, int64_t ldb, int64_t ldy, int64_t ldaux, int M,
                    int N, int K) {
    ACR_CHECK_ARG(M > 0 && N > 0 && K >= 64 && (K % 64) == 0 && (N % 8) == 0, "%s: need K %% 64 == 0 and N %% 8 == 0 (M=%d N=%d K=%d)", who, M, N, K);
    ACR_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && (ldy % 8) == 0 && (ldaux % 8) == 0 && lda >= K && ldb >= K && ldy >= N && ldaux >= N,
                  "%s: leading dimensions must cover the rows and be multiples of 8 elements", who);
    ACR_CHECK_ARG(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0, "%s: operands must be 16-byte aligned", who);
    ACR_CHECK_ARG((int64_t)((M + GW_BM - 1) / GW_BM) * ((N + GB_BN - 1) / GB_BN) < (1ll << 31), "%s: grid too large", who);
    return ACR_OK;
}

extern "C" int acr_linear_gelu_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, const void* bias, void* h,
                                    void* act, int64_t ldy, int32_t M, int32_t N, int32_t K, void* stream) {
    ACR_CHECK_ARG(a && b && bias && h && act, "acr_linear_gelu_bf16: null pointer");
    int rc = gw_check("acr_linear_gelu_bf16", a, lda, b, ldb, ldy, ldy, M, N, K);
    if (rc) return rc;
    ACR_CHECK_ARG(((uintptr_t)h & 15) == 0 && ((uintptr_t)act & 15) == 0 && ((uintptr_t)bias & 15) == 0, "acr_linear_gelu_bf16: 16-byte alignment");
    const dim3 grid((unsigned)(((M + GW_BM - 1) / GW_BM) * ((N + GB_BN - 1) / GB_BN)));
    hipLaunchKernelGGL((gemm_nt_bf16_wide_kernel<true, false, 1>), grid, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)a, lda,
                       (const bf16_t*)b, ldb, (const bf16_t*)bias, (const bf16_t*)nullptr, (int64_t)0, (bf16_t*)h, ldy, M, N, K,
                       (bf16_t*)act, ldy);
    return acr_check_launch("acr_linear_gelu_bf16");
}

extern "C" int acr_linear_dgelu_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, const void* h, int64_t ldh,
                                     void* y, int64_t ldy, int32_t M, int32_t N, int32_t K, void* stream) {
    ACR_CHECK_ARG(a && b && h && y, "acr_linear_dgelu_bf16: null pointer");
    int rc = gw_check("acr_linear_dgelu_bf16", a, lda, b, ldb, ldy, ldh, M, N, K);
    if (rc) return rc;
    ACR_CHECK_ARG(((uintptr_t)h & 15) == 0 && ((uintptr_t)y & 15) == 0, "acr_linear_dgelu_bf16: 16-byte alignment");
    const dim3 grid((unsigned)(((M + GW_BM - 1) / GW_BM) * ((N + GB_BN - 1) / GB_BN)));
    hipLaunchKernelGGL((gemm_nt_bf16_wide_kernel<false, false, 2>), grid, dim3(512), 0, (hipStream_t)stream, (const bf16_t*)a, lda,
                       (const bf16_t*)b, ldb, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (int64_t)0, (bf16_t*)y, ldy, M, N, K,
                       (bf16_t*)const_cast<void*>(h), ldh);
    return acr_check_launch("acr_linear_dgelu_bf16");
}

// ---------------------------------------------------------------------------------------------------------------
// Bias gradient of the projections: out[n] = sum_m dy[m, n] (fp32 accumulate, bf16 out), two deterministic stages.
// The stock reduction runs this tall-matrix column sum at ~1.3 TB/s (30 us for 25120 x 768); here each workgroup
// streams a 256-row slab with 16-byte loads (8 columns per lane) and the slab partials are summed in slab order.
// ---------------------------------------------------------------------------------------------------------------
#define CS_ROWS 128
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ dy, int64_t ld, int M, int N,
                                                             float* __restrict__ part) {
    const int nvec = N >> 3;                                // 8-column vectors per row
    const int slab = blockIdx.y, v = blockIdx.x * 64 + (threadIdx.x & 63), rgrp = threadIdx.x >> 6;
    __shared__ float sh[4][64][8];
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (v < nvec) {
        const int r0 = slab * CS_ROWS, r1 = min(r0 + CS_ROWS, M);
        for (int r = r0 + rgrp; r < r1; r += 4) {
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(dy + (int64_t)r * ld + v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)x[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[rgrp][threadIdx.x & 63][e] = acc[e];
    __syncthreads();
    if (rgrp == 0 && v < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            part[(int64_t)slab * N + v * 8 + e] = sh[0][threadIdx.x][e] + sh[1][threadIdx.x][e] + sh[2][threadIdx.x][e] + sh[3][threadIdx.x][e];
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nslab, int N,
                                                           bf16_t* __restrict__ out) {
    // 16 columns x 16 slab groups per block; groups sum their slabs in order, then a fixed-order combine through LDS
    __shared__ float sh[16][17];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + c;
    float s = 0.f;
    if (n < N)
        for (int i = g; i < nslab; i += 16) s += part[(int64_t)i * N + n];
    sh[g][c] = s;
    __syncthreads();
    if (g == 0 && n < N) {
        float t = sh[0][c];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += sh[k][c];
        out[n] = (bf16_t)t;
    }
}

extern "C" size_t acr_colsum_ws_floats(int32_t M, int32_t N) { return (size_t)((M + CS_ROWS - 1) / CS_ROWS) * (size_t)N; }

extern "C" int acr_colsum_bf16(const void* dy, int64_t ld, int32_t M, int32_t N, float* ws, void* out, void* stream) {
    ACR_CHECK_ARG(dy && ws && out, "acr_colsum_bf16: null pointer");
    ACR_CHECK_ARG(M > 0 && N > 0 && (N % 8) == 0 && (ld % 8) == 0 && ld >= N && ((uintptr_t)dy & 15) == 0,
                  "acr_colsum_bf16: need N %% 8 == 0, ld %% 8 == 0 and a 16-byte aligned matrix (M=%d N=%d)", M, N);
    const int nslab = (M + CS_ROWS - 1) / CS_ROWS;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N / 8 + 63) / 64, nslab), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dy, ld, M, N, ws);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const float*)ws,
                       nslab, N, (bf16_t*)out);
    return acr_check_launch("acr_colsum_bf16");
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the projections ("TN"): dW[N,K] = dY[M,N]^T . X[M,K], contraction over the M = B*T tokens.
// Both operands are stored with the contraction index as the ROW, so both MFMA fragments come from the hardware
// transpose read (ds_read_b64_tr_b16) of row-major [64 m][128 n|k] LDS tiles filled by LDS-DMA; 256-byte rows would
// put the 4 rows of a transpose read on the same banks, so 16-byte chunk c of row m is stored at chunk
// c ^ ((m & 3) << 2) (source-side swizzle, mirrored on the read) -> the four 64-byte row segments of a 32-lane half
// cover all 64 banks.  The output is small (N*K) and the contraction long, so the M range is split over S workgroups
// per 128x128 tile (S chosen to fill ~2 workgroups per CU); fp32 partial slabs are summed in split order by a second
// kernel -- deterministic, no float atomics.
// ---------------------------------------------------------------------------------------------------------------
#define WG_TILE (64 * 128)            // elements per operand tile

__device__ __forceinline__ void wg_stage(bf16_t* ldsbuf, const bf16_t* g, int64_t ld, int m0, int M, int col0,
                                         int wave, int lane) {
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rbase = (wave * 4 + i) * 4;                // 4 rows (1 KiB) per DMA instruction
        const int row = rbase + (lane >> 4);
        const int lc = (lane & 15) ^ ((row & 3) << 2);
        const bf16_t* src = g + (int64_t)min(m0 + row, M - 1) * ld + col0 + lc * 8;
        __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(ldsbuf + rbase * 128), 16, 0, 0);
    }
}
// fragment for k-step s (16 contraction rows) and 32-column block blk: element j of lane half h = tile[16s + 8(j>>2) + 4h + (j&3)][32 blk + (l&31)]
__device__ __forceinline__ bf16x8 wg_frag(const bf16_t* tile, int s, int blk, int lane) {
    const int i = lane & 15, h = lane >> 5;
    const int row = 16 * s + 4 * h + (i >> 2);
    const int col = 32 * blk + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
    const int sw = ((col >> 3) ^ ((row & 3) << 2)) << 3;     // row + 8 has the same (row & 3)
    const bf16_t* a0 = tile + row * 128 + sw + (col & 7);
    typedef __attribute__((address_space(3))) bf16x4* lds_p;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a0 + 8 * 128));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

// The same fragment through inline asm.  hipcc puts s_waitcnt vmcnt(0) in front of a ds_read_b64_tr_b16 BUILTIN that
// follows an LDS-DMA in the same block (it cannot tell the DMA's LDS write from the read), which drains the prefetch
// of the next tile before the current one is used.  The asm reads are invisible to that analysis; the caller waits with
// wg_frag_wait (lgkmcnt(0), tied to the registers) before the first MFMA.  Early-clobber outputs: the reads land
// asynchronously, an output must not share a register with the address of the second read.
__device__ __forceinline__ void wg_frag_issue(const bf16_t* tile, int s, int blk, int lane, bf16x4& lo, bf16x4& hi) {
    const int i = lane & 15, h = lane >> 5;
    const int row = 16 * s + 4 * h + (i >> 2);
    const int col = 32 * blk + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
    const int sw = ((col >> 3) ^ ((row & 3) << 2)) << 3;
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const bf16_t*)(tile + row * 128 + sw + (col & 7));
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=&v"(lo), "=&v"(hi) : "v"(addr));
}
#define WG_FRAG_WAIT8(L, H)                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                             \
                 : "+v"(L[0]), "+v"(L[1]), "+v"(L[2]), "+v"(L[3]), "+v"(L[4]), "+v"(L[5]), "+v"(L[6]), "+v"(L[7]),  \
                   "+v"(H[0]), "+v"(H[1]), "+v"(H[2]), "+v"(H[3]), "+v"(H[4]), "+v"(H[5]), "+v"(H[6]), "+v"(H[7]))

__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(const bf16_t* __restrict__ dY, int64_t ldy,
                                                           const bf16_t* __restrict__ X, int64_t ldx,
                                                           float* __restrict__ slabs, int M, int N, int K, int nsplit,
                                                           int steps_per_split) {
    __shared__ __attribute__((aligned(1024))) bf16_t smem[4 * WG_TILE];      // [A0 | B0 | A1 | B1]
    const int ntk = K >> 7, ntn = N >> 7;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int split = id % nsplit; id /= nsplit;
    const int tk = id % ntk, tn = id / ntk;
    (void)ntn;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int total_steps = (M + 63) >> 6;
    const int st0 = split * steps_per_split, st1 = min(st0 + steps_per_split, total_steps);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (st0 < st1) {
        wg_stage(smem, dY, ldy, st0 * 64, M, tn * 128, wave, lane);
        wg_stage(smem + WG_TILE, X, ldx, st0 * 64, M, tk * 128, wave, lane);
        acr_dma_barrier();
        int cur = 0;
        for (int stp = st0; stp < st1; ++stp, cur ^= 1) {
            if (stp + 1 < st1) {
                wg_stage(smem + (cur ^ 1) * 2 * WG_TILE, dY, ldy, (stp + 1) * 64, M, tn * 128, wave, lane);
                wg_stage(smem + (cur ^ 1) * 2 * WG_TILE + WG_TILE, X, ldx, (stp + 1) * 64, M, tk * 128, wave, lane);
            }
            bf16_t* as = smem + cur * 2 * WG_TILE;
            const bf16_t* bs = as + WG_TILE;
            if (stp * 64 + 64 > M) {
                // ragged last chunk: rows >= M alias row M-1 (clamped DMA); zero them in the dY tile so they add nothing
                for (int idx = tid; idx < 64 * 16; idx += 256) {
                    const int row = idx >> 4;
                    if (stp * 64 + row >= M) {
                        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                        *reinterpret_cast<bf16x8*>(as + row * 128 + (idx & 15) * 8) = z;
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 a0 = wg_frag(as, s, 2 * wi, lane), a1 = wg_frag(as, s, 2 * wi + 1, lane);
                const bf16x8 b0 = wg_frag(bs, s, 2 * wj, lane), b1 = wg_frag(bs, s, 2 * wj + 1, lane);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            }
            acr_dma_barrier();
        }
    }
    // acc[ni][kj][reg] = dW[tn*128 + 32(2wi+ni) + krow(reg,hh)][tk*128 + 32(2wj+kj) + r]
    float* slab = slabs + (int64_t)split * N * K;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int kj = 0; kj < 2; ++kj)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int n = tn * 128 + 32 * (2 * wi + ni) + acr_krow(reg, hh);
                const int kk = tk * 128 + 32 * (2 * wj + kj) + r;
                slab[(int64_t)n * K + kk] = acc[ni][kj][reg];
            }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int nsplit, int64_t nk,
                                                           bf16_t* __restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= nk) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(slabs + i);
    for (int sp = 1; sp < nsplit; ++sp) s += *reinterpret_cast<const f32x4*>(slabs + (int64_t)sp * nk + i);
    acr_store4<bf16_t>(out + i, s);
}

// ---------------------------------------------------------------------------------------------------------------
// 256x256 output tile variant of the weight-gradient GEMM: 4 waves (2 x 2), each a 128x128 accumulator.  The weight
// gradient is the ideal case for the big tile: few output tiles, a 25 120-long contraction split over ~256 workgroups
// (one per CU, ~45 steps each), so prologue/epilogue amortise and there is no tile quantisation.  Per 16-row slice a wave
// reads 8 fragments for 16 MFMAs (the 128x128-tile kernel above: 4 for 4), which takes the LDS off the critical path.
// LDS per stage: dY tile [2 halves][64 m][128 n] + X tile [2 halves][64 m][128 k] = 64 KiB, two stages; each half has
// exactly the layout of the kernel above, so wg_stage / wg_frag are shared.
// ---------------------------------------------------------------------------------------------------------------
#define WB_ROWS 32                         // contraction rows per ring stage
#define WB_HALF (WB_ROWS * 128)            // elements of one [32 m][128 cols] half tile
#define WB_STAGE (4 * WB_HALF)             // [dY half0 | dY half1 | X half0 | X half1] = 32 KiB

// Ring of FOUR 32-row stages (32 KiB each) filled three stages ahead, 8 DMA instructions per stage and thread, spread
// one per 4 MFMAs.  Three findings shaped this kernel (fc1's dW, 25120 x 3072 x 768, same launch geometry):
//  * the compiler puts s_waitcnt vmcnt(0) in front of ds_read_b64_tr_b16 *builtins* that follow an LDS-DMA in the same
//    block (it cannot disambiguate the DMA's LDS write from the read), which serialises DMA and compute: MFMA+reads
//    alone 116 us, DMA alone 61 us, together 181 us.  Fragment reads here are inline asm with counted lgkmcnt waits;
//  * a stage's DMAs issued as one burst park the in-order wave in the memory pipe's issue queue, so they are spread
//    over the step's MFMAs (1 KiB per 128 cycles per wave = the consumption rate);
//  * split-major tile order: the ~32 consecutive ids on one XCD are the tiles of ONE token range (L2 reuse).
// The barrier that publishes stage it+1 sits between the two 16-row slices of stage it, so the barrier and the first
// fragment reads of the next stage hide behind 16 MFMAs.  vmcnt bookkeeping at that barrier: stage it+2 (8) and the
// first half of stage it+3 (4) may stay in flight -> s_waitcnt vmcnt(12).
__global__ __launch_bounds__(256) void gemm_tn_bf16_big_kernel(const bf16_t* __restrict__ dY, int64_t ldy,
                                                               const bf16_t* __restrict__ X, int64_t ldx,
                                                               float* __restrict__ slabs, int M, int N, int K, int nsplit,
                                                               int steps_per_split) {
    __shared__ __attribute__((aligned(1024))) bf16_t smem[4 * WB_STAGE];
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    const int ntk = K >> 8, ntiles = (N >> 8) * ntk;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int split = id / ntiles; id -= split * ntiles;
    const int tk = id % ntk, tn = id / ntk;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int total_steps = (M + WB_ROWS - 1) / WB_ROWS;
    const int st0 = split * steps_per_split, st1 = min(st0 + steps_per_split, total_steps);
    const int n = st1 - st0;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float* slab = slabs + (int64_t)split * N * K;
    if (n > 0) {
        // DMA sources: instruction q (0..7) of a stage: operand q>>2 (dY, X), 128-column half (q>>1)&1, row group q&1
        uint32_t doff[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = (wave * 2 + (q & 1)) * 4 + (lane >> 4);
            const int lc = (lane & 15) ^ ((row & 3) << 2);
            const int col = ((q >> 2) ? tk : tn) * 256 + ((q >> 1) & 1) * 128 + lc * 8;
            doff[q] = (uint32_t)(((int64_t)row * ((q >> 2) ? ldx : ldy) + col) * 2);
        }
        auto dma = [&](int it, int q) {                      // one DMA instruction of stage `it` (all rows < M: see host)
            const int64_t m0 = (int64_t)(st0 + it) * WB_ROWS;
            const char* base = (q >> 2) ? reinterpret_cast<const char*>(X + m0 * ldx) : reinterpret_cast<const char*>(dY + m0 * ldy);
            bf16_t* dst = smem + (it & 3) * WB_STAGE + (q >> 1) * WB_HALF + (wave * 2 + (q & 1)) * 4 * 128;
            __builtin_amdgcn_global_load_lds((glb_vp)(base + doff[q]), (lds_vp)dst, 16, 0, 0);
        };
        // fragment addresses (bytes): row 16 s + 4 h + (i >> 2) of the wave's half tile, 32-column block t at
        // chunk (4 t ^ 4 x) with x = row & 3 -- see wg_frag; slice and +8-row offsets are immediates
        const int li = lane & 15, x = (li >> 2) & 3;
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)smem;
        const uint32_t lowb = (uint32_t)((4 * hh + (li >> 2)) * 256 + ((2 * ((lane >> 4) & 1) + ((li & 3) >> 1)) << 4) + ((li & 1) << 3));
        uint32_t fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fa[t] = lds0 + wi * (WB_HALF * 2) + lowb + ((t ^ x) << 6);
            fb[t] = lds0 + (2 + wj) * (WB_HALF * 2) + lowb + ((t ^ x) << 6);
        }
        bf16x4 x0l[8], x0h[8], x1l[8], x1h[8];              // [0..3] = dY fragments, [4..7] = X fragments
#define WB_RD(lo, hi, addr, OFF)                                                                                     \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                         \
                 : "=&v"(lo), "=&v"(hi) : "v"(addr), "i"(OFF), "i"((OFF) + 2048))
#define WB_READ8(L, H, so, OFF)                                                                                      \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                                               \
        WB_RD(L[t_], H[t_], fa[t_] + (so), OFF);                                                                     \
        WB_RD(L[4 + t_], H[4 + t_], fb[t_] + (so), OFF);                                                             \
    }
#define WB_WAIT(cnt, L, H)                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                       \
                 : "+v"(L[0]), "+v"(L[1]), "+v"(L[2]), "+v"(L[3]), "+v"(L[4]), "+v"(L[5]), "+v"(L[6]), "+v"(L[7]),   \
                   "+v"(H[0]), "+v"(H[1]), "+v"(H[2]), "+v"(H[3]), "+v"(H[4]), "+v"(H[5]), "+v"(H[6]), "+v"(H[7]))
#define WB_MFMA16(L, H)                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                 \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                                       \
            __builtin_shufflevector(L[i_], H[i_], 0, 1, 2, 3, 4, 5, 6, 7),                                           \
            __builtin_shufflevector(L[4 + j_], H[4 + j_], 0, 1, 2, 3, 4, 5, 6, 7), acc[i_][j_], 0, 0, 0)
#define WB_SPREAD4                                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                           \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                           \
    }
        // MODE 0: steady state (issues stage it+3); 1: a later stage exists but nothing left to issue; 2: last stage
        auto body = [&](int it, auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            const uint32_t so = (uint32_t)(it & 3) * (WB_STAGE * 2), so2 = (uint32_t)((it + 1) & 3) * (WB_STAGE * 2);
            WB_READ8(x1l, x1h, so, 4096);                   // slice 1 of stage it
            WB_WAIT(15, x0l, x0h);                          // <= 15 outstanding: the 16 older reads (slice 0) are done
            if (MODE == 0) {
                dma(it + 3, 0); dma(it + 3, 1); dma(it + 3, 2); dma(it + 3, 3);
            }
            WB_MFMA16(x0l, x0h);
            if (MODE == 0) { WB_SPREAD4 }
            WB_WAIT(0, x1l, x1h);                           // this wave holds every fragment of stage it
            if (MODE <= 1) {
                if (MODE == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");     // stage it+1 landed (see header)
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                WB_READ8(x0l, x0h, so2, 0);                 // slice 0 of stage it+1
            }
            if (MODE == 0) {
                dma(it + 3, 4); dma(it + 3, 5); dma(it + 3, 6); dma(it + 3, 7);
            }
            WB_MFMA16(x1l, x1h);
            if (MODE == 0) { WB_SPREAD4 }
        };
#pragma unroll
        for (int q = 0; q < 8; ++q) dma(0, q);
        if (n > 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) dma(1, q);
        }
        if (n > 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) dma(2, q);
        }
        if (n > 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (n > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        WB_READ8(x0l, x0h, 0u, 0);
        int it = 0;
        for (; it + 3 < n; ++it) body(it, std::integral_constant<int, 0>{});
        for (; it + 1 < n; ++it) body(it, std::integral_constant<int, 1>{});
        body(it, std::integral_constant<int, 2>{});
#undef WB_RD
#undef WB_READ8
#undef WB_WAIT
#undef WB_MFMA16
#undef WB_SPREAD4
    }
    // acc[i][j][reg] = dW[tn*256 + 128 wi + 32 i + krow(reg,hh)][tk*256 + 128 wj + 32 j + r]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int nn = tn * 256 + 128 * wi + 32 * i + acr_krow(reg, hh);
                const int kk = tk * 256 + 128 * wj + 32 * j + r;
                slab[(int64_t)nn * K + kk] = acc[i][j][reg];
            }
}

// ---------------------------------------------------------------------------------------------------------------
// Eight-wave flavour of the 256x256 weight-gradient kernel: waves 2 x 4, each a 128x64 accumulator (8 MFMA tiles = 128
// registers), TWO waves per SIMD so that one wave's barrier / wait / slab store overlaps the other's MFMAs (what made the
// 320x256 NT kernel reach 1 PF).  Same ring, same LDS image, same split-major order; 4 DMA instructions per stage and
// thread, 12 transpose reads per 16-row slice and wave for 8 MFMAs.
// ---------------------------------------------------------------------------------------------------------------
// Bias gradient for free: the dY fragments a wave reads ARE the columns whose sums the bias gradient needs (fragment t of
// a wave = 32 columns x 16 token rows, 8 rows per lane half).  Wave (wi, wj) sums fragment t = wj -- the 8 waves then
// cover the tile's 256 columns once -- ~16 VALU per 8 MFMAs, hidden behind the matrix pipe.  Every workgroup of a row
// of tiles does it (no imbalance), the tk = 0 one writes its per-split partial; the final sum over the splits is
// the existing two-stage column-sum's second stage.  Replaces a full extra pass over dY (1.1 ms per step).
__device__ __forceinline__ float w8_sum8(const bf16x4& lo, const bf16x4& hi) {
    return (((float)lo[0] + (float)lo[1]) + ((float)lo[2] + (float)lo[3])) + (((float)hi[0] + (float)hi[1]) + ((float)hi[2] + (float)hi[3]));
}

template <bool COLSUM>                                     // compile-time: a runtime test inside the step body breaks its
__global__ __launch_bounds__(512, 2) void gemm_tn_bf16_wide_kernel(const bf16_t* __restrict__ dY, int64_t ldy,   // scheduling (2x slower)
                                                                   const bf16_t* __restrict__ X, int64_t ldx,
                                                                   float* __restrict__ slabs, int M, int N, int K, int nsplit,
                                                                   int steps_per_split, float* __restrict__ cs_part) {
    __shared__ __attribute__((aligned(1024))) bf16_t smem[4 * WB_STAGE];
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    const int ntk = K >> 8, ntiles = (N >> 8) * ntk;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int split = id / ntiles; id -= split * ntiles;
    const int tk = id % ntk, tn = id / ntk;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 2, wj = wave & 3;                 // output rows (n) 128 wi .., output columns (k) 64 wj ..
    const int r = lane & 31, hh = lane >> 5;
    const int total_steps = (M + WB_ROWS - 1) / WB_ROWS;
    const int st0 = split * steps_per_split, st1 = min(st0 + steps_per_split, total_steps);
    const int n = st1 - st0;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float* slab = slabs + (int64_t)split * N * K;
    if (n > 0) {
        // DMA instruction g = wave * 4 + q (0..31) of a stage: operand g >> 4 (dY, X), 128-column half (g >> 3) & 1,
        // row group g & 7 (4 rows each)
        uint32_t doff[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gI = wave * 4 + q;
            const int row = (gI & 7) * 4 + (lane >> 4);
            const int lc = (lane & 15) ^ ((row & 3) << 2);
            const int col = ((gI >> 4) ? tk : tn) * 256 + ((gI >> 3) & 1) * 128 + lc * 8;
            doff[q] = (uint32_t)(((int64_t)row * ((gI >> 4) ? ldx : ldy) + col) * 2);
        }
        auto dma = [&](int it, int q) {                      // one DMA instruction of stage `it` (all rows < M: see host)
            const int gI = wave * 4 + q;
            const int64_t m0 = (int64_t)(st0 + it) * WB_ROWS;
            const char* base = (gI >> 4) ? reinterpret_cast<const char*>(X + m0 * ldx) : reinterpret_cast<const char*>(dY + m0 * ldy);
            bf16_t* dst = smem + (it & 3) * WB_STAGE + (gI >> 3) * WB_HALF + (gI & 7) * 4 * 128;
            __builtin_amdgcn_global_load_lds((glb_vp)(base + doff[q]), (lds_vp)dst, 16, 0, 0);
        };
        const int li = lane & 15, x = (li >> 2) & 3;
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)smem;
        const uint32_t lowb = (uint32_t)((4 * hh + (li >> 2)) * 256 + ((2 * ((lane >> 4) & 1) + ((li & 3) >> 1)) << 4) + ((li & 1) << 3));
        uint32_t fa[4], fb[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) fa[t] = lds0 + wi * (WB_HALF * 2) + lowb + ((t ^ x) << 6);
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[t] = lds0 + (2 + (wj >> 1)) * (WB_HALF * 2) + lowb + ((((wj & 1) * 2 + t) ^ x) << 6);
        bf16x4 x0l[6], x0h[6], x1l[6], x1h[6];              // [0..3] = dY fragments, [4..5] = X fragments
#define W8_RD(lo, hi, addr, OFF)                                                                                     \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                         \
                 : "=&v"(lo), "=&v"(hi) : "v"(addr), "i"(OFF), "i"((OFF) + 2048))
#define W8_READ6(L, H, so, OFF)                                                                                      \
    { _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) { W8_RD(L[t_], H[t_], fa[t_] + (so), OFF); }                   \
      _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) { W8_RD(L[4 + t_], H[4 + t_], fb[t_] + (so), OFF); } }
#define W8_WAIT(cnt, L, H)                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                       \
                 : "+v"(L[0]), "+v"(L[1]), "+v"(L[2]), "+v"(L[3]), "+v"(L[4]), "+v"(L[5]),                           \
                   "+v"(H[0]), "+v"(H[1]), "+v"(H[2]), "+v"(H[3]), "+v"(H[4]), "+v"(H[5]))
#define W8_MFMA8(L, H)                                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                 \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                                       \
            __builtin_shufflevector(L[i_], H[i_], 0, 1, 2, 3, 4, 5, 6, 7),                                           \
            __builtin_shufflevector(L[4 + j_], H[4 + j_], 0, 1, 2, 3, 4, 5, 6, 7), acc[i_][j_], 0, 0, 0)
#define W8_SPREAD2                                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                           \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                           \
    }
    // this wave's share of the column sums: fragment WJS = wj, a compile-time index -- the loop is instantiated once per
    // wj and each wave runs its own copy (selecting the fragment at run time cost the kernel half its speed)
#define W8_PICK(L, H) w8_sum8(L[WJS], H[WJS])
        float csum = 0.f;
        auto body = [&](int it, auto mode_tag, auto wj_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            constexpr int WJS = decltype(wj_tag)::value;
            const uint32_t so = (uint32_t)(it & 3) * (WB_STAGE * 2), so2 = (uint32_t)((it + 1) & 3) * (WB_STAGE * 2);
            W8_READ6(x1l, x1h, so, 4096);                   // slice 1 of stage it
            W8_WAIT(12, x0l, x0h);                          // the 12 older reads (slice 0) are done
            if (MODE == 0) { dma(it + 3, 0); dma(it + 3, 1); }
            W8_MFMA8(x0l, x0h);
            if (COLSUM) csum += W8_PICK(x0l, x0h);
            if (MODE == 0) { W8_SPREAD2 }
            W8_WAIT(0, x1l, x1h);                           // this wave holds every fragment of stage it
            if (COLSUM) csum += W8_PICK(x1l, x1h);
            if (MODE <= 1) {
                // stage it+1 landed; stage it+2 (4) and the first half of stage it+3 (2) may stay in flight
                if (MODE == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                W8_READ6(x0l, x0h, so2, 0);                 // slice 0 of stage it+1
            }
            if (MODE == 0) { dma(it + 3, 2); dma(it + 3, 3); }
            W8_MFMA8(x1l, x1h);
            if (MODE == 0) { W8_SPREAD2 }
        };
#pragma unroll
        for (int q = 0; q < 4; ++q) dma(0, q);
        if (n > 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dma(1, q);
        }
        if (n > 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dma(2, q);
        }
        if (n > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        W8_READ6(x0l, x0h, 0u, 0);
        auto run = [&](auto wj_tag) {
            int it = 0;
            for (; it + 3 < n; ++it) body(it, std::integral_constant<int, 0>{}, wj_tag);
            for (; it + 1 < n; ++it) body(it, std::integral_constant<int, 1>{}, wj_tag);
            body(it, std::integral_constant<int, 2>{}, wj_tag);
        };
        if (COLSUM) {
            if (wj == 0) run(std::integral_constant<int, 0>{});
            else if (wj == 1) run(std::integral_constant<int, 1>{});
            else if (wj == 2) run(std::integral_constant<int, 2>{});
            else run(std::integral_constant<int, 3>{});
        } else {
            run(std::integral_constant<int, 0>{});
        }
#undef W8_RD
#undef W8_READ6
#undef W8_WAIT
#undef W8_MFMA8
#undef W8_SPREAD2
#undef W8_PICK
        if (COLSUM && tk == 0) {
            csum += __shfl_xor(csum, 32);                   // the two lane halves hold different token rows of a column
            if (hh == 0) cs_part[(int64_t)split * N + tn * 256 + 128 * wi + 32 * wj + r] = csum;
        }
    } else if (COLSUM && tk == 0 && hh == 0) {
        cs_part[(int64_t)split * N + tn * 256 + 128 * wi + 32 * wj + r] = 0.f;       // empty split: contributes nothing
    }
    // acc[i][j][reg] = dW[tn*256 + 128 wi + 32 i + krow(reg,hh)][tk*256 + 64 wj + 32 j + r]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int nn = tn * 256 + 128 * wi + 32 * i + acr_krow(reg, hh);
                const int kk = tk * 256 + 64 * wj + 32 * j + r;
                slab[(int64_t)nn * K + kk] = acc[i][j][reg];
            }
}

static bool wgrad_big_ok(int M, int N, int K) {
    const int env = acr_opt(ACR_OPT_WGRAD_VARIANT);      // 1: 128x128 tiles, 2: 256x256
    // small problems (that would not fill the chip) and widths that are not multiples of 256 stay on the 128x128 kernel
    return env == 2 && (N % 256) == 0 && (K % 256) == 0 && M >= 4096;
}

// The 256x256 kernels have no ragged-row path: they take the first M_main = floor(M / 32) * 32 token rows; the remaining
// (< 32) rows go through the 128x128 kernel (which masks ragged rows) into ONE extra slab, summed with the others.
struct WgradPlan { bool big; int m_main, tail, nsplit, sps, nslab; };
static WgradPlan wgrad_plan(int M, int N, int K) {
    WgradPlan p;
    p.big = wgrad_big_ok(M, N, K);
    p.m_main = p.big ? (M / WB_ROWS) * WB_ROWS : M;
    p.tail = M - p.m_main;
    const int tiles = p.big ? (N / 256) * (K / 256) : (N / 128) * (K / 128);
    const int total_steps = p.big ? p.m_main / WB_ROWS : (M + 63) / 64;
    int s = (p.big ? 256 : 512) / tiles;                     // one 256x256 or ~2 128x128 workgroups per CU
    if (s < 1) s = 1;
    if (s > 32) s = 32;
    if (s > total_steps) s = total_steps;
    p.nsplit = s;
    p.sps = (total_steps + s - 1) / s;
    p.nslab = s + (p.tail ? 1 : 0);
    return p;
}

extern "C" size_t acr_wgrad_ws_floats(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0 || (N % 128) || (K % 128)) return 0;
    return (size_t)wgrad_plan(M, N, K).nslab * (size_t)N * (size_t)K;
}

// slabs of the main part (and of the ragged tail) into ws; cs (nullable): per-slab column sums of dy for the fused bias path
static void wgrad_launch(const WgradPlan& p, const bf16_t* dy, int64_t ldy, const bf16_t* x, int64_t ldx, int M, int N, int K, float* ws,
                         float* cs, hipStream_t st) {
    const int tn_waves = acr_opt(ACR_OPT_WGRAD_WAVES);     // 4 or 8 waves per workgroup
    if (p.big && (tn_waves == 8 || cs)) {
        if (cs)
            hipLaunchKernelGGL(gemm_tn_bf16_wide_kernel<true>, dim3((N / 256) * (K / 256) * p.nsplit), dim3(512), 0, st, dy, ldy, x, ldx,
                               ws, p.m_main, N, K, p.nsplit, p.sps, cs);
        else
            hipLaunchKernelGGL(gemm_tn_bf16_wide_kernel<false>, dim3((N / 256) * (K / 256) * p.nsplit), dim3(512), 0, st, dy, ldy, x,
                               ldx, ws, p.m_main, N, K, p.nsplit, p.sps, (float*)nullptr);
    } else if (p.big) {
        hipLaunchKernelGGL(gemm_tn_bf16_big_kernel, dim3((N / 256) * (K / 256) * p.nsplit), dim3(256), 0, st, dy, ldy, x, ldx, ws,
                           p.m_main, N, K, p.nsplit, p.sps);
    } else {
        hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3((N / 128) * (K / 128) * p.nsplit), dim3(256), 0, st, dy, ldy, x, ldx, ws, M, N, K,
                           p.nsplit, p.sps);
    }
    if (p.tail) {                                            // ragged rows of the big path: one more slab (and column-sum row)
        const bf16_t* dyt = dy + (int64_t)p.m_main * ldy;
        const bf16_t* xt = x + (int64_t)p.m_main * ldx;
        hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3((N / 128) * (K / 128)), dim3(256), 0, st, dyt, ldy, xt, ldx,
                           ws + (int64_t)p.nsplit * N * K, p.tail, N, K, 1, 1);
        if (cs)
            hipLaunchKernelGGL(colsum_partial_kernel, dim3((N / 8 + 63) / 64, 1), dim3(256), 0, st, dyt, ldy, p.tail, N,
                               cs + (int64_t)p.nsplit * N);
    }
}

extern "C" int acr_wgrad_bf16(const void* dy, int64_t ldy, const void* x, int64_t ldx, int32_t M, int32_t N, int32_t K,
                              float* ws, void* dw, void* stream) {
    ACR_CHECK_ARG(dy && x && ws && dw, "acr_wgrad_bf16: null pointer");
    ACR_CHECK_ARG(M > 0 && N > 0 && K > 0 && (N % 128) == 0 && (K % 128) == 0, "acr_wgrad_bf16: N and K must be multiples of 128 (N=%d K=%d)", N, K);
    ACR_CHECK_ARG((ldy % 8) == 0 && (ldx % 8) == 0 && ldy >= N && ldx >= K && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0,
                  "acr_wgrad_bf16: row pitches must be multiples of 8 elements and operands 16-byte aligned");
    const WgradPlan p = wgrad_plan(M, N, K);
    hipStream_t st = (hipStream_t)stream;
    wgrad_launch(p, (const bf16_t*)dy, ldy, (const bf16_t*)x, ldx, M, N, K, ws, nullptr, st);
    const int64_t nk = (int64_t)N * K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nk / 4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, p.nslab,
                       nk, (bf16_t*)dw);
    return acr_check_launch("acr_wgrad_bf16");
}

// Weight AND bias gradient of a Linear in one sweep over dY (see the column sums inside gemm_tn_bf16_wide_kernel); falls
// back to acr_wgrad_bf16 + acr_colsum_bf16 for shapes the eight-wave kernel does not take.
extern "C" size_t acr_wgrad_bias_ws_floats(int32_t M, int32_t N, int32_t K) {
    const size_t a = acr_wgrad_ws_floats(M, N, K);
    if (a == 0) return 0;
    const size_t cs = (size_t)((M + CS_ROWS - 1) / CS_ROWS) * (size_t)N;        // fallback column-sum partials
    const size_t fused = (size_t)wgrad_plan(M, N, K).nslab * (size_t)N;
    return a + (cs > fused ? cs : fused);
}

extern "C" int acr_wgrad_bias_bf16(const void* dy, int64_t ldy, const void* x, int64_t ldx, int32_t M, int32_t N, int32_t K,
                                   float* ws, void* dw, void* dbias, void* stream) {
    ACR_CHECK_ARG(dy && x && ws && dw && dbias, "acr_wgrad_bias_bf16: null pointer");
    const size_t slab_floats = acr_wgrad_ws_floats(M, N, K);
    ACR_CHECK_ARG(slab_floats > 0, "acr_wgrad_bias_bf16: N and K must be multiples of 128 (N=%d K=%d)", N, K);
    const WgradPlan p = wgrad_plan(M, N, K);
    if (!p.big) {
        int rc = acr_wgrad_bf16(dy, ldy, x, ldx, M, N, K, ws, dw, stream);
        if (rc) return rc;
        return acr_colsum_bf16(dy, ldy, M, N, ws + slab_floats, dbias, stream);
    }
    ACR_CHECK_ARG((ldy % 8) == 0 && (ldx % 8) == 0 && ldy >= N && ldx >= K && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0,
                  "acr_wgrad_bias_bf16: row pitches must be multiples of 8 elements and operands 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    float* cs = ws + slab_floats;
    wgrad_launch(p, (const bf16_t*)dy, ldy, (const bf16_t*)x, ldx, M, N, K, ws, cs, st);
    const int64_t nk = (int64_t)N * K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nk / 4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, p.nslab,
                       nk, (bf16_t*)dw);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 15) / 16), dim3(256), 0, st, (const float*)cs, p.nslab, N, (bf16_t*)dbias);
    return acr_check_launch("acr_wgrad_bias_bf16");
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 convolutions of the ResNetV2 stem in NCHW (models/resnetv2.py:186-190 conv1/conv3/downsample, stride 1).
// MIOpen runs them as NCHW->NHWC transpose + GEMM + transpose back (3x the bytes; 10 of the 64 ms step).  In NCHW a
// 1x1 convolution IS a GEMM per sample: Y[n] (Cout x HW) = W (Cout x Cin) . X[n] (Cin x HW) with the pixel index
// contiguous, i.e. the B operand is contraction-strided -- exactly what the transpose-read fragments of the TN kernel
// handle.  So: A = weight tile [128 co][64 ci] (128-byte rows, chunk swizzle, ds_read_b128 fragments), B = activation
// tile [64 ci][128 pixels] (256-byte rows, quarter swizzle, ds_read_b64_tr_b16 fragments), both by LDS-DMA, no layout
// change anywhere.  The same kernel computes the input gradient (A = W^T).  The weight gradient
// dW[co][ci] = sum_n sum_p dY[n][co][p] X[n][ci][p] has the contraction index contiguous in both operands ("NT"),
// summed over samples and pixel chunks with fp32 slabs (deterministic).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void c1_stage_b(bf16_t* ldsbuf, const bf16_t* xs, int64_t row_stride, int k0, int p0,
                                           int64_t max_off, int wave, int lane) {
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rbase = (wave * 4 + i) * 4;
        const int row = rbase + (lane >> 4);
        const int lc = (lane & 15) ^ ((row & 3) << 2);
        int64_t off = (int64_t)(k0 + row) * row_stride + p0 + lc * 8;
        off = off < max_off ? off : max_off;                 // pixel tiles past HW run into the next row: keep it in bounds
        __builtin_amdgcn_global_load_lds((glb_vp)(xs + off), (lds_vp)(ldsbuf + rbase * 128), 16, 0, 0);
    }
}

// A fragment in the k-slot order of wg_frag: half h, element j <-> k = 16 s + 8 (j >> 2) + 4 h + (j & 3)
__device__ __forceinline__ bf16x8 c1_frag_a(const bf16_t* ldsbuf, int row, int s, int h) {
    const int sw = (row >> 1) & 7;
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(ldsbuf + row * 64 + (((2 * s) ^ sw) << 3) + 4 * h);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(ldsbuf + row * 64 + (((2 * s + 1) ^ sw) << 3) + 4 * h);
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

__global__ __launch_bounds__(256) void conv1x1_nn_kernel(const bf16_t* __restrict__ W, int64_t ldw,
                                                         const bf16_t* __restrict__ X, bf16_t* __restrict__ Y,
                                                         const bf16_t* __restrict__ addend, int M, int K, int HW,
                                                         int nsamp) {
    __shared__ __attribute__((aligned(1024))) bf16_t smem[4 * GL_TILE];      // [A0 | B0 | A1 | B1]
    const int ntm = (M + 127) >> 7, ntp = (HW + 127) >> 7;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id % ntm; id /= ntm;
    const int tp = id % ntp;
    const int n = id / ntp;
    const int m0 = tm * 128, p0 = tp * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const bf16_t* xs = X + (int64_t)n * K * HW;
    const int64_t max_off = (int64_t)(nsamp - n) * K * HW - 8;              // last 16-byte chunk of the tensor, from xs
    const int nk = K >> 6;
    glds_stage(smem, W, ldw, m0, M, 0, wave, lane);
    c1_stage_b(smem + GL_TILE, xs, HW, 0, p0, max_off, wave, lane);
    acr_dma_barrier();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt, cur ^= 1) {
        if (kt + 1 < nk) {
            glds_stage(smem + (cur ^ 1) * 2 * GL_TILE, W, ldw, m0, M, (kt + 1) * 64, wave, lane);
            c1_stage_b(smem + (cur ^ 1) * 2 * GL_TILE + GL_TILE, xs, HW, (kt + 1) * 64, p0, max_off, wave, lane);
        }
        const bf16_t* as = smem + cur * 2 * GL_TILE;
        const bf16_t* bs = as + GL_TILE;
        const int ra = wm * 64 + r;
        bf16x4 bl[8], bh[8];                                // all 8 activation fragments of the step (asm reads)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            wg_frag_issue(bs, ks, 2 * wn, lane, bl[2 * ks], bh[2 * ks]);
            wg_frag_issue(bs, ks, 2 * wn + 1, lane, bl[2 * ks + 1], bh[2 * ks + 1]);
        }
        WG_FRAG_WAIT8(bl, bh);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 a0 = c1_frag_a(as, ra, ks, hh), a1 = c1_frag_a(as, ra + 32, ks, hh);
            const bf16x8 b0 = __builtin_shufflevector(bl[2 * ks], bh[2 * ks], 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8 b1 = __builtin_shufflevector(bl[2 * ks + 1], bh[2 * ks + 1], 0, 1, 2, 3, 4, 5, 6, 7);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        acr_dma_barrier();
    }
    // acc[mt][nt][reg] = Y[n][m0 + 64 wm + 32 mt + krow(reg,hh)][p0 + 64 wn + 32 nt + r]; LDS-staged 16-byte stores
    float* stile = reinterpret_cast<float*>(smem) + wave * 4096;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                stile[(mt * 32 + acr_krow(reg, hh)) * 64 + nt * 32 + r] = acc[mt][nt][reg];
    bf16_t* ys = Y + (int64_t)n * M * HW;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = lane + 64 * i;
        const int lrow = idx >> 3, c8 = (idx & 7) * 8;
        const int row = m0 + wm * 64 + lrow, col = p0 + wn * 64 + c8;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stile + lrow * 64 + c8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(stile + lrow * 64 + c8 + 4);
        if (row < M && col < HW) {
            f32x4 l2 = lo, h2 = hi;
            if (addend) {                                   // gradient arriving over the skip connection, summed in fp32
                const bf16x8 ad = *reinterpret_cast<const bf16x8*>(addend + (int64_t)n * M * HW + (int64_t)row * HW + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) { l2[e] += (float)ad[e]; h2[e] += (float)ad[4 + e]; }
            }
            bf16x8 o = {(bf16_t)l2[0], (bf16_t)l2[1], (bf16_t)l2[2], (bf16_t)l2[3],
                        (bf16_t)h2[0], (bf16_t)h2[1], (bf16_t)h2[2], (bf16_t)h2[3]};
            *reinterpret_cast<bf16x8*>(ys + (int64_t)row * HW + col) = o;
        }
    }
}

// dW[co][ci] = sum over (sample, 64-pixel chunk) of dY[n][co][p] X[n][ci][p]; both operands pixel-contiguous (NT).
__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ X,
                                                            float* __restrict__ slabs, int M, int N, int HW, int nsamp,
                                                            int nsplit, int steps_per_split) {
    __shared__ __attribute__((aligned(1024))) bf16_t smem[4 * GL_TILE];
    const int ntn = (N + 127) >> 7;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int split = id % nsplit; id /= nsplit;
    const int tn = id % ntn, tm = id / ntn;
    const int m0 = tm * 128, n0 = tn * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int kps = (HW + 63) >> 6;                          // k-steps per sample
    const int total = nsamp * kps;
    const int st0 = split * steps_per_split, st1 = min(st0 + steps_per_split, total);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto stage = [&](bf16_t* buf, int stp) {
        const int n = stp / kps, k0 = (stp - n * kps) * 64;
        // rows clamp to the last channel; columns past HW run into the next row (still inside the tensor except at its
        // very end, which the row clamp + pointer clamp below keep in bounds) and are zeroed in LDS before use
        const bf16_t* ya = dY + (int64_t)n * M * HW;
        const bf16_t* xb = X + (int64_t)n * N * HW;
        typedef __attribute__((address_space(3))) void* lds_vp;
        typedef const __attribute__((address_space(1))) void* glb_vp;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rbase = (wave * 4 + i) * 8;
            const int row = rbase + (lane >> 3);
            const int lc = (lane & 7) ^ ((row >> 1) & 7);
            const int col = min(k0 + lc * 8, HW - 8);                        // keep the 16-byte chunk inside the row
            __builtin_amdgcn_global_load_lds((glb_vp)(ya + (int64_t)min(m0 + row, M - 1) * HW + col),
                                             (lds_vp)(buf + rbase * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_vp)(xb + (int64_t)min(n0 + row, N - 1) * HW + col),
                                             (lds_vp)(buf + GL_TILE + rbase * 64), 16, 0, 0);
        }
    };
    if (st0 < st1) {
        stage(smem, st0);
        acr_dma_barrier();
        int cur = 0;
        for (int stp = st0; stp < st1; ++stp, cur ^= 1) {
            if (stp + 1 < st1) stage(smem + (cur ^ 1) * 2 * GL_TILE, stp + 1);
            bf16_t* as = smem + cur * 2 * GL_TILE;
            const bf16_t* bs = as + GL_TILE;
            const int k0 = (stp % kps) * 64;
            if (k0 + 64 > HW) {
                // ragged last chunk of a sample: chunks whose pixels are >= HW were clamped to the row's last chunk;
                // zero them in the dY tile so they contribute nothing
                for (int idx = tid; idx < 128 * 8; idx += 256) {
                    const int row = idx >> 3, pc = idx & 7;
                    const int lc = pc ^ ((row >> 1) & 7);
                    if (k0 + lc * 8 >= HW) {
                        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                        *reinterpret_cast<bf16x8*>(as + row * 64 + pc * 8) = z;
                    }
                }
                __syncthreads();
            }
            const int ra = wm * 64 + r, rb = wn * 64 + r;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = 2 * ks + hh;
                const bf16x8 a0 = glds_frag(as, ra, c), a1 = glds_frag(as, ra + 32, c);
                const bf16x8 b0 = glds_frag(bs, rb, c), b1 = glds_frag(bs, rb + 32, c);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            }
            acr_dma_barrier();
        }
    }
    float* slab = slabs + (int64_t)split * M * N;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = m0 + wm * 64 + mt * 32 + acr_krow(reg, hh);
                const int col = n0 + wn * 64 + nt * 32 + r;
                if (row < M && col < N) slab[(int64_t)row * N + col] = acc[mt][nt][reg];
            }
}

// Slab reduction for small weights (few outputs, many splits): 16 float4 outputs x 16 split groups per block, each
// group sums its splits in order, the 16 partials are combined in fixed order through LDS (deterministic).
__global__ __launch_bounds__(256) void wgrad_reduce_small_kernel(const float* __restrict__ slabs, int nsplit, int64_t nk,
                                                                 bf16_t* __restrict__ out) {
    __shared__ f32x4 part[16][17];
    const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int64_t i = ((int64_t)blockIdx.x * 16 + o) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < nk)
        for (int sp = g; sp < nsplit; sp += 16) s += *reinterpret_cast<const f32x4*>(slabs + (int64_t)sp * nk + i);
    part[g][o] = s;
    __syncthreads();
    if (g == 0 && i < nk) {
        f32x4 t = part[0][o];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += part[k][o];
        acr_store4<bf16_t>(out + i, t);
    }
}

static int c1_split(int M, int N, int HW, int nsamp) {
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    int s = 512 / tiles;
    const int total = nsamp * ((HW + 63) / 64);
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    if (s > total) s = total;
    return s;
}

extern "C" int acr_conv1x1_bf16(const void* w, int64_t ldw, const void* x, const void* addend, void* y, int32_t nsamp,
                                int32_t cout, int32_t cin, int32_t hw, void* stream) {
    ACR_CHECK_ARG(w && x && y, "acr_conv1x1_bf16: null pointer");
    ACR_CHECK_ARG(nsamp > 0 && cout > 0 && cin >= 64 && (cin % 64) == 0 && hw >= 8 && (hw % 8) == 0 && (ldw % 8) == 0 && ldw >= cin,
                  "acr_conv1x1_bf16: need cin %% 64 == 0, hw %% 8 == 0 (cout=%d cin=%d hw=%d)", cout, cin, hw);
    ACR_CHECK_ARG(((uintptr_t)w & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)addend & 15) == 0,
                  "acr_conv1x1_bf16: 16-byte alignment");
    const int64_t tiles = (int64_t)((cout + 127) / 128) * ((hw + 127) / 128) * nsamp;
    ACR_CHECK_ARG(tiles < (1ll << 31), "acr_conv1x1_bf16: grid too large");
    hipLaunchKernelGGL(conv1x1_nn_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, ldw,
                       (const bf16_t*)x, (bf16_t*)y, (const bf16_t*)addend, cout, cin, hw, nsamp);
    return acr_check_launch("acr_conv1x1_bf16");
}

extern "C" size_t acr_conv1x1_wgrad_ws_floats(int32_t nsamp, int32_t cout, int32_t cin, int32_t hw) {
    return (size_t)c1_split(cout, cin, hw, nsamp) * (size_t)cout * (size_t)cin;
}

extern "C" int acr_conv1x1_wgrad_bf16(const void* dy, const void* x, int32_t nsamp, int32_t cout, int32_t cin, int32_t hw,
                                      float* ws, void* dw, void* stream) {
    ACR_CHECK_ARG(dy && x && ws && dw, "acr_conv1x1_wgrad_bf16: null pointer");
    ACR_CHECK_ARG(nsamp > 0 && cout > 0 && cin > 0 && hw >= 8 && (hw % 8) == 0 && ((int64_t)cout * cin) % 4 == 0,
                  "acr_conv1x1_wgrad_bf16: need hw %% 8 == 0 and cout*cin %% 4 == 0");
    ACR_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0, "acr_conv1x1_wgrad_bf16: 16-byte alignment");
    const int nsplit = c1_split(cout, cin, hw, nsamp);
    const int total = nsamp * ((hw + 63) / 64);
    const int sps = (total + nsplit - 1) / nsplit;
    const int tiles = ((cout + 127) / 128) * ((cin + 127) / 128);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv1x1_wgrad_kernel, dim3(tiles * nsplit), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, ws,
                       cout, cin, hw, nsamp, nsplit, sps);
    const int64_t nk = (int64_t)cout * cin;
    hipLaunchKernelGGL(wgrad_reduce_small_kernel, dim3((unsigned)((nk / 4 + 15) / 16)), dim3(256), 0, st, (const float*)ws,
                       nsplit, nk, (bf16_t*)dw);
    return acr_check_launch("acr_conv1x1_wgrad_bf16");
}
