// 3x3 stride-1 SAME convolutions of the ResNetV2 stem (models/resnetv2.py:171-216 conv2 of every bottleneck;
// models/layers/std_conv.py:40-65) in NCHW fp32 with SPLIT PRODUCTS (acr_math ACR_MATH_BF16X3) as implicit GEMMs on the bf16
// MFMA -- the last library code on the f32_split step's hot path were MIOpen's Winograd / implicit-GEMM kernels and their
// NCHW <-> NHWC transposes.  No im2col buffer, no layout change:
//
//   forward   y[n][co][p] = sum_t sum_ci Wp[co][t*Cin + ci] * x[n][ci][p + off_t] * valid_t(p)      off_t = (ty-1) W + (tx-1)
//   input     dx = the same kernel on dy with Wd[ci][t'*Cout + co] = w[co][ci][8 - t']              (taps flipped, roles swapped)
//   weight    dWp[co][t*Cin + ci] = sum_n sum_p dy[n][co][p] * x[n][ci][p + off_t] * valid_t(p)    (conv3x3_wgrad_split_kernel)
//
// GEMM view of the forward, per sample: M = output channels, N = the H*W pixels in flattened order (tiles of 128 consecutive
// pixels), K = 9 Cin in 16-deep stages that never straddle a tap (Cin % 16 == 0).  The A operand is the packed weight, read
// exactly like a Linear's (k contiguous: gemm_f32.hip's KC image).  The B operand of a stage is 16 channel rows x 128 pixels
// SHIFTED by the stage's tap offset: an LDS-DMA whose source address is only 4-byte aligned for the +-1 column taps (measured
// to work: scripts/lab/micro/dma_unaligned.hip); pixels whose tap falls outside the image are zeroed in registers on the
// fragment (8 v_cndmask per fragment, a per-lane 9-bit validity mask computed once) right before the three-way split.
// Shifted reads run up to W + 1 floats before a sample's first channel row and up to 128 + W + 1 floats past its last one:
// inside the tensor that is the neighbouring channel / sample (values masked).  At the tensor's two ENDS they would leave the
// allocation: the (at most two) workgroups per launch whose window touches an end take a careful issue path (c3_edge_fix) --
// every lane's DMA source is clamped into the tensor, and the lanes whose 16 bytes were not entirely inside it overwrite their
// LDS slot with guarded element loads (zero outside) once the DMA has landed.  No byte outside [x, x + numel) is ever
// addressed; rounds 3-4 asked the caller for ACR_CONV3X3_PAD floats of readable slack instead.
// Arithmetic, tile structure, ring and counted waits are gemm_f32_split_kernel's.
#include <type_traits>

#include "acr_common.h"

#define C3_BM 128
#define C3_BN 128
#define C3_BK 16
#define C3_TILE (C3_BM * C3_BK)      // floats per operand per stage (8 KiB)
#define C3_SLOTS 4

typedef __attribute__((address_space(3))) void* c3_lds_vp;
typedef const __attribute__((address_space(1))) void* c3_glb_vp;

// Tap table of the generalised kernels (template argument TAB; the 3x3 stride-1 kernels never read it): the contraction is
// ntap * C, tap t multiplies the C channel rows tcb[t] .. of x SHIFTED by toff[t] = tdy * W + tdx pixels, zero where
// (y + tdy, x + tdx) leaves the H x W grid.  That is every strided SAME convolution after a space-to-depth pass (stride 2: the four
// pixel phases of the input become channel groups of a half-resolution grid, a tap reads ONE phase at a shift of 0 / +-1), its
// input gradient (one launch per output phase over the taps that feed it) and the 7x7 stem convolution (16 (dy, dx) taps x the
// 12 phase-channel rows + 4 zero rows).  x and y may be channel slices of larger tensors (xrows / yrows rows per sample).
#define C3_MAXTAP 16
struct C3Tab {
    int ntap, xrows, yrows, maxoff;  // maxoff = max |toff|
    int toff[C3_MAXTAP], tcb[C3_MAXTAP];
    int tdydx[C3_MAXTAP];            // (tdy + 8) | (tdx + 8) << 8
};
struct Conv3Args {
    const float* w;                  // packed weights (M, 9 * C): w[m][t * C + c]
    const float* x;                  // (nsamp, C, H * W)
    float* y;                        // (nsamp, M, H * W)
    int M, C, H, W, HW, nsamp;
    int tiles_m, tiles_n;
    int ksplit, sps;                 // K-split of small launches: `ksplit` parts of `sps` stages each into slabs ws[part] (laid out like y)
    float* ws;
    C3Tab t;
};

__device__ __forceinline__ void c3_split3(const f32x4& lo4, const f32x4& hi4, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? lo4[e] : hi4[e - 4];
        const __bf16 h0 = (__bf16)x;
        const float r1 = x - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const float r2 = r1 - (float)h1;
        p0[e] = h0; p1[e] = h1; p2[e] = (__bf16)r2;
    }
}
#define C3_MFMA6(ACC, A, Bv)                                                         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], Bv[2], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], Bv[0], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], Bv[1], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], Bv[1], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], Bv[0], ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], Bv[0], ACC, 0, 0, 0);

// 9-bit validity mask of pixel p: bit t set iff tap t of p lies inside the image
__device__ __forceinline__ int c3_valid9(int p, int H, int W, int HW) {
    if (p >= HW) return 0;
    const int y = p / W, x = p - y * W;
    int m = 0;
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
            const int yy = y + ty - 1, xx = x + tx - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) m |= 1 << (ty * 3 + tx);
        }
    return m;
}

template <bool TAB> __device__ __forceinline__ int c3_valid(const Conv3Args& g, int p) {
    if (!TAB) return c3_valid9(p, g.H, g.W, g.HW);
    if (p >= g.HW) return 0;
    const int y = p / g.W, x = p - y * g.W;
    int m = 0;
    for (int t = 0; t < g.t.ntap; ++t) {
        const int yy = y + (g.t.tdydx[t] & 255) - 8, xx = x + (g.t.tdydx[t] >> 8) - 8;
        if (yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) m |= 1 << t;
    }
    return m;
}

// Careful form of one 16-byte LDS-DMA piece for the workgroups at the tensor's two ends.  `idx` = float index of the lane's first
// element relative to `base` (may be negative or reach past `total`), `dst` = the piece's LDS base (the lane's 16 bytes land at
// dst + 4 * lane floats).  The DMA is always issued (the stage's vmcnt arithmetic stays what the counted waits assume) from an
// address clamped into [base, base + total - 4]; a lane whose window was not entirely inside then waits for it and overwrites
// its slot element by element, zero where the element lies outside the tensor (those values are masked by the caller anyway).
__device__ __forceinline__ void c3_dma_careful(const float* __restrict__ base, int64_t idx, int64_t total, float* dst, int lane) {
    const int64_t cl = idx < 0 ? 0 : (idx > total - 4 ? total - 4 : idx);
    __builtin_amdgcn_global_load_lds((c3_glb_vp)(base + cl), (c3_lds_vp)dst, 16, 0, 0);
    if (cl != idx) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the clamped DMA has written this lane's slot
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (idx + e >= 0 && idx + e < total) ? base[idx + e] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[4 * lane + e] = v[e];
    }
}

__global__ __launch_bounds__(256, 2) void conv3x3_split_kernel(const Conv3Args g) {
    __shared__ __attribute__((aligned(1024))) float smem[C3_SLOTS * 2 * C3_TILE];      // [slot][A | B], 64 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int ntile = g.tiles_m * g.tiles_n;
    const int t1 = acr_xcd_remap(blockIdx.x, ntile * g.nsamp * g.ksplit);
    const int part = t1 / (ntile * g.nsamp), t0 = t1 - part * (ntile * g.nsamp);
    const int sample = t0 / ntile, tt = t0 - sample * ntile;
    const int tn = tt / g.tiles_m, tm = tt - tn * g.tiles_m;               // the tile rows of one pixel tile are neighbours
    const int sbeg = part * g.sps;                                         // first stage of this part
    const int m0 = tm * C3_BM, n0 = tn * C3_BN;
    const int lda = 9 * g.C;
    const float* __restrict__ pa = g.w;
    const float* __restrict__ pb = g.x + (int64_t)sample * g.C * g.HW;
    const bool compute = m0 + wm * 64 < g.M;                               // Cout = 64: the lower wave row has no outputs
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // A: piece = 16 rows x 64 bytes, lane -> (row = l >> 2, 16-byte chunk l & 3), chunk XOR-swizzled by (row >> 2) & 3 on the source
    int offa[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + (lane >> 2);
        offa[i] = min(m0 + row, g.M - 1) * lda + (((lane & 3) ^ ((row >> 2) & 3)) << 2);
    }
    // B: piece = 2 channel rows of 128 pixels; lane -> (row = l >> 5, pixels 4 (l & 31) .. + 3)
    int rowb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) rowb[i] = ((wave * 2 + i) * 2 + (lane >> 5)) * g.HW;
    const int pix = n0 + 4 * (lane & 31);
    const int valid0 = c3_valid9(n0 + wn * 64 + r, g.H, g.W, g.HW), valid1 = c3_valid9(n0 + wn * 64 + 32 + r, g.H, g.W, g.HW);
    const int nst = min(lda / C3_BK - sbeg, g.sps);
    // the two ends of the tensor: the first pixel tile of sample 0 reaches up to W + 1 floats in front of it, the last pixel
    // tile(s) of the last sample up to 128 + W + 1 floats behind it
    const int64_t total = (int64_t)g.nsamp * g.C * g.HW;
    const bool edge = (sample == 0 && n0 < g.W + 1) || (sample == g.nsamp - 1 && n0 + C3_BN + g.W + 1 > g.HW);
    auto issue = [&](int st) {
        float* d = smem + (st & (C3_SLOTS - 1)) * 2 * C3_TILE;
        const int k0 = (sbeg + st) * C3_BK;
        const int tap = k0 / g.C, ci0 = k0 - tap * g.C;                    // uniform
        const int ty = tap / 3, off = (ty - 1) * g.W + (tap - 3 * ty - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((c3_glb_vp)(pa + k0 + offa[i]), (c3_lds_vp)(d + (wave * 2 + i) * 256), 16, 0, 0);
        if (edge) {                                          // uniform: this workgroup's windows can leave the tensor (see the header)
            const int64_t i0 = (int64_t)sample * g.C * g.HW + (int64_t)ci0 * g.HW + (pix + off);
#pragma unroll
            for (int i = 0; i < 2; ++i) c3_dma_careful(g.x, i0 + rowb[i], total, d + C3_TILE + (wave * 2 + i) * 256, lane);
            return;
        }
        const float* xb = pb + (int64_t)ci0 * g.HW + (pix + off);                // leaves the sample at its ends, never the tensor
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((c3_glb_vp)(xb + rowb[i]), (c3_lds_vp)(d + C3_TILE + (wave * 2 + i) * 256), 16, 0, 0);
    };
#pragma unroll
    for (int st = 0; st < C3_SLOTS - 1; ++st)
        if (st < nst) issue(st);
    bf16x8 ap[2][2][3], bp[2][2][3];                        // [register set][block][piece]
    f32x4 ra[2][2], rb[2][2];
#define C3_ALL(SET)                                                                                 \
    C3_MFMA6(acc[0][0], ap[SET][0], bp[SET][0]) C3_MFMA6(acc[0][1], ap[SET][0], bp[SET][1])        \
    C3_MFMA6(acc[1][0], ap[SET][1], bp[SET][0]) C3_MFMA6(acc[1][1], ap[SET][1], bp[SET][1])
    // stage st: wait until it has landed (stages st+1, st+2 may stay in flight: 4 DMA instructions each), publish it, refill the
    // slot stage st-1 was read from, read + mask + split stage st into register set SET while the MFMAs of stage st-1 (set SET^1) run
    auto step = [&](int st, auto set_tag, auto first_tag) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        if (st + 2 < nst) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + C3_SLOTS - 1 < nst) issue(st + C3_SLOTS - 1);
        if (!compute) return;
        const float* sa = smem + (st & (C3_SLOTS - 1)) * 2 * C3_TILE;
        const float* sb = sa + C3_TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + r, sw = (row >> 2) & 3;
            ra[i][0] = *reinterpret_cast<const f32x4*>(sa + row * C3_BK + (((2 * h) ^ sw) << 2));
            ra[i][1] = *reinterpret_cast<const f32x4*>(sa + row * C3_BK + (((2 * h + 1) ^ sw) << 2));
        }
        const int tap = ((sbeg + st) * C3_BK) / g.C;        // uniform
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* p = sb + (8 * h) * C3_BN + wn * 64 + j * 32 + r;
            const bool ok = (((j ? valid1 : valid0) >> tap) & 1) != 0;
            rb[j][0] = f32x4{p[0], p[C3_BN], p[2 * C3_BN], p[3 * C3_BN]};
            rb[j][1] = f32x4{p[4 * C3_BN], p[5 * C3_BN], p[6 * C3_BN], p[7 * C3_BN]};
            if (!ok) { rb[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; rb[j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) c3_split3(ra[i][0], ra[i][1], ap[SET][i][0], ap[SET][i][1], ap[SET][i][2]);
#pragma unroll
        for (int j = 0; j < 2; ++j) c3_split3(rb[j][0], rb[j][1], bp[SET][j][0], bp[SET][j][1], bp[SET][j][2]);
        if (!FIRST) {
            C3_ALL(SET ^ 1)
#pragma unroll
            for (int it = 0; it < 24; ++it) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA of stage st - 1
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);      // eight VALU instructions of stage st's split
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    step(0, std::integral_constant<int, 0>{}, std::true_type{});
    for (int st = 1; st < nst; st += 2) {
        step(st, std::integral_constant<int, 1>{}, std::false_type{});
        if (st + 1 < nst) step(st + 1, std::integral_constant<int, 0>{}, std::false_type{});
    }
    if (!compute) return;
    if (nst & 1) { C3_ALL(0) } else { C3_ALL(1) }
    // ---- epilogue: lane (r, h), register e of a 32x32 accumulator = row krow(e, h), column (pixel) r
    float* yb = (g.ksplit > 1 ? g.ws + (int64_t)part * g.nsamp * g.M * g.HW : g.y) + (int64_t)sample * g.M * g.HW;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + acr_krow(e, h);
                if (row < g.M && col < g.HW) yb[(int64_t)row * g.HW + col] = acc[i][j][e];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same product with the packed weight given as a split-product IMAGE (acr_x3_image of w_packed: tiled bf16 planes, tiny, made
// once per weight version): only the shifted activation tile is split in registers -- half the vector work per MFMA of the kernel
// above, which is bound by exactly that work (round 5; the 1x1 convolutions made the same move in round 4, gemm_f32_wimg_kernel,
// whose ring this is: 3 slots x [A planes 12 KiB | B fp32 8 KiB], DMA two stages ahead, 3 + 2 pieces per wave and stage, counted
// vmcnt, fence-free barrier).  B tile, tap shift, validity masks, careful edge path and K-split slabs are the kernel's above.
// ---------------------------------------------------------------------------------------------------------------------------------
#define C3W_PLANE_B 4096                  // one plane of a 128-row x 16-deep A stage in the tiled image
#define C3W_STAGE_B (3 * C3W_PLANE_B + C3_TILE * 4)        // 20 KiB
#define C3W_SLOTS 3
#define C3W_RD128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "i"(OFF))
#define C3W_RD32(dst, addr, OFF) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "i"(OFF))
template <bool TAB> __global__ __launch_bounds__(256, 2) void conv3x3_wimg_kernel(const Conv3Args g) {
    __shared__ __attribute__((aligned(1024))) char smem[C3W_SLOTS * C3W_STAGE_B];       // 60 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int ntile = g.tiles_m * g.tiles_n;
    const int t1 = acr_xcd_remap(blockIdx.x, ntile * g.nsamp * g.ksplit);
    const int part = t1 / (ntile * g.nsamp), t0 = t1 - part * (ntile * g.nsamp);
    const int sample = t0 / ntile, tt = t0 - sample * ntile;
    const int tn = tt / g.tiles_m, tm = tt - tn * g.tiles_m;
    const int sbeg = part * g.sps;
    const int m0 = tm * C3_BM, n0 = tn * C3_BN;
    const int nkb = (TAB ? g.t.ntap : 9) * g.C / C3_BK;     // stages per row block of the weight image
    const int xrows = TAB ? g.t.xrows : g.C, yrows = TAB ? g.t.yrows : g.M, maxoff = TAB ? g.t.maxoff : g.W + 1;
    const char* __restrict__ pa = reinterpret_cast<const char*>(g.w) + ((int64_t)tm * nkb + sbeg) * (3 * C3W_PLANE_B) + wave * 3072 + lane * 16;
    const float* __restrict__ pb = g.x + (int64_t)sample * xrows * g.HW;
    const bool compute = m0 + wm * 64 < g.M;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int rowb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) rowb[i] = ((wave * 2 + i) * 2 + (lane >> 5)) * g.HW;
    const int pix = n0 + 4 * (lane & 31);
    const int valid0 = c3_valid<TAB>(g, n0 + wn * 64 + r), valid1 = c3_valid<TAB>(g, n0 + wn * 64 + 32 + r);
    const int nst = min(nkb - sbeg, g.sps);
    const int64_t total = (int64_t)g.nsamp * xrows * g.HW;
    const bool edge = (sample == 0 && n0 < maxoff) || (sample == g.nsamp - 1 && n0 + C3_BN + maxoff > g.HW);
    auto issue = [&](int st, int slot) {                    // 3 A pieces (contiguous KiB of the image) + 2 shifted B pieces
        char* d = smem + slot * C3W_STAGE_B;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_global_load_lds((c3_glb_vp)(pa + (int64_t)st * (3 * C3W_PLANE_B) + i * 1024), (c3_lds_vp)(d + (wave * 3 + i) * 1024), 16, 0, 0);
        const int k0 = (sbeg + st) * C3_BK;
        const int tap = k0 / g.C;
        int ci0 = k0 - tap * g.C, off;
        if (TAB) { off = g.t.toff[tap]; ci0 += g.t.tcb[tap]; }
        else { const int ty = tap / 3; off = (ty - 1) * g.W + (tap - 3 * ty - 1); }
        float* db = reinterpret_cast<float*>(d + 3 * C3W_PLANE_B);
        if (edge) {
            const int64_t i0 = (int64_t)sample * xrows * g.HW + (int64_t)ci0 * g.HW + (pix + off);
#pragma unroll
            for (int i = 0; i < 2; ++i) c3_dma_careful(g.x, i0 + rowb[i], total, db + (wave * 2 + i) * 256, lane);
            return;
        }
        const float* xb = pb + (int64_t)ci0 * g.HW + (pix + off);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((c3_glb_vp)(xb + rowb[i]), (c3_lds_vp)(db + (wave * 2 + i) * 256), 16, 0, 0);
    };
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    const uint32_t fa = lbase + (wm * 64 + r) * 32 + (h ^ ((r >> 3) & 1)) * 16;
    const uint32_t fb = lbase + 3 * C3W_PLANE_B + ((8 * h) * C3_BN + wn * 64 + r) * 4;
    issue(0, 0);
    issue(min(1, nst - 1), 1);
    bf16x8 ap[2][2][3], bp[2][2][3];                        // [register set][block][plane]
    float rb[2][8];
#define C3W_MFMA6(SET, I, J)                                                                                                 \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][2], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][2], bp[SET][J][0], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][1], bp[SET][J][1], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][1], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][1], bp[SET][J][0], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][0], acc[I][J], 0, 0, 0);
    auto step = [&](int st, int slot, auto set_tag, auto first_tag) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        // stage st has landed when at most the 5 pieces of stage st + 1 are outstanding; the careful edge path may have issued
        // more (younger) operations: a count that is too high only makes this wait stricter
        asm volatile("s_waitcnt vmcnt(5)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");    // (lgkmcnt: the edge path's plain LDS stores)
        acr_barrier_nofence();                              // __syncthreads()'s fence would drain the stages in flight (vmcnt(0))
        const int rslot = slot == 0 ? 2 : slot - 1;         // (st + 2) % 3
        issue(min(st + 2, nst - 1), rslot);                 // past the end: the last stage again, into a slot nobody reads
        if (!compute) return;
        const uint32_t fas = fa + slot * C3W_STAGE_B, fbs = fb + slot * C3W_STAGE_B;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            C3W_RD32(rb[j][0], fbs, 0 * C3_BN * 4 + j * 128); C3W_RD32(rb[j][1], fbs, 1 * C3_BN * 4 + j * 128);
            C3W_RD32(rb[j][2], fbs, 2 * C3_BN * 4 + j * 128); C3W_RD32(rb[j][3], fbs, 3 * C3_BN * 4 + j * 128);
            C3W_RD32(rb[j][4], fbs, 4 * C3_BN * 4 + j * 128); C3W_RD32(rb[j][5], fbs, 5 * C3_BN * 4 + j * 128);
            C3W_RD32(rb[j][6], fbs, 6 * C3_BN * 4 + j * 128); C3W_RD32(rb[j][7], fbs, 7 * C3_BN * 4 + j * 128);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            C3W_RD128(ap[SET][i][0], fas, 0 * C3W_PLANE_B + i * 1024); C3W_RD128(ap[SET][i][1], fas, 1 * C3W_PLANE_B + i * 1024);
            C3W_RD128(ap[SET][i][2], fas, 2 * C3W_PLANE_B + i * 1024);
        }
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[0][2]), "+v"(rb[0][3]), "+v"(rb[0][4]), "+v"(rb[0][5]), "+v"(rb[0][6]),
                     "+v"(rb[0][7]), "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[1][2]), "+v"(rb[1][3]), "+v"(rb[1][4]), "+v"(rb[1][5]), "+v"(rb[1][6]), "+v"(rb[1][7]));
        __builtin_amdgcn_sched_barrier(0);
        const int tap = ((sbeg + st) * C3_BK) / g.C;        // uniform
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = (((j ? valid1 : valid0) >> tap) & 1) != 0;
            f32x4 lo = {rb[j][0], rb[j][1], rb[j][2], rb[j][3]}, hi = {rb[j][4], rb[j][5], rb[j][6], rb[j][7]};
            if (!ok) { lo = f32x4{0.f, 0.f, 0.f, 0.f}; hi = f32x4{0.f, 0.f, 0.f, 0.f}; }
            c3_split3(lo, hi, bp[SET][j][0], bp[SET][j][1], bp[SET][j][2]);
        }
        if (!FIRST) {
            C3W_MFMA6(SET ^ 1, 0, 0) C3W_MFMA6(SET ^ 1, 0, 1) C3W_MFMA6(SET ^ 1, 1, 0) C3W_MFMA6(SET ^ 1, 1, 1)
#pragma unroll
            for (int it = 0; it < 24; ++it) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA of stage st - 1
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // four VALU instructions of stage st's split
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ap[SET][0][0]), "+v"(ap[SET][0][1]), "+v"(ap[SET][0][2]), "+v"(ap[SET][1][0]), "+v"(ap[SET][1][1]), "+v"(ap[SET][1][2]));
    };
    step(0, 0, std::integral_constant<int, 0>{}, std::true_type{});
    int slot = 1;
    for (int st = 1; st < nst; st += 2) {
        step(st, slot, std::integral_constant<int, 1>{}, std::false_type{});
        slot = slot == 2 ? 0 : slot + 1;
        if (st + 1 < nst) {
            step(st + 1, slot, std::integral_constant<int, 0>{}, std::false_type{});
            slot = slot == 2 ? 0 : slot + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the refills past the end (nobody may leave LDS-DMA in flight)
    if (!compute) return;
    if (nst & 1) { C3W_MFMA6(0, 0, 0) C3W_MFMA6(0, 0, 1) C3W_MFMA6(0, 1, 0) C3W_MFMA6(0, 1, 1) }
    else { C3W_MFMA6(1, 0, 0) C3W_MFMA6(1, 0, 1) C3W_MFMA6(1, 1, 0) C3W_MFMA6(1, 1, 1) }
#undef C3W_MFMA6
    float* yb = g.ksplit > 1 ? g.ws + ((int64_t)part * g.nsamp + sample) * g.M * g.HW : g.y + (int64_t)sample * yrows * g.HW;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + acr_krow(e, h);
                if (row < g.M && col < g.HW) yb[(int64_t)row * g.HW + col] = acc[i][j][e];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same kernel for at most 64 output channels (stage 0 of the stem, and CAM generation's large maps): a 128-row tile would
// leave the lower wave row without outputs (two of four waves only feed the DMA), so the tile is 64 rows x 256 PIXELS and all four
// waves compute 64 x 64 on their own 64 pixels; the A stage is the upper half of the image's 128-row block (3 planes x 2 KiB).
// Ring: 3 slots x [A 6 KiB | B fp32 16 KiB]; per wave and stage 2 A pieces (wave 3 re-fetches pieces 4, 5: identical bytes to
// the same place -- keeps the count uniform) + 4 B pieces (one channel row of 256 pixels each) = 6.
// ---------------------------------------------------------------------------------------------------------------------------------
#define C3N_BN 256
#define C3N_A_B (3 * 2048)
#define C3N_STAGE_B (C3N_A_B + C3_BK * C3N_BN * 4)          // 22 KiB
template <bool TAB> __global__ __launch_bounds__(256, 2) void conv3x3_wimg64_kernel(const Conv3Args g) {
    __shared__ __attribute__((aligned(1024))) char smem[C3W_SLOTS * C3N_STAGE_B];       // 66 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, wn = wave;
    const int ntile = g.tiles_n;                            // one tile row
    const int t1 = acr_xcd_remap(blockIdx.x, ntile * g.nsamp * g.ksplit);
    const int part = t1 / (ntile * g.nsamp), t0 = t1 - part * (ntile * g.nsamp);
    const int sample = t0 / ntile, tn = t0 - sample * ntile;
    const int sbeg = part * g.sps;
    const int n0 = tn * C3N_BN;
    const int nkb = (TAB ? g.t.ntap : 9) * g.C / C3_BK;
    const int xrows = TAB ? g.t.xrows : g.C, yrows = TAB ? g.t.yrows : g.M, maxoff = TAB ? g.t.maxoff : g.W + 1;
    const int qa = wave < 3 ? 2 * wave : 4;                 // this wave's two A pieces: qa, qa + 1 (plane q >> 1, KiB q & 1 of its upper half)
    const char* __restrict__ pa = reinterpret_cast<const char*>(g.w) + (int64_t)sbeg * (3 * C3W_PLANE_B) + lane * 16;
    const float* __restrict__ pb = g.x + (int64_t)sample * xrows * g.HW;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int pix = n0 + 4 * lane;
    const int valid0 = c3_valid<TAB>(g, n0 + wn * 64 + r), valid1 = c3_valid<TAB>(g, n0 + wn * 64 + 32 + r);
    const int nst = min(nkb - sbeg, g.sps);
    const int64_t total = (int64_t)g.nsamp * xrows * g.HW;
    const bool edge = (sample == 0 && n0 < maxoff) || (sample == g.nsamp - 1 && n0 + C3N_BN + maxoff > g.HW);
    auto issue = [&](int st, int slot) {
        char* d = smem + slot * C3N_STAGE_B;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = qa + i;
            __builtin_amdgcn_global_load_lds((c3_glb_vp)(pa + (int64_t)st * (3 * C3W_PLANE_B) + (q >> 1) * C3W_PLANE_B + (q & 1) * 1024),
                                             (c3_lds_vp)(d + q * 1024), 16, 0, 0);
        }
        const int k0 = (sbeg + st) * C3_BK;
        const int tap = k0 / g.C;
        int ci0 = k0 - tap * g.C, off;
        if (TAB) { off = g.t.toff[tap]; ci0 += g.t.tcb[tap]; }
        else { const int ty = tap / 3; off = (ty - 1) * g.W + (tap - 3 * ty - 1); }
        float* db = reinterpret_cast<float*>(d + C3N_A_B);
        if (edge) {
            const int64_t i0 = (int64_t)sample * xrows * g.HW + (int64_t)ci0 * g.HW + (pix + off);
#pragma unroll
            for (int i = 0; i < 4; ++i) c3_dma_careful(g.x, i0 + (int64_t)(4 * wave + i) * g.HW, total, db + (4 * wave + i) * C3N_BN, lane);
            return;
        }
        const float* xb = pb + (int64_t)ci0 * g.HW + (pix + off);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((c3_glb_vp)(xb + (int64_t)(4 * wave + i) * g.HW), (c3_lds_vp)(db + (4 * wave + i) * C3N_BN), 16, 0, 0);
    };
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
    const uint32_t fa = lbase + r * 32 + (h ^ ((r >> 3) & 1)) * 16;
    const uint32_t fb = lbase + C3N_A_B + ((8 * h) * C3N_BN + wn * 64 + r) * 4;
    issue(0, 0);
    issue(min(1, nst - 1), 1);
    bf16x8 ap[2][2][3], bp[2][2][3];
    float rb[2][8];
#define C3N_MFMA6(SET, I, J)                                                                                                 \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][2], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][2], bp[SET][J][0], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][1], bp[SET][J][1], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][1], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][1], bp[SET][J][0], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][0], acc[I][J], 0, 0, 0);
    auto step = [&](int st, int slot, auto set_tag, auto first_tag) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        asm volatile("s_waitcnt vmcnt(6)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");      // younger: the 6 pieces of stage st + 1
        acr_barrier_nofence();
        const int rslot = slot == 0 ? 2 : slot - 1;
        issue(min(st + 2, nst - 1), rslot);
        const uint32_t fas = fa + slot * C3N_STAGE_B, fbs = fb + slot * C3N_STAGE_B;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            C3W_RD32(rb[j][0], fbs, 0 * C3N_BN * 4 + j * 128); C3W_RD32(rb[j][1], fbs, 1 * C3N_BN * 4 + j * 128);
            C3W_RD32(rb[j][2], fbs, 2 * C3N_BN * 4 + j * 128); C3W_RD32(rb[j][3], fbs, 3 * C3N_BN * 4 + j * 128);
            C3W_RD32(rb[j][4], fbs, 4 * C3N_BN * 4 + j * 128); C3W_RD32(rb[j][5], fbs, 5 * C3N_BN * 4 + j * 128);
            C3W_RD32(rb[j][6], fbs, 6 * C3N_BN * 4 + j * 128); C3W_RD32(rb[j][7], fbs, 7 * C3N_BN * 4 + j * 128);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            C3W_RD128(ap[SET][i][0], fas, 0 * 2048 + i * 1024); C3W_RD128(ap[SET][i][1], fas, 1 * 2048 + i * 1024);
            C3W_RD128(ap[SET][i][2], fas, 2 * 2048 + i * 1024);
        }
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[0][2]), "+v"(rb[0][3]), "+v"(rb[0][4]), "+v"(rb[0][5]), "+v"(rb[0][6]),
                     "+v"(rb[0][7]), "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[1][2]), "+v"(rb[1][3]), "+v"(rb[1][4]), "+v"(rb[1][5]), "+v"(rb[1][6]), "+v"(rb[1][7]));
        __builtin_amdgcn_sched_barrier(0);
        const int tap = ((sbeg + st) * C3_BK) / g.C;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = (((j ? valid1 : valid0) >> tap) & 1) != 0;
            f32x4 lo = {rb[j][0], rb[j][1], rb[j][2], rb[j][3]}, hi = {rb[j][4], rb[j][5], rb[j][6], rb[j][7]};
            if (!ok) { lo = f32x4{0.f, 0.f, 0.f, 0.f}; hi = f32x4{0.f, 0.f, 0.f, 0.f}; }
            c3_split3(lo, hi, bp[SET][j][0], bp[SET][j][1], bp[SET][j][2]);
        }
        if (!FIRST) {
            C3N_MFMA6(SET ^ 1, 0, 0) C3N_MFMA6(SET ^ 1, 0, 1) C3N_MFMA6(SET ^ 1, 1, 0) C3N_MFMA6(SET ^ 1, 1, 1)
#pragma unroll
            for (int it = 0; it < 24; ++it) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ap[SET][0][0]), "+v"(ap[SET][0][1]), "+v"(ap[SET][0][2]), "+v"(ap[SET][1][0]), "+v"(ap[SET][1][1]), "+v"(ap[SET][1][2]));
    };
    step(0, 0, std::integral_constant<int, 0>{}, std::true_type{});
    int slot = 1;
    for (int st = 1; st < nst; st += 2) {
        step(st, slot, std::integral_constant<int, 1>{}, std::false_type{});
        slot = slot == 2 ? 0 : slot + 1;
        if (st + 1 < nst) {
            step(st + 1, slot, std::integral_constant<int, 0>{}, std::false_type{});
            slot = slot == 2 ? 0 : slot + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (nst & 1) { C3N_MFMA6(0, 0, 0) C3N_MFMA6(0, 0, 1) C3N_MFMA6(0, 1, 0) C3N_MFMA6(0, 1, 1) }
    else { C3N_MFMA6(1, 0, 0) C3N_MFMA6(1, 0, 1) C3N_MFMA6(1, 1, 0) C3N_MFMA6(1, 1, 1) }
#undef C3N_MFMA6
    float* yb = g.ksplit > 1 ? g.ws + ((int64_t)part * g.nsamp + sample) * g.M * g.HW : g.y + (int64_t)sample * yrows * g.HW;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = i * 32 + acr_krow(e, h);
                if (row < g.M && col < g.HW) yb[(int64_t)row * g.HW + col] = acc[i][j][e];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// weight gradient: c[co][t*C + ci] = sum_p dy[co][p] * x[ci][p + off_t] * valid_t(p) per (sample, pixel part) into fp32 slabs,
// summed in a fixed order afterwards.  Both operands are contiguous in the contraction index p (gemm_f32.hip's KC images): A rows
// = dy's channels, B rows = the 9 C (tap, channel) pairs, each an x channel row SHIFTED by its tap's offset (per-lane source
// pointers, 4-byte aligned).  The validity of a tap varies ALONG the contraction here: every lane's fragment is 8 consecutive
// pixels of one (tap, channel) row, handled as two halves of 4 -- image rows are multiples of 4 pixels long, so a half never
// straddles an image row: one row test per half, and only its first (tx = 0) or last (tx = 2) element can touch the column border.
// ---------------------------------------------------------------------------------------------------------------------------------
struct Conv3WgArgs {
    const float* dy;                 // (nsamp, M, HW)
    const float* x;                  // (nsamp, C, HW)
    float* ws;                       // slabs (nsamp * ksplit, M, 9 C)
    int M, C, H, W, HW, nsamp, ksplit, kps;
    int tiles_m, tiles_n;
    C3Tab t;                         // <true> kernels: B rows = the ntap * C (tap, channel) pairs of the table (yrows unused)
};

template <bool TAB> __global__ __launch_bounds__(256, 2) void conv3x3_wgrad_split_kernel(const Conv3WgArgs g) {
    __shared__ __attribute__((aligned(1024))) float smem[C3_SLOTS * 2 * C3_TILE];      // [slot][A | B], 64 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int ntile = g.tiles_m * g.tiles_n, nsplit = g.nsamp * g.ksplit;
    const int t0 = acr_xcd_remap(blockIdx.x, ntile * nsplit);
    const int split = t0 / ntile, tt = t0 - split * ntile;
    const int tm = tt / g.tiles_n, tn = tt - tm * g.tiles_n;
    const int m0 = tm * C3_BM, n0 = tn * C3_BN;
    const int sample = split / g.ksplit;
    const int kbeg = (split - sample * g.ksplit) * g.kps, kend = min(g.HW, kbeg + g.kps);        // host: (kend - kbeg) % 16 == 0
    const int N9 = (TAB ? g.t.ntap : 9) * g.C;
    const int xrows = TAB ? g.t.xrows : g.C, maxoff = TAB ? g.t.maxoff : g.W + 1;
    const float* __restrict__ pa = g.dy + (int64_t)sample * g.M * g.HW;
    const float* __restrict__ pb = g.x + (int64_t)sample * xrows * g.HW;
    const bool compute = m0 + wm * 64 < g.M && n0 + wn * 64 < N9;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // DMA pieces: 16 rows x 64 bytes, lane -> (row = l >> 2, 16-byte chunk l & 3), chunk XOR-swizzled by (row >> 2) & 3 on the source
    int offa[2], offb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + (lane >> 2);
        const int lc4 = ((lane & 3) ^ ((row >> 2) & 3)) << 2;
        offa[i] = min(m0 + row, g.M - 1) * g.HW + lc4;
        const int nr = min(n0 + row, N9 - 1);               // (tap, channel) row of B
        const int tap = nr / g.C, ci = nr - tap * g.C;
        const int ty = tap / 3;
        offb[i] = TAB ? (g.t.tcb[tap] + ci) * g.HW + g.t.toff[tap] + lc4 : ci * g.HW + (ty - 1) * g.W + (tap - 3 * ty - 1) + lc4;
    }
    // the lane's two B fragments: rows n0 + wn*64 + j*32 + r -> their taps
    int tyj[2], txj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nr = min(n0 + wn * 64 + j * 32 + r, N9 - 1);
        const int tap = nr / g.C;
        if (TAB) { tyj[j] = (g.t.tdydx[tap] & 255) - 8; txj[j] = (g.t.tdydx[tap] >> 8) - 8; }      // TAB: the shifts themselves
        else { tyj[j] = tap / 3; txj[j] = tap - 3 * tyj[j]; }
    }
    int xh0 = (kbeg + 8 * h) % g.W, xh1 = (kbeg + 8 * h + 4) % g.W;      // column of the first pixel of the lane's two halves
    const int nst = (kend - kbeg) / C3_BK;
    // the shifted rows of x leave the tensor only in front of sample 0's first pixels and behind the last sample's last ones
    const int64_t total = (int64_t)g.nsamp * xrows * g.HW;
    const bool edge = (sample == 0 && kbeg < maxoff) || (sample == g.nsamp - 1 && kend + maxoff > g.HW);
    auto issue = [&](int st) {
        float* d = smem + (st & (C3_SLOTS - 1)) * 2 * C3_TILE;
        const int k0 = kbeg + st * C3_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((c3_glb_vp)(pa + k0 + offa[i]), (c3_lds_vp)(d + (wave * 2 + i) * 256), 16, 0, 0);
        if (edge) {                                          // uniform (see c3_dma_careful)
            const int64_t i0 = (int64_t)sample * xrows * g.HW + k0;
#pragma unroll
            for (int i = 0; i < 2; ++i) c3_dma_careful(g.x, i0 + offb[i], total, d + C3_TILE + (wave * 2 + i) * 256, lane);
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((c3_glb_vp)(pb + k0 + offb[i]), (c3_lds_vp)(d + C3_TILE + (wave * 2 + i) * 256), 16, 0, 0);
    };
#pragma unroll
    for (int st = 0; st < C3_SLOTS - 1; ++st)
        if (st < nst) issue(st);
    bf16x8 ap[2][2][3], bp[2][2][3];
    f32x4 ra[2][2], rb[2][2];
    auto step = [&](int st, auto set_tag, auto first_tag) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        if (st + 2 < nst) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + C3_SLOTS - 1 < nst) issue(st + C3_SLOTS - 1);
        if (!compute) return;
        const float* sa = smem + (st & (C3_SLOTS - 1)) * 2 * C3_TILE;
        const float* sb = sa + C3_TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + r, sw = (row >> 2) & 3;
            ra[i][0] = *reinterpret_cast<const f32x4*>(sa + row * C3_BK + (((2 * h) ^ sw) << 2));
            ra[i][1] = *reinterpret_cast<const f32x4*>(sa + row * C3_BK + (((2 * h + 1) ^ sw) << 2));
        }
        const int p0 = kbeg + st * C3_BK + 8 * h;           // first pixel of the lane's fragment
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wn * 64 + j * 32 + r, sw = (row >> 2) & 3;
            rb[j][0] = *reinterpret_cast<const f32x4*>(sb + row * C3_BK + (((2 * h) ^ sw) << 2));
            rb[j][1] = *reinterpret_cast<const f32x4*>(sb + row * C3_BK + (((2 * h + 1) ^ sw) << 2));
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int p = p0 + 4 * hf, xc = hf ? xh1 : xh0;
                f32x4 v = rb[j][hf];
                if (TAB) {                                   // any shift: the half's row moves as a whole, its columns one by one
                    const int pr = p + tyj[j] * g.W;
                    if (pr < 0 || pr >= g.HW) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int col = xc + e + txj[j];
                        if (col < 0 || col >= g.W) v[e] = 0.f;
                    }
                } else {
                    const bool yok = tyj[j] == 1 || (tyj[j] == 0 ? p >= g.W : p < g.HW - g.W);
                    const bool kf = txj[j] == 0 && xc == 0, kl = txj[j] == 2 && xc + 4 == g.W;
                    if (!yok) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (kf) v[0] = 0.f;
                    if (kl) v[3] = 0.f;
                }
                rb[j][hf] = v;
            }
        }
        xh0 += C3_BK; if (xh0 >= g.W) xh0 -= g.W;
        xh1 += C3_BK; if (xh1 >= g.W) xh1 -= g.W;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) c3_split3(ra[i][0], ra[i][1], ap[SET][i][0], ap[SET][i][1], ap[SET][i][2]);
#pragma unroll
        for (int j = 0; j < 2; ++j) c3_split3(rb[j][0], rb[j][1], bp[SET][j][0], bp[SET][j][1], bp[SET][j][2]);
        if (!FIRST) {
            C3_ALL(SET ^ 1)
#pragma unroll
            for (int it = 0; it < 24; ++it) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    step(0, std::integral_constant<int, 0>{}, std::true_type{});
    for (int st = 1; st < nst; st += 2) {
        step(st, std::integral_constant<int, 1>{}, std::false_type{});
        if (st + 1 < nst) step(st + 1, std::integral_constant<int, 0>{}, std::false_type{});
    }
    if (!compute) return;
    if (nst & 1) { C3_ALL(0) } else { C3_ALL(1) }
    float* slab = g.ws + (int64_t)split * g.M * N9;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + acr_krow(e, h);
                if (row < g.M && col < N9) slab[(int64_t)row * N9 + col] = acc[i][j][e];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The weight gradient for at most 64 output channels (stage 0's 3x3s: cout = 64, 9 C = 576; the 7x7 stem convolution: 64 x 256):
// in the 128 x 128 tile above the lower wave row has no dy rows and half the workgroup only feeds the DMA.  Here the tile is 64 dy
// rows x 256 (tap, channel) rows: every wave multiplies the same 64 dy rows by its own 64 B rows.  Ring of 3 slots x [A 4 KiB | B
// 16 KiB], DMA two stages ahead, 5 pieces per wave and stage (A piece `wave`, B pieces 4 wave ..), counted vmcnt.
// ---------------------------------------------------------------------------------------------------------------------------------
#define C3G_BN 256
#define C3G_A (64 * C3_BK)                                   // floats
#define C3G_STAGE (C3G_A + C3G_BN * C3_BK)                   // 5120 floats = 20 KiB
template <bool TAB> __global__ __launch_bounds__(256, 2) void conv3x3_wgrad64_kernel(const Conv3WgArgs g) {
    __shared__ __attribute__((aligned(1024))) float smem[3 * C3G_STAGE];               // 60 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int ntile = g.tiles_n, nsplit = g.nsamp * g.ksplit;
    const int t0 = acr_xcd_remap(blockIdx.x, ntile * nsplit);
    const int split = t0 / ntile, tn = t0 - split * ntile;
    const int n0 = tn * C3G_BN;
    const int sample = split / g.ksplit;
    const int kbeg = (split - sample * g.ksplit) * g.kps, kend = min(g.HW, kbeg + g.kps);        // host: (kend - kbeg) % 16 == 0
    const int N9 = (TAB ? g.t.ntap : 9) * g.C;
    const int xrows = TAB ? g.t.xrows : g.C, maxoff = TAB ? g.t.maxoff : g.W + 1;
    const float* __restrict__ pa = g.dy + (int64_t)sample * g.M * g.HW;
    const float* __restrict__ pb = g.x + (int64_t)sample * xrows * g.HW;
    const bool compute = n0 + wave * 64 < N9;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // DMA pieces: 16 rows x 64 bytes, lane -> (row = l >> 2, 16-byte chunk l & 3), chunk XOR-swizzled by (row >> 2) & 3 on the source
    int offa, offb[4];
    {
        const int row = wave * 16 + (lane >> 2);
        offa = min(row, g.M - 1) * g.HW + (((lane & 3) ^ ((row >> 2) & 3)) << 2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 16 + (lane >> 2);
        const int lc4 = ((lane & 3) ^ ((row >> 2) & 3)) << 2;
        const int nr = min(n0 + row, N9 - 1);               // (tap, channel) row of B
        const int tap = nr / g.C, ci = nr - tap * g.C;
        const int ty = tap / 3;
        offb[i] = TAB ? (g.t.tcb[tap] + ci) * g.HW + g.t.toff[tap] + lc4 : ci * g.HW + (ty - 1) * g.W + (tap - 3 * ty - 1) + lc4;
    }
    int tyj[2], txj[2];                                     // the lane's two B fragments: rows n0 + wave*64 + j*32 + r -> their shifts
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nr = min(n0 + wave * 64 + j * 32 + r, N9 - 1);
        const int tap = nr / g.C;
        if (TAB) { tyj[j] = (g.t.tdydx[tap] & 255) - 8; txj[j] = (g.t.tdydx[tap] >> 8) - 8; }
        else { tyj[j] = tap / 3 - 1; txj[j] = tap - 3 * (tap / 3) - 1; }
    }
    int xh0 = (kbeg + 8 * h) % g.W, xh1 = (kbeg + 8 * h + 4) % g.W;      // column of the first pixel of the lane's two halves
    const int nst = (kend - kbeg) / C3_BK;
    const int64_t total = (int64_t)g.nsamp * xrows * g.HW;
    const bool edge = (sample == 0 && kbeg < maxoff) || (sample == g.nsamp - 1 && kend + maxoff > g.HW);
    auto issue = [&](int st, int slot) {
        float* d = smem + slot * C3G_STAGE;
        const int k0 = kbeg + st * C3_BK;
        __builtin_amdgcn_global_load_lds((c3_glb_vp)(pa + k0 + offa), (c3_lds_vp)(d + wave * 256), 16, 0, 0);
        if (edge) {                                          // uniform (see c3_dma_careful)
            const int64_t i0 = (int64_t)sample * xrows * g.HW + k0;
#pragma unroll
            for (int i = 0; i < 4; ++i) c3_dma_careful(g.x, i0 + offb[i], total, d + C3G_A + (wave * 4 + i) * 256, lane);
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((c3_glb_vp)(pb + k0 + offb[i]), (c3_lds_vp)(d + C3G_A + (wave * 4 + i) * 256), 16, 0, 0);
    };
    issue(0, 0);
    issue(min(1, nst - 1), 1);
    bf16x8 ap[2][2][3], bp[2][2][3];
    f32x4 ra[2][2], rb[2][2];
    auto step = [&](int st, int slot, auto set_tag, auto first_tag) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        // stage st has landed when at most the 5 pieces of stage st + 1 are outstanding (the careful path's extra operations only
        // make the wait stricter; its plain LDS stores are covered by lgkmcnt)
        asm volatile("s_waitcnt vmcnt(5)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        acr_barrier_nofence();
        issue(min(st + 2, nst - 1), slot == 0 ? 2 : slot - 1);      // past the end: the last stage again, into a slot nobody reads
        if (!compute) return;
        const float* sa = smem + slot * C3G_STAGE;
        const float* sb = sa + C3G_A;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = i * 32 + r, sw = (row >> 2) & 3;
            ra[i][0] = *reinterpret_cast<const f32x4*>(sa + row * C3_BK + (((2 * h) ^ sw) << 2));
            ra[i][1] = *reinterpret_cast<const f32x4*>(sa + row * C3_BK + (((2 * h + 1) ^ sw) << 2));
        }
        const int p0 = kbeg + st * C3_BK + 8 * h;           // first pixel of the lane's fragment
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wave * 64 + j * 32 + r, sw = (row >> 2) & 3;
            rb[j][0] = *reinterpret_cast<const f32x4*>(sb + row * C3_BK + (((2 * h) ^ sw) << 2));
            rb[j][1] = *reinterpret_cast<const f32x4*>(sb + row * C3_BK + (((2 * h + 1) ^ sw) << 2));
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int p = p0 + 4 * hf, xc = hf ? xh1 : xh0;
                f32x4 v = rb[j][hf];
                const int pr = p + tyj[j] * g.W;             // the half's row moves as a whole, its columns one by one
                if (pr < 0 || pr >= g.HW) v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (TAB) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int col = xc + e + txj[j];
                        if (col < 0 || col >= g.W) v[e] = 0.f;
                    }
                } else {
                    if (txj[j] < 0 && xc == 0) v[0] = 0.f;
                    if (txj[j] > 0 && xc + 4 == g.W) v[3] = 0.f;
                }
                rb[j][hf] = v;
            }
        }
        xh0 += C3_BK; if (xh0 >= g.W) xh0 -= g.W;
        xh1 += C3_BK; if (xh1 >= g.W) xh1 -= g.W;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) c3_split3(ra[i][0], ra[i][1], ap[SET][i][0], ap[SET][i][1], ap[SET][i][2]);
#pragma unroll
        for (int j = 0; j < 2; ++j) c3_split3(rb[j][0], rb[j][1], bp[SET][j][0], bp[SET][j][1], bp[SET][j][2]);
        if (!FIRST) {
            C3_ALL(SET ^ 1)
#pragma unroll
            for (int it = 0; it < 24; ++it) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    step(0, 0, std::integral_constant<int, 0>{}, std::true_type{});
    int slot = 1;
    for (int st = 1; st < nst; st += 2) {
        step(st, slot, std::integral_constant<int, 1>{}, std::false_type{});
        slot = slot == 2 ? 0 : slot + 1;
        if (st + 1 < nst) {
            step(st + 1, slot, std::integral_constant<int, 0>{}, std::false_type{});
            slot = slot == 2 ? 0 : slot + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the refills past the end (nobody may leave LDS-DMA in flight)
    if (!compute) return;
    if (nst & 1) { C3_ALL(0) } else { C3_ALL(1) }
    float* slab = g.ws + (int64_t)split * g.M * N9;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wave * 64 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = i * 32 + acr_krow(e, h);
                if (row < g.M && col < N9) slab[(int64_t)row * N9 + col] = acc[i][j][e];
            }
        }
}

// out[i] = sum_s slab[s][i] in slab order (deterministic), float4 per thread
__global__ __launch_bounds__(256) void conv3x3_reduce_kernel(const float* __restrict__ ws, int nslab, int64_t n4, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 s = reinterpret_cast<const f32x4*>(ws)[i];
    for (int k = 1; k < nslab; ++k) {
        const f32x4 v = reinterpret_cast<const f32x4*>(ws)[(int64_t)k * n4 + i];
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    reinterpret_cast<f32x4*>(out)[i] = s;
}

static int c3_wgrad_ksplit(int nsamp, int cout, int cin, int HW, int ntap = 9) {
    const int tiles = (cout <= 64 ? (ntap * cin + C3G_BN - 1) / C3G_BN : ((cout + C3_BM - 1) / C3_BM) * ((ntap * cin + C3_BN - 1) / C3_BN)) * nsamp;
    // The chip holds 512 of these workgroups (two per CU) and a launch costs whole rounds of them: 288 tiles (128 channels at 56 x 56)
    // as ONE round ran 319 us with 44 % of the slots empty.  Pick the pixel split that minimises
    //     rounds x (32-pixel stages per part + 3 of prologue / epilogue)  +  slab traffic (written and read once, ~4 TB/s, in stage units)
    // with parts of at least 256 pixels; one 32-pixel stage of two co-resident workgroups takes ~3.2 us (scripts/lab/conv_stem_time.py).
    int best = 1;
    double bcost = 1e30;
    for (int ks = 1; ks <= 8 && HW / ks >= 256; ++ks) {
        const int nst = ((HW + ks - 1) / ks + 31) / 32;
        const int rounds = (tiles * ks + 511) / 512;
        const double slab = ks > 1 ? (double)ks * nsamp * cout * ntap * cin * 8.0 / 4e12 / 3.2e-6 : 0.0;
        const double cost = rounds * (nst + 3.0) + slab;
        if (cost < bcost * 0.97) { bcost = cost; best = ks; }
    }
    const int kps = ((HW + best - 1) / best + 31) / 32 * 32;
    return (HW + kps - 1) / kps;
}
extern "C" size_t acr_conv3x3_wgrad_ws_floats(int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W) {
    return (size_t)nsamp * c3_wgrad_ksplit(nsamp, cout, cin, H * W) * cout * 9 * cin;
}

extern "C" int acr_conv3x3_wgrad_f32(int32_t math, const float* dy, const float* x, int32_t nsamp, int32_t cout, int32_t cin, int32_t H,
                                     int32_t W, float* ws, float* dw_packed, void* stream) {
    ACR_CHECK_ARG(dy && x && ws && dw_packed, "acr_conv3x3_wgrad_f32: null pointer");
    if (math != ACR_MATH_BF16X3) {
        acr_set_error("acr_conv3x3_wgrad_f32: built for ACR_MATH_BF16X3 only");
        return ACR_ERR_UNSUPPORTED;
    }
    const int HW = H * W;
    ACR_CHECK_ARG(nsamp > 0 && cout > 0 && cin > 0 && (cout % 4) == 0 && (cin % 4) == 0 && H > 0 && W > 0 && (W % 4) == 0 && W >= C3_BK &&
                      (HW % C3_BK) == 0,
                  "acr_conv3x3_wgrad_f32: need cout, cin %% 4 == 0, W %% 4 == 0, W >= 16, H*W %% 16 == 0 (n=%d co=%d ci=%d %dx%d)", nsamp, cout, cin, H, W);
    ACR_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)ws & 15) == 0 && ((uintptr_t)dw_packed & 15) == 0,
                  "acr_conv3x3_wgrad_f32: 16-byte alignment");
    ACR_CHECK_ARG((int64_t)cout * HW < (1ll << 30) && (int64_t)cin * HW < (1ll << 30), "acr_conv3x3_wgrad_f32: operand too large for 32-bit offsets");
    Conv3WgArgs g;
    g.dy = dy; g.x = x; g.ws = ws; g.M = cout; g.C = cin; g.H = H; g.W = W; g.HW = HW; g.nsamp = nsamp;
    g.ksplit = c3_wgrad_ksplit(nsamp, cout, cin, HW);
    g.kps = ((HW + g.ksplit - 1) / g.ksplit + 31) / 32 * 32;
    const bool wide64 = cout <= 64;                         // 64 x 256 tiles: all four waves compute (conv3x3_wgrad64_kernel)
    g.tiles_m = wide64 ? 1 : (cout + C3_BM - 1) / C3_BM; g.tiles_n = wide64 ? (9 * cin + C3G_BN - 1) / C3G_BN : (9 * cin + C3_BN - 1) / C3_BN;
    const int64_t nwg = (int64_t)g.tiles_m * g.tiles_n * nsamp * g.ksplit;
    ACR_CHECK_ARG(nwg < (1ll << 31), "acr_conv3x3_wgrad_f32: grid too large");
    hipStream_t st = (hipStream_t)stream;
    if (wide64) hipLaunchKernelGGL(conv3x3_wgrad64_kernel<false>, dim3((unsigned)nwg), dim3(256), 0, st, g);
    else hipLaunchKernelGGL(conv3x3_wgrad_split_kernel<false>, dim3((unsigned)nwg), dim3(256), 0, st, g);
    const int64_t n4 = (int64_t)cout * 9 * cin / 4;
    if (!acr_slab_sum_wide(ws, nsamp * g.ksplit, n4, dw_packed, st))
        hipLaunchKernelGGL(conv3x3_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, nsamp * g.ksplit, n4,
                           dw_packed);
    return acr_check_launch("acr_conv3x3_wgrad_f32");
}

// Small launches (CAM generation: two views of one image, 20-72 workgroups walking 144 stages each) are split along the
// contraction into up to 512 workgroups' worth of parts of at least 8 stages; the parts' slabs are summed in part order.
static int c3_fwd_ksplit(int nsamp, int cout, int cin, int HW, bool wide64 = false, int ntap = 9) {
    const int tiles = wide64 ? ((HW + 255) / 256) * nsamp : ((cout + C3_BM - 1) / C3_BM) * ((HW + C3_BN - 1) / C3_BN) * nsamp;
    const int nst = ntap * cin / C3_BK;
    if (tiles >= 192) return 1;
    int ks = 512 / tiles;
    if (ks > nst / 8) ks = nst / 8;
    if (ks < 2) return 1;
    const int sps = (nst + ks - 1) / ks;
    return (nst + sps - 1) / sps;
}
extern "C" size_t acr_conv3x3_ws_floats(int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W) {
    int ks = c3_fwd_ksplit(nsamp, cout, cin, H * W);
    if (cout <= 64) ks = max(ks, c3_fwd_ksplit(nsamp, cout, cin, H * W, true));      // acr_conv3x3_x3's 64 x 256 tiling
    return ks > 1 ? (size_t)ks * nsamp * cout * H * W : 0;
}
extern "C" int acr_conv3x3_f32(int32_t math, const float* w_packed, const float* x, float* y, int32_t nsamp, int32_t cout, int32_t cin,
                               int32_t H, int32_t W, float* ws, void* stream) {
    ACR_CHECK_ARG(w_packed && x && y, "acr_conv3x3_f32: null pointer");
    if (math != ACR_MATH_BF16X3) {
        acr_set_error("acr_conv3x3_f32: built for ACR_MATH_BF16X3 only (the exact-fp32 arithmetic keeps the library's Winograd kernels: "
                      "F(2,3) does 2.25x fewer multiplies than an implicit GEMM on the fp32 MFMA)");
        return ACR_ERR_UNSUPPORTED;
    }
    ACR_CHECK_ARG(nsamp > 0 && cout > 0 && cin > 0 && (cin % C3_BK) == 0 && H > 0 && W > 0 && ((int64_t)H * W) % 4 == 0 && (int64_t)H * W >= 4,
                  "acr_conv3x3_f32: need cin %% 16 == 0 and H*W %% 4 == 0 (n=%d co=%d ci=%d %dx%d)", nsamp, cout, cin, H, W);
    ACR_CHECK_ARG(((uintptr_t)w_packed & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "acr_conv3x3_f32: 16-byte alignment");
    ACR_CHECK_ARG((int64_t)cout * 9 * cin < (1ll << 30) && (int64_t)cin * H * W < (1ll << 30), "acr_conv3x3_f32: operand too large for 32-bit offsets");
    Conv3Args g;
    g.w = w_packed; g.x = x; g.y = y; g.M = cout; g.C = cin; g.H = H; g.W = W; g.HW = H * W; g.nsamp = nsamp;
    g.tiles_m = (cout + C3_BM - 1) / C3_BM; g.tiles_n = (g.HW + C3_BN - 1) / C3_BN;
    const int nst = 9 * cin / C3_BK;
    g.ksplit = (ws && ((uintptr_t)ws & 15) == 0) ? c3_fwd_ksplit(nsamp, cout, cin, g.HW) : 1;
    g.sps = (nst + g.ksplit - 1) / g.ksplit; g.ws = ws;
    const int64_t nwg = (int64_t)g.tiles_m * g.tiles_n * nsamp * g.ksplit;
    ACR_CHECK_ARG(nwg < (1ll << 31), "acr_conv3x3_f32: grid too large");
    hipLaunchKernelGGL(conv3x3_split_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, g);
    if (g.ksplit > 1) {
        const int64_t n4 = (int64_t)nsamp * cout * g.HW / 4;
        hipLaunchKernelGGL(conv3x3_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, g.ksplit, n4, y);
    }
    return acr_check_launch("acr_conv3x3_f32");
}

// acr_conv3x3_f32 with the packed weight as a split-product image (acr_x3_image of w_packed: rows = cout, cols = 9 cin): same
// contract, same workspace (acr_conv3x3_ws_floats), conv3x3_wimg_kernel.
extern "C" int acr_conv3x3_x3(const float* w_img, const float* x, float* y, int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W,
                              float* ws, void* stream) {
    ACR_CHECK_ARG(w_img && x && y, "acr_conv3x3_x3: null pointer");
    ACR_CHECK_ARG(nsamp > 0 && cout > 0 && cin > 0 && (cin % C3_BK) == 0 && H > 0 && W > 0 && ((int64_t)H * W) % 4 == 0 && (int64_t)H * W >= 4,
                  "acr_conv3x3_x3: need cin %% 16 == 0 and H*W %% 4 == 0 (n=%d co=%d ci=%d %dx%d)", nsamp, cout, cin, H, W);
    ACR_CHECK_ARG(((uintptr_t)w_img & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "acr_conv3x3_x3: 16-byte alignment");
    ACR_CHECK_ARG((int64_t)cin * H * W < (1ll << 30), "acr_conv3x3_x3: sample too large for 32-bit offsets");
    Conv3Args g;
    g.w = w_img; g.x = x; g.y = y; g.M = cout; g.C = cin; g.H = H; g.W = W; g.HW = H * W; g.nsamp = nsamp;
    const bool wide64 = cout <= 64;                         // 64 x 256 tiles: all four waves compute (conv3x3_wimg64_kernel)
    g.tiles_m = wide64 ? 1 : (cout + C3_BM - 1) / C3_BM; g.tiles_n = wide64 ? (g.HW + 255) / 256 : (g.HW + C3_BN - 1) / C3_BN;
    const int nst = 9 * cin / C3_BK;
    g.ksplit = (ws && ((uintptr_t)ws & 15) == 0) ? c3_fwd_ksplit(nsamp, cout, cin, g.HW, wide64) : 1;
    g.sps = (nst + g.ksplit - 1) / g.ksplit; g.ws = ws;
    const int64_t nwg = (int64_t)g.tiles_m * g.tiles_n * nsamp * g.ksplit;
    ACR_CHECK_ARG(nwg < (1ll << 31), "acr_conv3x3_x3: grid too large");
    if (wide64) hipLaunchKernelGGL(conv3x3_wimg64_kernel<false>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(conv3x3_wimg_kernel<false>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, g);
    if (g.ksplit > 1) {
        const int64_t n4 = (int64_t)nsamp * cout * g.HW / 4;
        hipLaunchKernelGGL(conv3x3_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, g.ksplit, n4, y);
    }
    return acr_check_launch("acr_conv3x3_x3");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Strided SAME convolutions (the 7x7 stride-2 stem convolution, models/resnetv2.py:337-340; conv2 of the first bottleneck of stages
// 1 and 2, :196-199 with stride 2) as tap-table products over a space-to-depth copy of the input -- the last library code on the
// f32_split step.  acr_space_to_depth2_f32: xs[n][(py*2+px)*C + c][y][x] = x[n][c][2y+py][2x+px], zero past the
// image, H2 = ceil(H/2), W2 = ceil(W/2); channel rows 4C .. xrows-1 of every sample are zero (the 7x7 convolution's stage of 16
// rows = 12 phase-channels + 4).  acr_depth_to_space2_f32 is its inverse (the input gradient's last step; H even, W % 8 == 0).
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool VEC> __global__ __launch_bounds__(256) void c3_s2d_kernel(const float* __restrict__ x, float* __restrict__ xs, int C, int H, int W,
                                                                          int H2, int W2, int xrows, int64_t nthr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nthr) return;
    const int wq = (W2 + 3) >> 2;
    const int q = (int)(i % wq);
    int64_t t = i / wq;
    const int y = (int)(t % H2);
    t /= H2;
    const int c = (int)(t % C), n = (int)(t / C);
    const float* __restrict__ xr = x + ((int64_t)n * C + c) * H * W;
    float v[2][8];
#pragma unroll
    for (int py = 0; py < 2; ++py) {
        const int yy = 2 * y + py;
        if (VEC) {                                           // W % 8 == 0: whole 32-byte groups
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
            if (yy < H) {
                a = *reinterpret_cast<const f32x4*>(xr + (int64_t)yy * W + 8 * q);
                b = *reinterpret_cast<const f32x4*>(xr + (int64_t)yy * W + 8 * q + 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[py][e] = a[e]; v[py][4 + e] = b[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[py][e] = (yy < H && 8 * q + e < W) ? xr[(int64_t)yy * W + 8 * q + e] : 0.f;
        }
    }
    const int64_t plane = (int64_t)H2 * W2;
    float* __restrict__ o = xs + ((int64_t)n * xrows + c) * plane + (int64_t)y * W2 + 4 * q;
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            float* d = o + (int64_t)(py * 2 + px) * C * plane;
            if (VEC) *reinterpret_cast<f32x4*>(d) = f32x4{v[py][px], v[py][px + 2], v[py][px + 4], v[py][px + 6]};
            else
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < W2) d[e] = v[py][px + 2 * e];
        }
}
// rows [r0, xrows) of every sample := 0, float4 per thread (plane % 4 == 0)
__global__ __launch_bounds__(256) void c3_zero_rows_kernel(float* __restrict__ xs, int xrows, int r0, int64_t plane4, int64_t nthr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nthr) return;
    const int64_t per = (int64_t)(xrows - r0) * plane4;
    const int64_t n = i / per, j = i - n * per;
    reinterpret_cast<f32x4*>(xs)[(n * xrows + r0) * plane4 + j] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__global__ __launch_bounds__(256) void c3_d2s_kernel(const float* __restrict__ xs, float* __restrict__ x, int C, int H, int W, int64_t nthr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nthr) return;
    const int H2 = H >> 1, W2 = W >> 1, wq = W2 >> 2;
    const int q = (int)(i % wq);
    int64_t t = i / wq;
    const int y = (int)(t % H2);
    t /= H2;
    const int c = (int)(t % C), n = (int)(t / C);
    const int64_t plane = (int64_t)H2 * W2;
    const float* __restrict__ s = xs + ((int64_t)n * 4 * C + c) * plane + (int64_t)y * W2 + 4 * q;
    float* __restrict__ d = x + ((int64_t)n * C + c) * H * W + (int64_t)(2 * y) * W + 8 * q;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(s + (int64_t)(py * 2) * C * plane);
        const f32x4 b = *reinterpret_cast<const f32x4*>(s + (int64_t)(py * 2 + 1) * C * plane);
        *reinterpret_cast<f32x4*>(d + (int64_t)py * W) = f32x4{a[0], b[0], a[1], b[1]};
        *reinterpret_cast<f32x4*>(d + (int64_t)py * W + 4) = f32x4{a[2], b[2], a[3], b[3]};
    }
}

extern "C" int acr_space_to_depth2_f32(const float* x, float* xs, int32_t nsamp, int32_t C, int32_t H, int32_t W, int32_t xrows, void* stream) {
    ACR_CHECK_ARG(x && xs, "acr_space_to_depth2_f32: null pointer");
    ACR_CHECK_ARG(nsamp > 0 && C > 0 && H > 0 && W > 0 && xrows >= 4 * C, "acr_space_to_depth2_f32: bad shape (n=%d C=%d %dx%d rows=%d)", nsamp, C, H, W, xrows);
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const bool vec = (W % 8) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)xs & 15) == 0;
    ACR_CHECK_ARG(xrows == 4 * C || ((int64_t)H2 * W2) % 4 == 0 && ((uintptr_t)xs & 15) == 0, "acr_space_to_depth2_f32: zero rows need (H2*W2) %% 4 == 0");
    const int64_t nthr = (int64_t)nsamp * C * H2 * ((W2 + 3) / 4);
    ACR_CHECK_ARG((nthr + 255) / 256 < (1ll << 31), "acr_space_to_depth2_f32: grid too large");
    hipStream_t st = (hipStream_t)stream;
    if (vec) hipLaunchKernelGGL(c3_s2d_kernel<true>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, x, xs, C, H, W, H2, W2, xrows, nthr);
    else hipLaunchKernelGGL(c3_s2d_kernel<false>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, x, xs, C, H, W, H2, W2, xrows, nthr);
    if (xrows > 4 * C) {
        const int64_t plane4 = (int64_t)H2 * W2 / 4, nz = (int64_t)nsamp * (xrows - 4 * C) * plane4;
        hipLaunchKernelGGL(c3_zero_rows_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, xs, xrows, 4 * C, plane4, nz);
    }
    return acr_check_launch("acr_space_to_depth2_f32");
}
extern "C" int acr_depth_to_space2_f32(const float* xs, float* x, int32_t nsamp, int32_t C, int32_t H, int32_t W, void* stream) {
    ACR_CHECK_ARG(x && xs, "acr_depth_to_space2_f32: null pointer");
    ACR_CHECK_ARG(nsamp > 0 && C > 0 && H > 0 && W > 0 && (H % 2) == 0 && (W % 8) == 0, "acr_depth_to_space2_f32: need H even, W %% 8 == 0 (n=%d C=%d %dx%d)", nsamp, C, H, W);
    ACR_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)xs & 15) == 0, "acr_depth_to_space2_f32: 16-byte alignment");
    const int64_t nthr = (int64_t)nsamp * C * (H / 2) * (W / 8);
    ACR_CHECK_ARG((nthr + 255) / 256 < (1ll << 31), "acr_depth_to_space2_f32: grid too large");
    hipLaunchKernelGGL(c3_d2s_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xs, x, C, H, W, nthr);
    return acr_check_launch("acr_depth_to_space2_f32");
}

static int c3_fill_tab(C3Tab& t, const char* who, int ntap, const int32_t* tdy, const int32_t* tdx, const int32_t* tcb, int cin, int W, int xrows, int yrows) {
    ACR_CHECK_ARG(ntap > 0 && ntap <= C3_MAXTAP && tdy && tdx && tcb, "%s: need 1 .. %d taps", who, C3_MAXTAP);
    t.ntap = ntap; t.xrows = xrows; t.yrows = yrows; t.maxoff = 0;
    for (int i = 0; i < C3_MAXTAP; ++i) { t.toff[i] = 0; t.tcb[i] = 0; t.tdydx[i] = 8 | (8 << 8); }
    for (int i = 0; i < ntap; ++i) {
        ACR_CHECK_ARG(tdy[i] >= -7 && tdy[i] <= 7 && tdx[i] >= -7 && tdx[i] <= 7 && tcb[i] >= 0 && tcb[i] + cin <= xrows,
                      "%s: tap %d (dy=%d dx=%d rows %d..%d of %d) outside what the table encodes", who, i, tdy[i], tdx[i], tcb[i], tcb[i] + cin, xrows);
        t.toff[i] = tdy[i] * W + tdx[i]; t.tcb[i] = tcb[i]; t.tdydx[i] = (tdy[i] + 8) | ((tdx[i] + 8) << 8);
        const int a = t.toff[i] < 0 ? -t.toff[i] : t.toff[i];
        if (a > t.maxoff) t.maxoff = a;
    }
    return ACR_OK;
}

// y[n][yrow][p] = sum_t sum_c w_img[(co, t * cin + c)] * x[n][tcb[t] + c][p + tdy[t] * W + tdx[t]] * inside(p, t) for yrow < cout:
// the tap-table form of acr_conv3x3_x3 (same kernels, same image layout with 9 -> ntap).  x holds `xrows` channel rows per sample,
// y `yrows` (y may point into a channel slice of a larger tensor).  tdy / tdx / tcb are HOST arrays of ntap entries.
extern "C" size_t acr_conv_taps_ws_floats(int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W, int32_t ntap) {
    const int ks = c3_fwd_ksplit(nsamp, cout, cin, H * W, cout <= 64, ntap);
    return ks > 1 ? (size_t)ks * nsamp * cout * H * W : 0;
}
extern "C" int acr_conv_taps_x3(const float* w_img, const float* x, float* y, int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W,
                                int32_t ntap, const int32_t* tdy, const int32_t* tdx, const int32_t* tcb, int32_t xrows, int32_t yrows, float* ws,
                                void* stream) {
    ACR_CHECK_ARG(w_img && x && y, "acr_conv_taps_x3: null pointer");
    ACR_CHECK_ARG(nsamp > 0 && cout > 0 && cin > 0 && (cin % C3_BK) == 0 && H > 0 && W > 0 && ((int64_t)H * W) % 4 == 0 && yrows >= cout,
                  "acr_conv_taps_x3: need cin %% 16 == 0, H*W %% 4 == 0, yrows >= cout (n=%d co=%d ci=%d %dx%d)", nsamp, cout, cin, H, W);
    ACR_CHECK_ARG(((uintptr_t)w_img & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "acr_conv_taps_x3: 16-byte alignment");
    ACR_CHECK_ARG((int64_t)xrows * H * W < (1ll << 30), "acr_conv_taps_x3: sample too large for 32-bit offsets");
    Conv3Args g;
    const int rc = c3_fill_tab(g.t, "acr_conv_taps_x3", ntap, tdy, tdx, tcb, cin, W, xrows, yrows);
    if (rc != ACR_OK) return rc;
    g.w = w_img; g.x = x; g.y = y; g.M = cout; g.C = cin; g.H = H; g.W = W; g.HW = H * W; g.nsamp = nsamp;
    const bool wide64 = cout <= 64;
    g.tiles_m = wide64 ? 1 : (cout + C3_BM - 1) / C3_BM; g.tiles_n = wide64 ? (g.HW + 255) / 256 : (g.HW + C3_BN - 1) / C3_BN;
    const int nst = ntap * cin / C3_BK;
    g.ksplit = (ws && ((uintptr_t)ws & 15) == 0 && yrows == cout) ? c3_fwd_ksplit(nsamp, cout, cin, g.HW, wide64, ntap) : 1;
    g.sps = (nst + g.ksplit - 1) / g.ksplit; g.ws = ws;
    const int64_t nwg = (int64_t)g.tiles_m * g.tiles_n * nsamp * g.ksplit;
    ACR_CHECK_ARG(nwg < (1ll << 31), "acr_conv_taps_x3: grid too large");
    if (wide64) hipLaunchKernelGGL(conv3x3_wimg64_kernel<true>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(conv3x3_wimg_kernel<true>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, g);
    if (g.ksplit > 1) {
        const int64_t n4 = (int64_t)nsamp * cout * g.HW / 4;
        hipLaunchKernelGGL(conv3x3_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, g.ksplit, n4, y);
    }
    return acr_check_launch("acr_conv_taps_x3");
}

// dw_packed[co][t * cin + c] = sum_n sum_p dy[n][co][p] * x[n][tcb[t] + c][p + tdy[t] * W + tdx[t]] * inside(p, t): the tap-table form
// of acr_conv3x3_wgrad_f32 (same kernel, slabs summed in a fixed order).
extern "C" size_t acr_conv_taps_wgrad_ws_floats(int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W, int32_t ntap) {
    return (size_t)nsamp * c3_wgrad_ksplit(nsamp, cout, cin, H * W, ntap) * cout * ntap * cin;
}
extern "C" int acr_conv_taps_wgrad_f32(int32_t math, const float* dy, const float* x, int32_t nsamp, int32_t cout, int32_t cin, int32_t H, int32_t W,
                                       int32_t ntap, const int32_t* tdy, const int32_t* tdx, const int32_t* tcb, int32_t xrows, float* ws,
                                       float* dw_packed, void* stream) {
    ACR_CHECK_ARG(dy && x && ws && dw_packed, "acr_conv_taps_wgrad_f32: null pointer");
    if (math != ACR_MATH_BF16X3) {
        acr_set_error("acr_conv_taps_wgrad_f32: built for ACR_MATH_BF16X3 only");
        return ACR_ERR_UNSUPPORTED;
    }
    const int HW = H * W;
    ACR_CHECK_ARG(nsamp > 0 && cout > 0 && cin > 0 && (cout % 4) == 0 && (cin % 4) == 0 && H > 0 && W > 0 && (W % 4) == 0 && W >= C3_BK &&
                      (HW % C3_BK) == 0,
                  "acr_conv_taps_wgrad_f32: need cout, cin %% 4 == 0, W %% 4 == 0, W >= 16, H*W %% 16 == 0 (n=%d co=%d ci=%d %dx%d)", nsamp, cout, cin, H, W);
    ACR_CHECK_ARG(((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)ws & 15) == 0 && ((uintptr_t)dw_packed & 15) == 0,
                  "acr_conv_taps_wgrad_f32: 16-byte alignment");
    ACR_CHECK_ARG((int64_t)cout * HW < (1ll << 30) && (int64_t)xrows * HW < (1ll << 30), "acr_conv_taps_wgrad_f32: operand too large for 32-bit offsets");
    Conv3WgArgs g;
    const int rc = c3_fill_tab(g.t, "acr_conv_taps_wgrad_f32", ntap, tdy, tdx, tcb, cin, W, xrows, 0);
    if (rc != ACR_OK) return rc;
    g.dy = dy; g.x = x; g.ws = ws; g.M = cout; g.C = cin; g.H = H; g.W = W; g.HW = HW; g.nsamp = nsamp;
    g.ksplit = c3_wgrad_ksplit(nsamp, cout, cin, HW, ntap);
    g.kps = ((HW + g.ksplit - 1) / g.ksplit + 31) / 32 * 32;
    const bool wide64 = cout <= 64;
    g.tiles_m = wide64 ? 1 : (cout + C3_BM - 1) / C3_BM; g.tiles_n = wide64 ? (ntap * cin + C3G_BN - 1) / C3G_BN : (ntap * cin + C3_BN - 1) / C3_BN;
    const int64_t nwg = (int64_t)g.tiles_m * g.tiles_n * nsamp * g.ksplit;
    ACR_CHECK_ARG(nwg < (1ll << 31), "acr_conv_taps_wgrad_f32: grid too large");
    hipStream_t st = (hipStream_t)stream;
    if (wide64) hipLaunchKernelGGL(conv3x3_wgrad64_kernel<true>, dim3((unsigned)nwg), dim3(256), 0, st, g);
    else hipLaunchKernelGGL(conv3x3_wgrad_split_kernel<true>, dim3((unsigned)nwg), dim3(256), 0, st, g);
    const int64_t n4 = (int64_t)cout * ntap * cin / 4;
    if (!acr_slab_sum_wide(ws, nsamp * g.ksplit, n4, dw_packed, st))
        hipLaunchKernelGGL(conv3x3_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, nsamp * g.ksplit, n4, dw_packed);
    return acr_check_launch("acr_conv_taps_wgrad_f32");
}
