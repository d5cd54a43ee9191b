// fp32 GEMMs of the transformer blocks at the REFERENCE precision (train_acr.py:137 runs fp32 end to end):
//   NT  y  = x W^T + b (+ resid)      models/vision_transformer.py:158-164 (fc1, fc2), :200 (qkv), :212 (proj)
//   NN  dx = dy W                      their input gradients
//   TN  dW = dy^T x, db = colsum(dy)   their weight / bias gradients (contraction over all tokens, split over workgroups)
// on v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate (a k-ordered fmaf chain, MI355X_MICROARCH.md "Matrix
// cores"), 64 FLOP/clk/SIMD = 157 TFLOP/s.
//
// Design for gfx950.  The fp32 MFMA needs ONE operand register per 4096 FLOP, i.e. 1/8 of the operand bytes per cycle of
// the bf16 32x32x16 form, so neither LDS bandwidth nor the L2 -> LDS path limits it (128x128 tiles: 1 ds_read_b128 per 4
// MFMAs = 256 cycles; 32 KiB of operands per 4096 MFMA cycles per workgroup).  What matters instead is (a) never leaving
// the matrix pipe idle and (b) tile quantisation: a 25 120-token activation against N = 768 is 591 tiles of 128x256 but
// only 297 of 256x256 for 256 CUs.  Hence: 128x128x32 tiles, 256 threads = 4 waves of 64x64 (4 accumulators of 16
// registers), TWO workgroups per CU so that each SIMD holds two waves from different workgroups: while one sits in its
// barrier / LDS refill the other issues MFMAs.  Operands are staged global -> registers -> LDS one K-chunk ahead (the global
// loads of chunk c+1 are in flight during the 64 MFMAs of chunk c) with two LDS buffers and one barrier per chunk.
//
// Operand layouts.  "KC" = contraction index contiguous in memory (x, W in NT): LDS image [i][k] with a 36-float pitch --
// ds_read_b128 fragment reads are bank-conflict free for every 16-lane group of the instruction (36*r mod 64 is a
// permutation of the 16 four-bank slots over any 16 consecutive r).  "KS" = contraction index strided (W in NN, dy and x
// in TN): LDS image [k][i], fragment = 4 ds_read_b32 of 32 consecutive floats.  No transposed copies of any weight.
// The 32x32x2 MFMA sums over k in any order as long as A and B agree: lane half h takes k = 8q + 4h + s of a chunk in
// step (q, s), which is what makes one 16-byte read feed four MFMAs.
#include <math.h>

#include <type_traits>

#include "acr_common.h"

typedef __bf16 bf16_t;

#define F_BM 128
#define F_BN 128
#define F_BK 32
#define F_PKC 36                 // [i][k] pitch (floats)
#define F_PKS 128                // [k][i] pitch (floats)
#define F_STAGE (F_BM * F_PKC)   // floats per operand per stage (>= F_BK * F_PKS = 4096)

struct GemmF32Args {
    const float* a; int64_t lda;
    const float* b; int64_t ldb;
    const float* bias;             // (N) or null
    const float* aux; int64_t ldaux;   // resid (ACT 0) / saved pre-activation h (ACT 2), (M,N) or null
    float* c; int64_t ldc;
    float* c2;                     // ACT 1: GELU(c), same pitch
    float* cs;                     // TN: per-split column sums of A (bias gradient slabs) or null
    int M, N, K;
    int tiles_m, tiles_n, nsplit, kps;   // kps: contraction elements per split (multiple of F_BK)
    int tile0, tiles_launch;             // this launch covers tiles tile0 .. tile0 + tiles_launch - 1 (each nsplit times)
    int nkb_a, nkb_b;                    // gemm_f32_planes_tn_kernel: stages (16 features) per token block of the a / b image
    int img_nkb;                         // image epilogues (ACT 5, 6): stages per row block of the OUTPUT image c2 points at (ceil(N / 16))
    // z-slices: workgroup slice z = split index.  K-split (weight gradient of a Linear): operands shared, k range z*k_zs..;
    // batch (1x1 convolutions per sample): operands / outputs advance by *_zs per slice, k range the whole contraction
    int64_t a_zs, b_zs, c_zs, aux_zs;
    int k_zs, ksplit;                    // slice z = split / ksplit (operand / output offsets), contraction part split % ksplit
};

// Tile order of the NT / NN products: bands of 8 tile rows, column-major inside a band, so that the 64 workgroups an XCD
// holds at a time (it walks one contiguous range of this order, acr_xcd_remap) form an 8 x 8 block of tiles: 8 + 8 operand
// panels (6.3 MB at K = 768) per 64 tiles instead of one A panel + ALL B panels per tile row (N = 3072: the 9.4 MB weight
// exceeds one XCD's 4 MB L2 and was re-fetched for every tile row: 1.04 GB fetched for 86 MB of operands).
#define F_BAND 8
__device__ __forceinline__ void tile_coords(int tt, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int per_band = F_BAND * tiles_n;
    const int band = tt / per_band, in_band = tt - band * per_band;
    const int first = band * F_BAND;
    const int rows = min(tiles_m - first, F_BAND);
    tn = in_band / rows;
    tm = first + (in_band - tn * rows);
}

// one K-chunk of one operand, global -> registers (4 float4 per thread), addresses clamped into the matrix so that every
// load is unconditional and nothing touches the loaded registers before store_chunk (the loads stay in flight across the
// chunk's MFMAs).  KC: rows = the operand's non-contraction index, k contiguous; KS: rows = k, 128 contiguous elements.
template <bool KC>
__device__ __forceinline__ void load_chunk(f32x4 (&r)[4], const float* __restrict__ p, int64_t ld, int i0, int dim, int k0,
                                           int kend, int tid) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int f = tid + 256 * ps;
        if (KC) {
            const int i = min(i0 + (f >> 3), dim - 1), k = min(k0 + 4 * (f & 7), kend - 4);
            r[ps] = *reinterpret_cast<const f32x4*>(p + (int64_t)i * ld + k);
        } else {
            const int k = min(k0 + (f >> 5), kend - 1), i = min(i0 + 4 * (f & 31), dim - 4);
            r[ps] = *reinterpret_cast<const f32x4*>(p + (int64_t)k * ld + i);
        }
    }
}

// registers -> LDS; contraction indices at or beyond kend are stored as zeros (only the last chunk of a split has any)
template <bool KC>
__device__ __forceinline__ void store_chunk(float* __restrict__ s, f32x4 (&r)[4], int k0, int kend, int tid) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int f = tid + 256 * ps;
        if (KC) {
            if (k0 + 4 * (f & 7) >= kend) r[ps] = z;
            *reinterpret_cast<f32x4*>(s + (f >> 3) * F_PKC + 4 * (f & 7)) = r[ps];
        } else {
            if (k0 + (f >> 5) >= kend) r[ps] = z;
            *reinterpret_cast<f32x4*>(s + (f >> 5) * F_PKS + 4 * (f & 31)) = r[ps];
        }
    }
}

// fragment of 32 rows starting at `base` for k-group q of the chunk: v[s] = T[base + r][8q + 4h + s]
template <bool KC>
__device__ __forceinline__ f32x4 read_frag(const float* __restrict__ s, int base, int q, int r, int h) {
    if (KC) return *reinterpret_cast<const f32x4*>(s + (base + r) * F_PKC + 8 * q + 4 * h);
    const float* p = s + (8 * q + 4 * h) * F_PKS + base + r;
    f32x4 v = {p[0], p[F_PKS], p[2 * F_PKS], p[3 * F_PKS]};
    return v;
}

// epilogue of one wave's 64x64 block (rows mb.., columns nb..): lane (r, h), register e of a 32x32 accumulator = row
// krow(e, h), column r.  EDGE = false: the tile is interior, every access is unconditional (loads batch, no branches).
template <int ACT, bool EDGE>
__device__ __forceinline__ void epilogue_f32(const GemmF32Args& g, f32x16 (&acc)[2][2], int mb, int nb, int r, int h) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = nb + j * 32 + r;
        const bool cok = !EDGE || col < g.N;
        const float bj = (ACT != 2 && g.bias && cok) ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float x[16];
            if (ACT == 2 || (ACT == 0 && g.aux)) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = mb + i * 32 + acr_krow(e, h);
                    x[e] = (!EDGE || (row < g.M && cok)) ? g.aux[(int64_t)row * g.ldaux + col] : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = mb + i * 32 + acr_krow(e, h);
                if (EDGE && !(row < g.M && cok)) continue;
                float v = acc[i][j][e] + bj;
                float* cp = g.c + (int64_t)row * g.ldc + col;
                if (ACT == 0) {
                    *cp = g.aux ? v + x[e] : v;
                } else if (ACT == 1) {                          // one erff serves GELU and GELU'
                    const float er = erff(v * 0.70710678118654752440f);
                    g.c2[(int64_t)row * g.ldc + col] = v * 0.5f * (1.0f + er);
                    *cp = 0.5f * (1.0f + er) + v * (expf(-0.5f * v * v) * 0.39894228040143267794f);
                } else {
                    *cp = v * x[e];
                }
            }
        }
    }
}

// ACT: 0 = (+bias)(+resid), 1 = c = GELU'(h), c2 = GELU(h) with h = acc + bias, 2 = c = acc * aux, 3 = split slab (no epilogue)
template <bool A_KC, bool B_KC, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmF32Args g) {
    __shared__ __attribute__((aligned(16))) float smem[4 * F_STAGE];      // [buf][A|B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int ntile = g.tiles_launch;
    const int t = acr_xcd_remap(blockIdx.x, ntile * g.nsplit);
    const int split = t / ntile, tt = g.tile0 + (t - split * ntile);
    int tm, tn;
    if (ACT == 3) { tm = tt / g.tiles_n; tn = tt - tm * g.tiles_n; }
    else tile_coords(tt, g.tiles_m, g.tiles_n, tm, tn);
    const int m0 = tm * F_BM, n0 = tn * F_BN;
    const int zs = split / g.ksplit;
    const int kbeg = (split - zs * g.ksplit) * g.k_zs, kend = min(g.K, kbeg + g.kps);
    const float* __restrict__ pa = g.a + (int64_t)zs * g.a_zs;
    const float* __restrict__ pb = g.b + (int64_t)zs * g.b_zs;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Register sets 0 / 1 hold the chunks in flight from global memory.  Chunk c is computed from LDS buffer c & 1 while
    // chunk c + 1 (loaded during chunk c - 1) waits in set (c + 1) & 1 and the loads of chunk c + 2 are issued into set
    // c & 1: every load has TWO chunks of MFMAs (>= 8192 matrix-pipe cycles) to land before its LDS store.
    f32x4 ra0[4], rb0[4], ra1[4], rb1[4];
    float csum[4] = {0.f, 0.f, 0.f, 0.f};          // TN bias gradient: this thread's column sums of its A chunks (KS layout)
    const bool want_cs = ACT == 3 && !A_KC && g.cs && tn == 0;
    auto add_cs = [&](f32x4 (&x)[4]) {             // after store_chunk: the K tail is already zeroed in x
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[e] += x[ps][e];
    };
    auto compute = [&](int buf) {
        const float* sa = smem + buf * 2 * F_STAGE;
        const float* sb = sa + F_STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = read_frag<A_KC>(sa, wm * 64 + i * 32, q, r, h);
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = read_frag<B_KC>(sb, wn * 64 + j * 32, q, r, h);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bv[j][s], acc[i][j], 0, 0, 0);
        }
    };
    // one chunk: issue the loads of chunk k0 + 2 BK into (la, lb), compute chunk k0 from LDS[buf], then move chunk
    // k0 + BK from (sa_, sb_) to LDS[buf ^ 1]
    auto step = [&](f32x4 (&la)[4], f32x4 (&lb)[4], f32x4 (&sa_)[4], f32x4 (&sb_)[4], int k0, int buf) {
        if (k0 + 2 * F_BK < kend) {
            load_chunk<A_KC>(la, pa, g.lda, m0, g.M, k0 + 2 * F_BK, kend, tid);
            load_chunk<B_KC>(lb, pb, g.ldb, n0, g.N, k0 + 2 * F_BK, kend, tid);
        }
        compute(buf);
        if (k0 + F_BK < kend) {
            float* d = smem + (buf ^ 1) * 2 * F_STAGE;
            store_chunk<A_KC>(d, sa_, k0 + F_BK, kend, tid);
            store_chunk<B_KC>(d + F_STAGE, sb_, k0 + F_BK, kend, tid);
            if (want_cs) add_cs(sa_);
        }
        __syncthreads();
    };
    load_chunk<A_KC>(ra0, pa, g.lda, m0, g.M, kbeg, kend, tid);
    load_chunk<B_KC>(rb0, pb, g.ldb, n0, g.N, kbeg, kend, tid);
    if (kbeg + F_BK < kend) {
        load_chunk<A_KC>(ra1, pa, g.lda, m0, g.M, kbeg + F_BK, kend, tid);
        load_chunk<B_KC>(rb1, pb, g.ldb, n0, g.N, kbeg + F_BK, kend, tid);
    }
    store_chunk<A_KC>(smem, ra0, kbeg, kend, tid);
    store_chunk<B_KC>(smem + F_STAGE, rb0, kbeg, kend, tid);
    if (want_cs) add_cs(ra0);
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += 2 * F_BK) {
        step(ra0, rb0, ra1, rb1, k0, 0);
        if (k0 + F_BK < kend) step(ra1, rb1, ra0, rb0, k0 + F_BK, 1);
    }

    // ---- epilogue: lane (r, h), register e of a 32x32 accumulator = row krow(e, h), column r
    if (ACT == 3) {
        float* slab = g.c + (int64_t)split * g.M * g.ldc;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wm * 64 + i * 32 + acr_krow(e, h);
                    if (row < g.M && col < g.N) slab[(int64_t)row * g.ldc + col] = acc[i][j][e];
                }
            }
        if (!A_KC && g.cs && tn == 0) {
            // column sums of this split's A rows: thread (kr = tid/32, i4 = tid%32) holds 4 columns; sum the 8 kr rows via LDS
            float* red = smem;                                   // all fragment reads are behind the loop's last barrier
            *reinterpret_cast<f32x4*>(red + (tid >> 5) * 128 + 4 * (tid & 31)) = *reinterpret_cast<f32x4*>(csum);
            __syncthreads();
            if (tid < 128) {
                float s = 0.f;
#pragma unroll
                for (int kr = 0; kr < 8; ++kr) s += red[kr * 128 + tid];
                if (m0 + tid < g.M) g.cs[(int64_t)split * g.M + m0 + tid] = s;
            }
        }
        return;
    }
    GemmF32Args gz = g;                                     // this slice's output / addend
    gz.c += (int64_t)zs * g.c_zs;
    if (gz.aux) gz.aux += (int64_t)zs * g.aux_zs;
    if (m0 + F_BM <= g.M && n0 + F_BN <= g.N)
        epilogue_f32<ACT, false>(gz, acc, m0 + wm * 64, n0 + wn * 64, r, h);
    else
        epilogue_f32<ACT, true>(gz, acc, m0 + wm * 64, n0 + wn * 64, r, h);
}


// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (contraction length a multiple of 32): chunks go global -> LDS directly (global_load_lds_dwordx4),
// no staging registers and no ds_write path.  Measured on the register-staged kernel above (scripts/lab): its 8 loads +
// 8 ds_write_b128 per 64 MFMAs cost 18 % of the matrix pipe (4096^3: 150 TF with neither, 128 / 134 TF with one of them,
// 120 TF with both) although clocks, prefetch depth and wave priorities are not the cause -- VGPR traffic of loads and
// LDS stores competes with the MFMA operand reads.  A DMA wave-instruction writes 64 lanes x 16 B = 1 KiB linearly:
//   KC operand: 8 unpadded 128-byte rows; ds_read_b128 bank conflicts are removed by an XOR swizzle applied on the
//               SOURCE address (lane fetches 16-byte chunk p ^ ((row >> 1) & 7) into slot p) and mirrored on the read;
//   KS operand: 2 k-rows of 128 floats; fragment reads are 32 consecutive floats, no swizzle needed.
// ---------------------------------------------------------------------------------------------------------------
#define F_DTILE (F_BM * F_BK)        // floats per operand per stage, unpadded (16 KiB)

// Per-lane element offsets of this wave's 4 DMA pieces inside one operand, relative to (row 0 of the tile, contraction index
// 0 of the chunk): computed ONCE per workgroup.  Per chunk the source is then  uniform base (+ k advance, scalar) + this
// offset -- no vector arithmetic inside the loop (VALU instructions run on the lanes the fp32 MFMA uses; the per-chunk
// address math was ~40 of the loop's 58 VALU instructions).  Offsets are 32-bit: the host checks dim * ld < 2^31.
template <bool KC>
__device__ __forceinline__ void dma_offsets(int (&off)[4], int64_t ld, int i0, int dim, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;
        if (KC) {
            const int row = piece * 8 + (lane >> 3);
            const int lc = (lane & 7) ^ ((row >> 1) & 7);
            off[i] = min(i0 + row, dim - 1) * (int)ld + lc * 4;
        } else {
            const int kr = piece * 2 + (lane >> 5);
            off[i] = kr * (int)ld + min(i0 + 4 * (lane & 31), dim - 4);
        }
    }
}
// ub: uniform pointer to (row 0, contraction index k0) of the operand: p + k0 (KC) or p + k0 * ld (KS)
__device__ __forceinline__ void dma_chunk(float* s, const float* __restrict__ ub, const int (&off)[4], int wave) {
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((glb_vp)(ub + off[i]), (lds_vp)(s + (wave * 4 + i) * 256), 16, 0, 0);
}

template <bool KC>
__device__ __forceinline__ f32x4 dma_frag(const float* __restrict__ s, int base, int q, int r, int h) {
    if (KC) {
        const int row = base + r;
        return *reinterpret_cast<const f32x4*>(s + row * F_BK + (((2 * q + h) ^ ((row >> 1) & 7)) << 2));
    }
    const float* p = s + (8 * q + 4 * h) * F_BM + base + r;
    f32x4 v = {p[0], p[F_BM], p[2 * F_BM], p[3 * F_BM]};
    return v;
}


// everything after the K loop of a DMA-ring kernel: tail slab (ACT 4), split slab + bias-gradient column sums (ACT 3) or the
// epilogue.  `smem` must be free (all fragment reads behind a barrier).
template <bool A_KC, int ACT>
__device__ __forceinline__ void gemm_f32_finish(const GemmF32Args& g, f32x16 (&acc)[2][2], float* smem, int split, int tt, int tn, int m0,
                                                int n0, int zs, int wm, int wn, int r, int h, int tid, float csum, bool want_cs) {
    if (ACT == 4) {                                         // K-split tail tile: raw accumulators into a compact slab
        float* slab = g.c + ((int64_t)split * g.tiles_launch + (tt - g.tile0)) * (F_BM * F_BN);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = wn * 64 + j * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) slab[(wm * 64 + i * 32 + acr_krow(e, h)) * F_BN + col] = acc[i][j][e];
            }
        return;
    }
    if (ACT == 3) {
        float* slab = g.c + (int64_t)split * g.M * g.ldc;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wm * 64 + i * 32 + acr_krow(e, h);
                    if (row < g.M && col < g.N) slab[(int64_t)row * g.ldc + col] = acc[i][j][e];
                }
            }
        if (want_cs) {
            float* red = smem;                              // behind the loop's last barrier
            red[tid] = csum;
            __syncthreads();
            if (tid < 128 && m0 + tid < g.M) g.cs[(int64_t)split * g.M + m0 + tid] = red[tid] + red[tid + 128];
        }
        return;
    }
    GemmF32Args gz = g;
    gz.c += (int64_t)zs * g.c_zs;
    if (gz.aux) gz.aux += (int64_t)zs * g.aux_zs;
    if (m0 + F_BM <= g.M && n0 + F_BN <= g.N)
        epilogue_f32<ACT, false>(gz, acc, m0 + wm * 64, n0 + wn * 64, r, h);
    else
        epilogue_f32<ACT, true>(gz, acc, m0 + wm * 64, n0 + wn * 64, r, h);
}

template <bool A_KC, bool B_KC, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_f32_dma_kernel(const GemmF32Args g) {
    __shared__ __attribute__((aligned(1024))) float smem[4 * F_DTILE];      // [A0 | B0 | A1 | B1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int ntile = g.tiles_launch;
    const int t = acr_xcd_remap(blockIdx.x, ntile * g.nsplit);
    const int split = t / ntile, tt = g.tile0 + (t - split * ntile);
    int tm, tn;
    if (ACT == 3) { tm = tt / g.tiles_n; tn = tt - tm * g.tiles_n; }
    else tile_coords(tt, g.tiles_m, g.tiles_n, tm, tn);
    const int m0 = tm * F_BM, n0 = tn * F_BN;
    const int zs = split / g.ksplit;
    const int kbeg = (split - zs * g.ksplit) * g.k_zs, kend = min(g.K, kbeg + g.kps);      // host: (kend - kbeg) % F_BK == 0
    const float* __restrict__ pa = g.a + (int64_t)zs * g.a_zs;
    const float* __restrict__ pb = g.b + (int64_t)zs * g.b_zs;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float csum = 0.f;          // TN bias gradient: column tid & 127 of the A chunks, k rows of parity tid >> 7
    const bool want_cs = ACT == 3 && !A_KC && g.cs && tn == 0;

    int offa[4], offb[4];
    dma_offsets<A_KC>(offa, g.lda, m0, g.M, wave, lane);
    dma_offsets<B_KC>(offb, g.ldb, n0, g.N, wave, lane);
    const int64_t ka = A_KC ? 1 : g.lda, kb = B_KC ? 1 : g.ldb;      // operand advance per contraction index
    dma_chunk(smem, pa + kbeg * ka, offa, wave);
    dma_chunk(smem + F_DTILE, pb + kbeg * kb, offb, wave);
    acr_dma_barrier();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += F_BK, cur ^= 1) {
        if (k0 + F_BK < kend) {
            float* d = smem + (cur ^ 1) * 2 * F_DTILE;
            dma_chunk(d, pa + (k0 + F_BK) * ka, offa, wave);
            dma_chunk(d + F_DTILE, pb + (k0 + F_BK) * kb, offb, wave);
        }
        const float* sa = smem + cur * 2 * F_DTILE;
        const float* sb = sa + F_DTILE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = dma_frag<A_KC>(sa, wm * 64 + i * 32, q, r, h);
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = dma_frag<B_KC>(sb, wn * 64 + j * 32, q, r, h);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bv[j][s], acc[i][j], 0, 0, 0);
        }
        if (want_cs) {                                      // [k][i] image: 16 of the chunk's 32 k rows per thread
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) csum += sa[(2 * kk + (tid >> 7)) * F_BM + (tid & 127)];
        }
        acr_dma_barrier();                                  // the chunk in flight has landed; buffer `cur` is free
    }
    gemm_f32_finish<A_KC, ACT>(g, acc, smem, split, tt, tn, m0, n0, zs, wm, wn, r, h, tid, csum, want_cs);
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 products on the bf16 MFMA (math = ACR_MATH_BF16X3, a per-call argument): the same tiles, operands, epilogues and tail plan as the kernels
// above, but every fp32 product is evaluated on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA's rate) as SIX exact terms of a
// three-way operand split,
//     a = a0 + a1 + a2,  b = b0 + b1 + b2   (a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1): 3 x 8 = 24 mantissa bits)
//     a b ~ a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0        (the dropped terms are <= 2^-24 |a b|)
// every bf16 x bf16 product is exact in fp32 and the sums accumulate in fp32: against float64 the result is as accurate as the
// exact-fp32 MFMA chain (scripts/lab/gemm_split.py: rms error 8.6e-7 vs 9.9e-7 of rms y at K = 3072).  Operands stay fp32 in
// HBM and in LDS; fragments are split in registers right after their ds_read.
// Structure: a 32-deep chunk is now ~1.5 us of matrix work for two co-resident workgroups, less than the 3 us a first-touch
// DMA takes to land -- with the two-slot ring above the loop is latency-bound (3.2 us per chunk measured, x1.3 only).  So:
// 16-deep stages (one MFMA k-step), a FOUR-slot ring filled three stages ahead with counted vmcnt, and the split of stage t
// interleaved (sched_group_barrier) with the 24 MFMAs of stage t - 1, whose pieces wait in a second register set.
// ---------------------------------------------------------------------------------------------------------------
#define S_BK 16
#define S_TILE (F_BM * S_BK)         // floats per operand per stage (8 KiB)
#define S_SLOTS 4

__device__ __forceinline__ void split3_bf16(const f32x4& lo4, const f32x4& hi4, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
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
// per-lane element offsets of this wave's 2 DMA pieces of one 16-deep stage of one operand (computed once per workgroup).
// KC: piece = 16 rows x 64 bytes, lane -> (row = l >> 2, 16-byte chunk l & 3), chunk XOR-swizzled by (row >> 2) & 3 on the
// SOURCE address (mirrored by the fragment reads: every 16-lane group of a ds_read_b128 then hits 16 different bank quads);
// KS: piece = 2 k rows of 128 floats.
template <bool KC>
__device__ __forceinline__ void split_dma_offsets(int (&off)[2], int64_t ld, int i0, int dim, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i;
        if (KC) {
            const int row = piece * 16 + (lane >> 2);
            const int lc = (lane & 3) ^ ((row >> 2) & 3);
            off[i] = min(i0 + row, dim - 1) * (int)ld + lc * 4;
        } else {
            const int kr = piece * 2 + (lane >> 5);
            off[i] = kr * (int)ld + min(i0 + 4 * (lane & 31), dim - 4);
        }
    }
}
__device__ __forceinline__ void split_dma_stage(float* s, const float* __restrict__ ub, const int (&off)[2], int wave) {
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((glb_vp)(ub + off[i]), (lds_vp)(s + (wave * 2 + i) * 256), 16, 0, 0);
}
// the 8 consecutive contraction indices 8 h + (0..7) of lane (r, h) for 32 rows starting at `base`
template <bool KC>
__device__ __forceinline__ void split_frag8(const float* __restrict__ s, int base, int r, int h, f32x4& lo4, f32x4& hi4) {
    if (KC) {
        const int row = base + r, sw = (row >> 2) & 3;
        lo4 = *reinterpret_cast<const f32x4*>(s + row * S_BK + (((2 * h) ^ sw) << 2));
        hi4 = *reinterpret_cast<const f32x4*>(s + row * S_BK + (((2 * h + 1) ^ sw) << 2));
    } else {
        const float* p = s + (8 * h) * F_BM + base + r;
        lo4 = f32x4{p[0], p[F_BM], p[2 * F_BM], p[3 * F_BM]};
        hi4 = f32x4{p[4 * F_BM], p[5 * F_BM], p[6 * F_BM], p[7 * F_BM]};
    }
}

// The same fragment through inline-asm LDS reads.  hipcc cannot tell an LDS-DMA's LDS write from a read of another ring slot:
// in front of the first compiler-visible LDS load behind a DMA it waits for ALL outstanding vector-memory operations
// (`s_waitcnt vmcnt(0)`, found in the ISA right before the stage barrier) -- which turned the three-stages-ahead ring into a
// one-stage-ahead one: every stage waited for the DMAs issued one stage earlier.  asm reads are invisible to that analysis; the
// counted vmcnt wait + barrier in front of them and the lgkmcnt wait behind them (SPLIT_LDS_WAIT8) are the synchronisation.
// Lane bases (LDS byte addresses inside slot 0's A resp. B tile): KC two per fragment (the two swizzled 16-byte chunks), KS one.
template <bool KC>
__device__ __forceinline__ void split_frag_bases(uint32_t (&b)[2], const float* tile0, int base, int r, int h) {
    typedef const __attribute__((address_space(3))) char* lds_cp;
    const uint32_t t = (uint32_t)(uintptr_t)(lds_cp)tile0;
    if (KC) {
        const int row = base + r, sw = (row >> 2) & 3;
        b[0] = t + row * (S_BK * 4) + (((2 * h) ^ sw) << 4);
        b[1] = t + row * (S_BK * 4) + (((2 * h + 1) ^ sw) << 4);
    } else {
        b[0] = b[1] = t + ((8 * h) * F_BM + base + r) * 4;
    }
}
#define SPLIT_RD128(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=&v"(dst) : "v"(addr))
#define SPLIT_RD32(dst, addr, OFF) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "i"(OFF))
template <bool KC>
__device__ __forceinline__ void split_frag8_x(const uint32_t (&b)[2], uint32_t slot_off, f32x4& lo4, f32x4& hi4) {
    if (KC) {
        SPLIT_RD128(lo4, b[0] + slot_off);
        SPLIT_RD128(hi4, b[1] + slot_off);
    } else {
        const uint32_t a = b[0] + slot_off;
        SPLIT_RD32(lo4[0], a, 0); SPLIT_RD32(lo4[1], a, F_BM * 4); SPLIT_RD32(lo4[2], a, 2 * F_BM * 4); SPLIT_RD32(lo4[3], a, 3 * F_BM * 4);
        SPLIT_RD32(hi4[0], a, 4 * F_BM * 4); SPLIT_RD32(hi4[1], a, 5 * F_BM * 4); SPLIT_RD32(hi4[2], a, 6 * F_BM * 4); SPLIT_RD32(hi4[3], a, 7 * F_BM * 4);
    }
}
#define SPLIT_LDS_WAIT8(a0, a1, a2, a3, a4, a5, a6, a7) \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))

template <bool A_KC, bool B_KC, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_f32_split_kernel(const GemmF32Args g) {
    __shared__ __attribute__((aligned(1024))) float smem[S_SLOTS * 2 * S_TILE];      // [slot][A | B], 64 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int ntile = g.tiles_launch;
    const int t0 = acr_xcd_remap(blockIdx.x, ntile * g.nsplit);
    const int split = t0 / ntile, tt = g.tile0 + (t0 - split * ntile);
    int tm, tn;
    if (ACT == 3) { tm = tt / g.tiles_n; tn = tt - tm * g.tiles_n; }
    else tile_coords(tt, g.tiles_m, g.tiles_n, tm, tn);
    const int m0 = tm * F_BM, n0 = tn * F_BN;
    const int zs = split / g.ksplit;
    const int kbeg = (split - zs * g.ksplit) * g.k_zs, kend = min(g.K, kbeg + g.kps);      // host: (kend - kbeg) % 16 == 0
    const float* __restrict__ pa = g.a + (int64_t)zs * g.a_zs;
    const float* __restrict__ pb = g.b + (int64_t)zs * g.b_zs;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float csum = 0.f;          // TN bias gradient: column tid & 127 of the A stages, k rows of parity tid >> 7
    const bool want_cs = ACT == 3 && !A_KC && g.cs && tn == 0;
    int offa[2], offb[2];
    split_dma_offsets<A_KC>(offa, g.lda, m0, g.M, wave, lane);
    split_dma_offsets<B_KC>(offb, g.ldb, n0, g.N, wave, lane);
    const int64_t ka = A_KC ? 1 : g.lda, kb = B_KC ? 1 : g.ldb;      // operand advance per contraction index
    const int nst = (kend - kbeg) / S_BK;
    uint32_t fa[2][2], fb[2][2];                             // fragment lane bases in slot 0
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        split_frag_bases<A_KC>(fa[i], smem, wm * 64 + i * 32, r, h);
        split_frag_bases<B_KC>(fb[i], smem + S_TILE, wn * 64 + i * 32, r, h);
    }
    const uint32_t cs_base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem + ((tid >> 7) * F_BM + (tid & 127)) * 4;
    auto issue = [&](int st) {
        float* d = smem + (st & (S_SLOTS - 1)) * 2 * S_TILE;
        split_dma_stage(d, pa + (int64_t)(kbeg + st * S_BK) * ka, offa, wave);
        split_dma_stage(d + S_TILE, pb + (int64_t)(kbeg + st * S_BK) * kb, offb, wave);
    };
#pragma unroll
    for (int st = 0; st < S_SLOTS - 1; ++st)
        if (st < nst) issue(st);
    bf16x8 ap[2][2][3], bp[2][2][3];                        // [register set][block][piece]
    f32x4 ra[2][2], rb[2][2];
#define ACR_SPLIT_MFMA6(SET, I, J)                                                                                        \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][2], acc[I][J], 0, 0, 0);               \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][2], bp[SET][J][0], acc[I][J], 0, 0, 0);               \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][1], bp[SET][J][1], acc[I][J], 0, 0, 0);               \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][1], acc[I][J], 0, 0, 0);               \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][1], bp[SET][J][0], acc[I][J], 0, 0, 0);               \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][0], acc[I][J], 0, 0, 0);
    // stage st: wait until it has landed (stages st+1, st+2 may stay in flight: 4 DMA instructions each), publish it, refill the
    // slot stage st-1 was read from, read + split stage st into register set SET while the MFMAs of stage st-1 (set SET^1) run
    auto step = [&](int st, auto set_tag, auto first_tag) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        if (st + 2 < nst) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acr_barrier_nofence();                              // stage st landed for every wave; all reads of stage st - 1 were waited for (SPLIT_LDS_WAIT8)
        if (st + S_SLOTS - 1 < nst) issue(st + S_SLOTS - 1);
        const uint32_t so = (uint32_t)(st & (S_SLOTS - 1)) * (2 * S_TILE * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) split_frag8_x<A_KC>(fa[i], so, ra[i][0], ra[i][1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) split_frag8_x<B_KC>(fb[j], so, rb[j][0], rb[j][1]);
        if (want_cs) {                                      // [k][i] image: 8 of the stage's 16 k rows per thread
            float c8[8];
            const uint32_t ca = cs_base + so;
            SPLIT_RD32(c8[0], ca, 0); SPLIT_RD32(c8[1], ca, 2 * F_BM * 4); SPLIT_RD32(c8[2], ca, 4 * F_BM * 4); SPLIT_RD32(c8[3], ca, 6 * F_BM * 4);
            SPLIT_RD32(c8[4], ca, 8 * F_BM * 4); SPLIT_RD32(c8[5], ca, 10 * F_BM * 4); SPLIT_RD32(c8[6], ca, 12 * F_BM * 4); SPLIT_RD32(c8[7], ca, 14 * F_BM * 4);
            SPLIT_LDS_WAIT8(c8[0], c8[1], c8[2], c8[3], c8[4], c8[5], c8[6], c8[7]);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) csum += c8[kk];
        }
        SPLIT_LDS_WAIT8(ra[0][0], ra[0][1], ra[1][0], ra[1][1], rb[0][0], rb[0][1], rb[1][0], rb[1][1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) split3_bf16(ra[i][0], ra[i][1], ap[SET][i][0], ap[SET][i][1], ap[SET][i][2]);
#pragma unroll
        for (int j = 0; j < 2; ++j) split3_bf16(rb[j][0], rb[j][1], bp[SET][j][0], bp[SET][j][1], bp[SET][j][2]);
        if (!FIRST) {
            ACR_SPLIT_MFMA6(SET ^ 1, 0, 0) ACR_SPLIT_MFMA6(SET ^ 1, 0, 1) ACR_SPLIT_MFMA6(SET ^ 1, 1, 0) ACR_SPLIT_MFMA6(SET ^ 1, 1, 1)
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
    if (nst & 1) { ACR_SPLIT_MFMA6(0, 0, 0) ACR_SPLIT_MFMA6(0, 0, 1) ACR_SPLIT_MFMA6(0, 1, 0) ACR_SPLIT_MFMA6(0, 1, 1) }
    else { ACR_SPLIT_MFMA6(1, 0, 0) ACR_SPLIT_MFMA6(1, 0, 1) ACR_SPLIT_MFMA6(1, 1, 0) ACR_SPLIT_MFMA6(1, 1, 1) }
#undef ACR_SPLIT_MFMA6
    __syncthreads();                                        // every wave is done with the ring: the finish may reuse it
    gemm_f32_finish<A_KC, ACT>(g, acc, smem, split, tt, tn, m0, n0, zs, wm, wn, r, h, tid, csum, want_cs);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Split products on PRE-SPLIT, PRE-TILED operands (round 4).  gemm_f32_split_kernel splits its operand tiles in registers:
// ~200 VALU instructions per 24 MFMAs per wave, and every operand element is split again by every workgroup that reads it
// (N / 128 resp. M / 128 times); its counters (profiles/r04_pmc_split_gemm.txt) show the VALU port -- which also issues the
// MFMAs -- busy 66 % of the time and the matrix pipe 57 %.  Here every operand is split ONCE per product by a streaming pass
// (planes_tile_kernel / planes_tile_t_kernel: HBM-bound, 10 bytes per element) into three bf16 planes in the workspace,
// transposed on the way where the product needs it, so that ONE kernel flavour (both operands [row][k]) serves NT, NN and TN
// and its loop is DMA + ds_read_b128 + MFMA only.
// The planes are stored TILED, in exactly the image the kernel wants in LDS: for row block rb (128 rows) and stage kb (16
// contraction elements) the three 4 KiB planes [128 rows][32 bytes] follow each other,
//     byte offset = ((rb * nkb + kb) * 3 + p) * 4096 + row * 32 + 16 * (khalf ^ bit 3 of row) + 2 * (k & 7),
// so a stage of an operand is 12 KiB of CONTIGUOUS memory and every LDS-DMA instruction copies one contiguous KiB.  The first
// version kept dense row-major planes: 32 bytes per row and stage made every DMA instruction touch 32 cache lines, the
// texture-address units were busy 95 % of the kernel and the matrix pipe 37 % (profiles/r04_pmc_planes_gemm_first_version.txt).
// The half swap (bit 3 of the row) makes the 16 lanes a ds_read_b128 serves per cycle hit 16 different 16-byte bank groups.
// Rows past the operand's end and contraction indices past K are zero in the image (no clamps, no K % 16 condition).
// Ring of 3 slots x [A p0 p1 p2 | B p0 p1 p2], DMA two stages ahead (6 pieces per wave and stage; waves 0-1 fetch A, 2-3 B).
// The reads of stage st and the refill of the ring are interleaved with the 24 MFMAs of stage st - 1 in program order
// (sched_barrier between the groups: inline-asm reads are invisible to sched_group_barrier).
// ---------------------------------------------------------------------------------------------------------------------------------
#define P_BK 16
#define P_SLOTS 3
#define P_TILE_B 4096                 // bytes per plane per operand per stage (128 rows x 32 B)
#define P_STAGE_B (6 * P_TILE_B)
#define PL_RD(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "i"(OFF))

// ---- image epilogues: the product's output leaves the kernel AS the image the next product reads -----------------------------
// ACT 5 (fc1 forward): h = acc + bias; c = GELU'(h) in fp32 (all the backward needs of h), c2 = IMAGE of GELU(h) -- fc2's operand
//        in the forward and in its weight gradient; the fp32 activation is never written.
// ACT 6 (fc2's input gradient): c2 = IMAGE of acc * aux (aux = the saved GELU'(h)) -- fc1's dy for its input and weight gradient;
//        cs = per-tile-row parts of its column sums (fc1's bias gradient, summed in tile-row order by planes_colsum_kernel);
//        no fp32 output at all.
// Saves the 25 120 x 3072 image passes of both tensors (0.135 ms each, 24 per step) and their fp32 writes.  Rows past M and
// columns past N are written as zeros (the image contract).
#define X3E_PITCH 132                 // floats: 528 bytes, 16-byte aligned rows
__device__ __forceinline__ void planes_split8(const float (&x)[8], bf16x8& p0, bf16x8& p1, bf16x8& p2);
__device__ __forceinline__ int planes_chunk_off(int rr, int kh);
// one 16-byte chunk (row row_t of row block tm, columns 8 c8 .. + 7 of output tile column tn) of all three planes
__device__ __forceinline__ void x3_image_chunk_store(char* img, int tm, int tn, int nkb, int row_t, int c8, const float (&v)[8]) {
    const int kb = tn * 8 + (c8 >> 1);
    if (kb >= nkb) return;
    bf16x8 p0, p1, p2;
    planes_split8(v, p0, p1, p2);
    char* dst = img + ((int64_t)tm * nkb + kb) * (3 * 4096) + planes_chunk_off(row_t, c8 & 1);
    *reinterpret_cast<bf16x8*>(dst) = p0;
    *reinterpret_cast<bf16x8*>(dst + 4096) = p1;
    *reinterpret_cast<bf16x8*>(dst + 2 * 4096) = p2;
}
// Thread -> chunks of an output tile: pass (half, j) handles row 32 j + (tid >> 3), columns 64 half + 8 (tid & 7) .. + 7 -- the
// mapping of planes_tile_kernel, so that the column sums below add the same numbers in the same order as the image pass would
// (a bias gradient does not depend on which of the two produced the image, bit for bit).
// column sums of a tile from the per-thread sums csum[half][e], through `red` (>= 4096 floats of LDS nobody else is using)
__device__ __forceinline__ void x3_tile_colsum(const float (&csum)[2][8], float* red, int tid, float* parts_row, int n0, int N) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 3) * 128 + hf * 64 + (tid & 7) * 8 + e] = csum[hf][e];
    __syncthreads();
    if (tid < 128 && n0 + tid < N) {
        float t = red[tid];
        for (int q = 1; q < 32; ++q) t += red[q * 128 + tid];
        parts_row[n0 + tid] = t;
    }
}
template <int ACT>
__device__ __forceinline__ void x3_finish_image(const GemmF32Args& g, f32x16 (&acc)[2][2], float* tl, int tm, int tn, int m0, int n0, int wm,
                                                int wn, int r, int h, int tid) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col_t = wn * 64 + j * 32 + r, col = n0 + col_t;
        const bool cok = col < g.N;
        const float bj = (ACT == 5 && g.bias && cok) ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float x[16];
            if (ACT == 6) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wm * 64 + i * 32 + acr_krow(e, h);
                    x[e] = (row < g.M && cok) ? g.aux[(int64_t)row * g.ldaux + col] : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row_t = wm * 64 + i * 32 + acr_krow(e, h), row = m0 + row_t;
                const bool ok = row < g.M && cok;
                float val;
                if (ACT == 5) {
                    const float v = acc[i][j][e] + bj;
                    const float er = erff(v * 0.70710678118654752440f);
                    val = v * 0.5f * (1.0f + er);
                    if (ok) g.c[(int64_t)row * g.ldc + col] = 0.5f * (1.0f + er) + v * (expf(-0.5f * v * v) * 0.39894228040143267794f);
                } else {
                    val = acc[i][j][e] * x[e];
                }
                tl[row_t * X3E_PITCH + col_t] = ok ? val : 0.f;
            }
        }
    }
    __syncthreads();
    char* img = reinterpret_cast<char*>(g.c2);
    float csum[2][8];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[hf][e] = 0.f;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row_t = j * 32 + (tid >> 3), c8 = hf * 8 + (tid & 7);
            const f32x4 a = *reinterpret_cast<const f32x4*>(tl + row_t * X3E_PITCH + c8 * 8);
            const f32x4 b = *reinterpret_cast<const f32x4*>(tl + row_t * X3E_PITCH + c8 * 8 + 4);
            const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            if (ACT == 6) {
#pragma unroll
                for (int e = 0; e < 8; ++e) csum[hf][e] += v[e];
            }
            x3_image_chunk_store(img, tm, tn, g.img_nkb, row_t, c8, v);
        }
    if (ACT == 6 && g.cs) {
        __syncthreads();                                    // every thread is done reading the tile: reuse it for the reduction
        x3_tile_colsum(csum, tl, tid, g.cs + (int64_t)tm * g.N, n0, g.N);
    }
}
// The same for K-split tail tiles (gemm_tail_plan): one workgroup per tail tile sums its `nsplit` slabs in part order -- a slab is
// row-major, so a thread's 8 consecutive columns are two float4 per part -- applies the epilogue and writes the image chunks.
template <int ACT>
__global__ __launch_bounds__(256) void gemm_x3_tail_image_kernel(const GemmF32Args g, const float* __restrict__ ws, int ntail, int nsplit) {
    __shared__ float red[4096];
    const int tid = threadIdx.x, tix = blockIdx.x;
    const int tt = g.tile0 + tix;
    int tm, tn;
    tile_coords(tt, g.tiles_m, g.tiles_n, tm, tn);
    const int m0 = tm * F_BM, n0 = tn * F_BN;
    char* img = reinterpret_cast<char*>(g.c2);
    float csum[2][8];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[hf][e] = 0.f;
#pragma unroll
    for (int hj = 0; hj < 8; ++hj) {
        const int hf = hj >> 2, j = hj & 3;
        const int row_t = j * 32 + (tid >> 3), c8 = hf * 8 + (tid & 7);
        const int row = m0 + row_t, col = n0 + c8 * 8;
        const float* p = ws + (int64_t)tix * (F_BM * F_BN) + row_t * F_BN + c8 * 8;
        f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
        for (int k = 1; k < nsplit; ++k) {
            const float* q = p + (int64_t)k * ntail * (F_BM * F_BN);
            const f32x4 u = *reinterpret_cast<const f32x4*>(q), w = *reinterpret_cast<const f32x4*>(q + 4);
            a[0] += u[0]; a[1] += u[1]; a[2] += u[2]; a[3] += u[3]; b[0] += w[0]; b[1] += w[1]; b[2] += w[2]; b[3] += w[3];
        }
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        const bool rok = row < g.M;
        if (rok && col < g.N) {                             // host: N % 8 == 0 for the image epilogues
            if (ACT == 5) {
                f32x4 d0, d1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float hv = v[e] + (g.bias ? g.bias[col + e] : 0.f);
                    const float er = erff(hv * 0.70710678118654752440f);
                    v[e] = hv * 0.5f * (1.0f + er);
                    const float d = 0.5f * (1.0f + er) + hv * (expf(-0.5f * hv * hv) * 0.39894228040143267794f);
                    if (e < 4) d0[e] = d; else d1[e - 4] = d;
                }
                float* cp = g.c + (int64_t)row * g.ldc + col;
                *reinterpret_cast<f32x4*>(cp) = d0; *reinterpret_cast<f32x4*>(cp + 4) = d1;
            } else {
                const float* xp = g.aux + (int64_t)row * g.ldaux + col;
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(xp), x1 = *reinterpret_cast<const f32x4*>(xp + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= e < 4 ? x0[e] : x1[e - 4];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        if (ACT == 6) {
#pragma unroll
            for (int e = 0; e < 8; ++e) csum[hf][e] += v[e];
        }
        x3_image_chunk_store(img, tm, tn, g.img_nkb, row_t, c8, v);
    }
    if (ACT == 6 && g.cs) x3_tile_colsum(csum, red, tid, g.cs + (int64_t)tm * g.N, n0, g.N);
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void gemm_f32_planes_kernel(const GemmF32Args g) {
    __shared__ __attribute__((aligned(1024))) float smem[P_SLOTS * P_STAGE_B / 4];      // 72 KiB
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int ntile = g.tiles_launch;
    const int t0 = acr_xcd_remap(blockIdx.x, ntile * g.nsplit);
    const int split = t0 / ntile, tt = g.tile0 + (t0 - split * ntile);
    int tm, tn;
    if (ACT == 3) { tm = tt / g.tiles_n; tn = tt - tm * g.tiles_n; }
    else tile_coords(tt, g.tiles_m, g.tiles_n, tm, tn);
    const int m0 = tm * F_BM, n0 = tn * F_BN;
    const int zs = split / g.ksplit;
    const int kbeg = (split - zs * g.ksplit) * g.k_zs, kend = min(g.K, kbeg + g.kps);      // host: K, kps multiples of 16
    const int nkb = g.K / P_BK;                             // stages per row block in the tiled image
    // this wave's six KiB of every stage: waves 0, 1 the two halves of A's 12 KiB, waves 2, 3 of B's
    const char* __restrict__ pw = (wave < 2 ? reinterpret_cast<const char*>(g.a) + ((int64_t)tm * nkb + kbeg / P_BK) * (3 * P_TILE_B)
                                            : reinterpret_cast<const char*>(g.b) + ((int64_t)tn * nkb + kbeg / P_BK) * (3 * P_TILE_B)) +
                                  (wave & 1) * (6 * 1024) + lane * 16;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nst = (kend - kbeg) / P_BK;
    char* sm = reinterpret_cast<char*>(smem);
    auto dma1 = [&](int st, int slot, int i) {
        __builtin_amdgcn_global_load_lds((glb_vp)(pw + (int64_t)st * (3 * P_TILE_B) + i * 1024), (lds_vp)(sm + slot * P_STAGE_B + (wave * 6 + i) * 1024), 16, 0, 0);
    };
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)sm;
    const uint32_t hx = (h ^ ((r >> 3) & 1)) * 16;
    const uint32_t fa = lbase + (wm * 64 + r) * 32 + hx, fb = lbase + 3 * P_TILE_B + (wn * 64 + r) * 32 + hx;
#pragma unroll
    for (int i = 0; i < 6; ++i) dma1(0, 0, i);
#pragma unroll
    for (int i = 0; i < 6; ++i) dma1(min(1, nst - 1), 1, i);
    bf16x8 ap[2][2][3], bp[2][2][3];                        // [register set][block][plane]
#define PL_MFMA(SET, I, J, PA, PB) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][PA], bp[SET][J][PB], acc[I][J], 0, 0, 0);
#define PL_PAIR(SET, I, J, T)                                                       \
    if (T == 0) { PL_MFMA(SET, I, J, 0, 2) PL_MFMA(SET, I, J, 2, 0) }              \
    else if (T == 1) { PL_MFMA(SET, I, J, 1, 1) PL_MFMA(SET, I, J, 0, 1) }         \
    else { PL_MFMA(SET, I, J, 1, 0) PL_MFMA(SET, I, J, 0, 0) }
    // step st (slot = st % 3): stage st has landed for everyone -> refill the slot stage st - 1 was read from with stage st + 2
    // (past the end: the last stage again, into a slot nobody reads -- keeps the DMA count per step, hence the vmcnt, constant),
    // read stage st into register set SET while the MFMAs of stage st - 1 (set SET ^ 1) run
    auto step = [&](int st, int slot, auto set_tag, auto first_tag) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // younger: the 6 pieces of stage st + 1
        acr_barrier_nofence();
        const int rslot = slot == 0 ? 2 : slot - 1;         // (st + 2) % 3
        const int rst = min(st + 2, nst - 1);
        const uint32_t fas = fa + slot * P_STAGE_B, fbs = fb + slot * P_STAGE_B;
#define PL_GROUP(K12)                                                                                                   \
        if (!FIRST) { PL_PAIR(SET ^ 1, ((K12) / 6), (((K12) / 3) & 1), ((K12) % 3)) }                                   \
        if ((K12) < 6) PL_RD(ap[SET][(K12) / 3][(K12) % 3], fas, ((K12) % 3) * P_TILE_B + ((K12) / 3) * 1024);          \
        else PL_RD(bp[SET][((K12) - 6) / 3][(K12) % 3], fbs, ((K12) % 3) * P_TILE_B + (((K12) - 6) / 3) * 1024);        \
        if ((K12) & 1) dma1(rst, rslot, (K12) >> 1);                                                                   \
        __builtin_amdgcn_sched_barrier(0);
        PL_GROUP(0) PL_GROUP(1) PL_GROUP(2) PL_GROUP(3) PL_GROUP(4) PL_GROUP(5)
        PL_GROUP(6) PL_GROUP(7) PL_GROUP(8) PL_GROUP(9) PL_GROUP(10) PL_GROUP(11)
#undef PL_GROUP
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ap[SET][0][0]), "+v"(ap[SET][0][1]), "+v"(ap[SET][0][2]), "+v"(ap[SET][1][0]), "+v"(ap[SET][1][1]), "+v"(ap[SET][1][2]),
                       "+v"(bp[SET][0][0]), "+v"(bp[SET][0][1]), "+v"(bp[SET][0][2]), "+v"(bp[SET][1][0]), "+v"(bp[SET][1][1]), "+v"(bp[SET][1][2]));
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
#define PL_ALL(SET)                                                                                      \
    PL_PAIR(SET, 0, 0, 0) PL_PAIR(SET, 0, 0, 1) PL_PAIR(SET, 0, 0, 2) PL_PAIR(SET, 0, 1, 0) PL_PAIR(SET, 0, 1, 1) PL_PAIR(SET, 0, 1, 2) \
    PL_PAIR(SET, 1, 0, 0) PL_PAIR(SET, 1, 0, 1) PL_PAIR(SET, 1, 0, 2) PL_PAIR(SET, 1, 1, 0) PL_PAIR(SET, 1, 1, 1) PL_PAIR(SET, 1, 1, 2)
    if (nst & 1) { PL_ALL(0) } else { PL_ALL(1) }
#undef PL_ALL
#undef PL_PAIR
#undef PL_MFMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the refills past the end
    __syncthreads();                                        // every wave is done with the ring: the finish may reuse it
    if (ACT == 5 || ACT == 6) x3_finish_image<ACT>(g, acc, smem, tm, tn, m0, n0, wm, wn, r, h, tid);
    else gemm_f32_finish<true, ACT == 5 || ACT == 6 ? 0 : ACT>(g, acc, smem, split, tt, tn, m0, n0, zs, wm, wn, r, h, tid, 0.f, false);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient on the SAME images the forward / input-gradient products read (no transposed copies of dy or x):
//   c[m][n] = sum_t a[t][m] b[t][n],   a = dy image, b = x image, both tiled [token block of 128][stage of 16 features].
// The contraction index is now the image's ROW: a 16-token stage of a 128-feature operand tile is, for each of its 8 feature
// stages and 3 planes, 16 consecutive 32-byte rows = 512 contiguous bytes (one LDS-DMA instruction copies two of them), and a
// fragment -- 8 tokens of one feature per lane -- is read TRANSPOSED from those [16 tokens][16 features] chunks by
// ds_read_b64_tr_b16 (a 16-lane group takes 4 token rows x 16 features = 128 contiguous bytes, conflict-free; lane (r, h)
// receives tokens 4 h + (0..3) and 8 + 4 h + (0..3) of feature r: the same permutation of the contraction index for both
// operands).  Slot = [a: plane][feature stage][512 B] | [b: ...]; ring, waits, interleaving as gemm_f32_planes_kernel.
// ---------------------------------------------------------------------------------------------------------------------------------
#define PL_RDTR(lo, hi, alo, ahi, OFF)                                                                  \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"          \
                 : "=&v"(lo), "=&v"(hi) : "v"(alo), "v"(ahi), "i"(OFF))
__global__ __launch_bounds__(256, 2) void gemm_f32_planes_tn_kernel(const GemmF32Args g) {
    __shared__ __attribute__((aligned(1024))) float smem[P_SLOTS * P_STAGE_B / 4];      // 72 KiB
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int ntile = g.tiles_launch;
    const int t0 = acr_xcd_remap(blockIdx.x, ntile * g.nsplit);
    const int split = t0 / ntile, tt = g.tile0 + (t0 - split * ntile);
    const int tm = tt / g.tiles_n, tn = tt - tm * g.tiles_n;
    const int m0 = tm * F_BM, n0 = tn * F_BN;
    const int kbeg = split * g.k_zs, kend = min(g.K, kbeg + g.kps);      // tokens; host: K, kps multiples of 16
    const int nkb = wave < 2 ? g.nkb_a : g.nkb_b;          // feature stages per token block of this wave's operand
    const int f0 = (wave < 2 ? tm : tn) * 8;               // first feature stage of the tile
    const char* __restrict__ pw = reinterpret_cast<const char*>(wave < 2 ? g.a : g.b);
    // piece q = 6 (wave & 1) + i of the operand's 12: plane q >> 2, chunk pair q & 3 (feature stages f0 + 2 (q & 3) + (lane >> 5));
    // feature stages past the operand's end (M or N not a multiple of 128) alias the last one: rows the finish never stores
    int offd[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int q = (wave & 1) * 6 + i, pl = q >> 2, fs = min(f0 + 2 * (q & 3) + (lane >> 5), nkb - 1);
        offd[i] = (fs * 3 + pl) * P_TILE_B + (lane & 31) * 16;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nst = (kend - kbeg) / P_BK;
    char* sm = reinterpret_cast<char*>(smem);
    auto dma1 = [&](int st, int slot, int i) {
        const int tk = kbeg + st * P_BK;                    // uniform
        const char* src = pw + ((int64_t)(tk >> 7) * nkb) * (3 * P_TILE_B) + (tk & 127) * 32;
        const int q = (wave & 1) * 6 + i;
        __builtin_amdgcn_global_load_lds((glb_vp)(src + offd[i]), (lds_vp)(sm + slot * P_STAGE_B + (wave >> 1) * (3 * P_TILE_B) + (q >> 2) * P_TILE_B + (q & 3) * 1024),
                                         16, 0, 0);
    };
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)sm;
    const int i16 = lane & 15, g1 = (lane >> 4) & 1, qq = i16 >> 2, pp = i16 & 3;
    // lo: token row 4 h + qq (< 8: halves as stored), hi: row 8 + 4 h + qq (halves swapped)
    const uint32_t tlo = (4 * h + qq) * 32 + ((pp >> 1) << 4) + 8 * (pp & 1), thi = (8 + 4 * h + qq) * 32 + (((pp >> 1) ^ 1) << 4) + 8 * (pp & 1);
    const uint32_t fa_lo = lbase + (wm * 4 + g1) * 512 + tlo, fa_hi = lbase + (wm * 4 + g1) * 512 + thi;
    const uint32_t fb_lo = lbase + 3 * P_TILE_B + (wn * 4 + g1) * 512 + tlo, fb_hi = lbase + 3 * P_TILE_B + (wn * 4 + g1) * 512 + thi;
#pragma unroll
    for (int i = 0; i < 6; ++i) dma1(0, 0, i);
#pragma unroll
    for (int i = 0; i < 6; ++i) dma1(min(1, nst - 1), 1, i);
    bf16x4 al[2][2][3], ah[2][2][3], bl[2][2][3], bh[2][2][3];      // [register set][block][plane], tokens lo / hi
#define PT_MFMA(SET, I, J, PA, PB)                                                                                                        \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(al[SET][I][PA], ah[SET][I][PA], 0, 1, 2, 3, 4, 5, 6, 7), \
                                                        __builtin_shufflevector(bl[SET][J][PB], bh[SET][J][PB], 0, 1, 2, 3, 4, 5, 6, 7), acc[I][J], 0, 0, 0);
#define PT_PAIR(SET, I, J, T)                                                       \
    if (T == 0) { PT_MFMA(SET, I, J, 0, 2) PT_MFMA(SET, I, J, 2, 0) }              \
    else if (T == 1) { PT_MFMA(SET, I, J, 1, 1) PT_MFMA(SET, I, J, 0, 1) }         \
    else { PT_MFMA(SET, I, J, 1, 0) PT_MFMA(SET, I, J, 0, 0) }
    auto step = [&](int st, int slot, auto set_tag, auto first_tag) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // younger: the 6 pieces of stage st + 1
        acr_barrier_nofence();
        const int rslot = slot == 0 ? 2 : slot - 1;         // (st + 2) % 3
        const int rst = min(st + 2, nst - 1);
        const uint32_t so = slot * P_STAGE_B;
        const uint32_t a_lo = fa_lo + so, a_hi = fa_hi + so, b_lo = fb_lo + so, b_hi = fb_hi + so;
#define PT_GROUP(K12)                                                                                                                   \
        if (!FIRST) { PT_PAIR(SET ^ 1, ((K12) / 6), (((K12) / 3) & 1), ((K12) % 3)) }                                                   \
        if ((K12) < 6) PL_RDTR(al[SET][(K12) / 3][(K12) % 3], ah[SET][(K12) / 3][(K12) % 3], a_lo, a_hi, ((K12) % 3) * P_TILE_B + ((K12) / 3) * 1024); \
        else PL_RDTR(bl[SET][((K12) - 6) / 3][(K12) % 3], bh[SET][((K12) - 6) / 3][(K12) % 3], b_lo, b_hi, ((K12) % 3) * P_TILE_B + (((K12) - 6) / 3) * 1024); \
        if ((K12) & 1) dma1(rst, rslot, (K12) >> 1);                                                                                   \
        __builtin_amdgcn_sched_barrier(0);
        PT_GROUP(0) PT_GROUP(1) PT_GROUP(2) PT_GROUP(3) PT_GROUP(4) PT_GROUP(5)
        PT_GROUP(6) PT_GROUP(7) PT_GROUP(8) PT_GROUP(9) PT_GROUP(10) PT_GROUP(11)
#undef PT_GROUP
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(al[SET][0][0]), "+v"(al[SET][0][1]), "+v"(al[SET][0][2]), "+v"(al[SET][1][0]), "+v"(al[SET][1][1]), "+v"(al[SET][1][2]),
                       "+v"(ah[SET][0][0]), "+v"(ah[SET][0][1]), "+v"(ah[SET][0][2]), "+v"(ah[SET][1][0]), "+v"(ah[SET][1][1]), "+v"(ah[SET][1][2]),
                       "+v"(bl[SET][0][0]), "+v"(bl[SET][0][1]), "+v"(bl[SET][0][2]), "+v"(bl[SET][1][0]), "+v"(bl[SET][1][1]), "+v"(bl[SET][1][2]),
                       "+v"(bh[SET][0][0]), "+v"(bh[SET][0][1]), "+v"(bh[SET][0][2]), "+v"(bh[SET][1][0]), "+v"(bh[SET][1][1]), "+v"(bh[SET][1][2]));
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
#define PT_ALL(SET)                                                                                      \
    PT_PAIR(SET, 0, 0, 0) PT_PAIR(SET, 0, 0, 1) PT_PAIR(SET, 0, 0, 2) PT_PAIR(SET, 0, 1, 0) PT_PAIR(SET, 0, 1, 1) PT_PAIR(SET, 0, 1, 2) \
    PT_PAIR(SET, 1, 0, 0) PT_PAIR(SET, 1, 0, 1) PT_PAIR(SET, 1, 0, 2) PT_PAIR(SET, 1, 1, 0) PT_PAIR(SET, 1, 1, 1) PT_PAIR(SET, 1, 1, 2)
    if (nst & 1) { PT_ALL(0) } else { PT_ALL(1) }
#undef PT_ALL
#undef PT_PAIR
#undef PT_MFMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gemm_f32_finish<true, 3>(g, acc, smem, split, tt, tn, m0, n0, 0, wm, wn, r, h, tid, 0.f, false);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The stem's 1x1 convolutions (forward and input gradient) under split products: y[n] (M x pixels) = W (M x K) . x[n] (K x pixels).
// The weight is tiny and shared by every workgroup: it comes as an IMAGE (acr_x3_image / acr_x3_image_t of the standardised
// weight), so only the activation tile -- fp32 [k][pixels] as stored, no pass over the big tensors -- is split in registers:
// half the VALU work of gemm_f32_split_kernel per MFMA (that kernel is VALU-port bound, profiles/r04_pmc_split_gemm.txt).
// Slot = [A p0 p1 p2 (12 KiB, one contiguous piece of the image) | B fp32 16 k-rows x 128 pixels (8 KiB)], ring of 3, DMA two
// stages ahead, 5 pieces per wave and stage (3 of A, 2 of B); waits, barrier and interleaving as gemm_f32_planes_kernel.
// ---------------------------------------------------------------------------------------------------------------------------------
#define W_STAGE_B (3 * P_TILE_B + S_TILE * 4)        // 20 KiB
template <int ACT>
__global__ __launch_bounds__(256, 2) void gemm_f32_wimg_kernel(const GemmF32Args g) {
    __shared__ __attribute__((aligned(1024))) float smem[P_SLOTS * W_STAGE_B / 4];      // 60 KiB
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int ntile = g.tiles_launch;
    const int t0 = acr_xcd_remap(blockIdx.x, ntile * g.nsplit);
    const int split = t0 / ntile, tt = g.tile0 + (t0 - split * ntile);
    int tm, tn;
    if (ACT == 3) { tm = tt / g.tiles_n; tn = tt - tm * g.tiles_n; }
    else tile_coords(tt, g.tiles_m, g.tiles_n, tm, tn);
    const int m0 = tm * F_BM, n0 = tn * F_BN;
    const int zs = split / g.ksplit;
    const int kbeg = (split - zs * g.ksplit) * g.k_zs, kend = min(g.K, kbeg + g.kps);      // host: multiples of 16
    const int nkb = (g.K + P_BK - 1) / P_BK;
    const char* __restrict__ pa = reinterpret_cast<const char*>(g.a) + ((int64_t)tm * nkb + kbeg / P_BK) * (3 * P_TILE_B) + wave * 3072 + lane * 16;
    const float* __restrict__ pb = g.b + (int64_t)zs * g.b_zs + (int64_t)kbeg * g.ldb;
    int offb[2];                                            // B pieces 2 wave + i: k rows 2 (2 wave + i) + (lane >> 5), 4 pixels per lane
#pragma unroll
    for (int i = 0; i < 2; ++i) offb[i] = ((wave * 2 + i) * 2 + (lane >> 5)) * (int)g.ldb + min(n0 + 4 * (lane & 31), g.N - 4);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nst = (kend - kbeg) / P_BK;
    char* sm = reinterpret_cast<char*>(smem);
    auto dma1 = [&](int st, int slot, int i) {              // i = 0..2: A pieces 3 wave + i; i = 3, 4: B pieces 2 wave + (i - 3)
        if (i < 3)
            __builtin_amdgcn_global_load_lds((glb_vp)(pa + (int64_t)st * (3 * P_TILE_B) + i * 1024), (lds_vp)(sm + slot * W_STAGE_B + (wave * 3 + i) * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((glb_vp)(pb + (int64_t)st * P_BK * g.ldb + offb[i - 3]),
                                             (lds_vp)(sm + slot * W_STAGE_B + 3 * P_TILE_B + (wave * 2 + i - 3) * 1024), 16, 0, 0);
    };
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)sm;
    const uint32_t fa = lbase + (wm * 64 + r) * 32 + (h ^ ((r >> 3) & 1)) * 16;
    const uint32_t fb = lbase + 3 * P_TILE_B + ((8 * h) * F_BN + wn * 64 + r) * 4;
#pragma unroll
    for (int i = 0; i < 5; ++i) dma1(0, 0, i);
#pragma unroll
    for (int i = 0; i < 5; ++i) dma1(min(1, nst - 1), 1, i);
    bf16x8 ap[2][2][3], bp[2][2][3];                        // [register set][block][plane]
    float rb[2][8];
#define WI_MFMA6(SET, I, J)                                                                                                  \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][2], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][2], bp[SET][J][0], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][1], bp[SET][J][1], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][1], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][1], bp[SET][J][0], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][0], acc[I][J], 0, 0, 0);
#define WI_RD32(dst, addr, OFF) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "i"(OFF))
    auto step = [&](int st, int slot, auto set_tag, auto first_tag) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");    // younger: the 5 pieces of stage st + 1
        acr_barrier_nofence();
        const int rslot = slot == 0 ? 2 : slot - 1;         // (st + 2) % 3
        const int rst = min(st + 2, nst - 1);
#pragma unroll
        for (int i = 0; i < 5; ++i) dma1(rst, rslot, i);
        const uint32_t fas = fa + slot * W_STAGE_B, fbs = fb + slot * W_STAGE_B;
        // stage st: the B fragments raw (fp32, 8 k of one pixel per lane), the A fragments as planes
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            WI_RD32(rb[j][0], fbs, 0 * F_BN * 4 + j * 128); WI_RD32(rb[j][1], fbs, 1 * F_BN * 4 + j * 128); WI_RD32(rb[j][2], fbs, 2 * F_BN * 4 + j * 128);
            WI_RD32(rb[j][3], fbs, 3 * F_BN * 4 + j * 128); WI_RD32(rb[j][4], fbs, 4 * F_BN * 4 + j * 128); WI_RD32(rb[j][5], fbs, 5 * F_BN * 4 + j * 128);
            WI_RD32(rb[j][6], fbs, 6 * F_BN * 4 + j * 128); WI_RD32(rb[j][7], fbs, 7 * F_BN * 4 + j * 128);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            PL_RD(ap[SET][i][0], fas, 0 * P_TILE_B + i * 1024); PL_RD(ap[SET][i][1], fas, 1 * P_TILE_B + i * 1024); PL_RD(ap[SET][i][2], fas, 2 * P_TILE_B + i * 1024);
        }
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[0][2]), "+v"(rb[0][3]), "+v"(rb[0][4]), "+v"(rb[0][5]), "+v"(rb[0][6]),
                     "+v"(rb[0][7]), "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[1][2]), "+v"(rb[1][3]), "+v"(rb[1][4]), "+v"(rb[1][5]), "+v"(rb[1][6]), "+v"(rb[1][7]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 lo = {rb[j][0], rb[j][1], rb[j][2], rb[j][3]}, hi = {rb[j][4], rb[j][5], rb[j][6], rb[j][7]};
            split3_bf16(lo, hi, bp[SET][j][0], bp[SET][j][1], bp[SET][j][2]);
        }
        if (!FIRST) {
            WI_MFMA6(SET ^ 1, 0, 0) WI_MFMA6(SET ^ 1, 0, 1) WI_MFMA6(SET ^ 1, 1, 0) WI_MFMA6(SET ^ 1, 1, 1)
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
    if (nst & 1) { WI_MFMA6(0, 0, 0) WI_MFMA6(0, 0, 1) WI_MFMA6(0, 1, 0) WI_MFMA6(0, 1, 1) }
    else { WI_MFMA6(1, 0, 0) WI_MFMA6(1, 0, 1) WI_MFMA6(1, 1, 0) WI_MFMA6(1, 1, 1) }
#undef WI_MFMA6
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the refills past the end
    __syncthreads();
    gemm_f32_finish<true, ACT>(g, acc, smem, split, tt, tn, m0, n0, zs, wm, wn, r, h, tid, 0.f, false);
}

// The same product for at most 64 output channels (the stem's 1x1 convolutions into / out of the 64-channel maps at 112 x 112, and
// CAM generation's large maps): with a 128-row tile the lower wave row has no outputs -- two of four waves only feed the DMA.
// Tile = 64 rows x 256 PIXELS, all four waves compute 64 x 64 on their own 64 pixels; the A stage is the upper half of the
// image's 128-row block (3 planes x 2 KiB).  Ring 3 slots x [A 6 KiB | B fp32 16 KiB]; per wave and stage 2 A pieces (wave 3
// re-fetches pieces 4, 5: identical bytes to the same place, the count stays uniform) + 4 B pieces (one k row of 256 pixels).
#define W64_BN 256
#define W64_A_B (3 * 2048)
#define W64_STAGE_B (W64_A_B + P_BK * W64_BN * 4)          // 22 KiB
template <int ACT>
__global__ __launch_bounds__(256, 2) void gemm_f32_wimg64_kernel(const GemmF32Args g) {
    __shared__ __attribute__((aligned(1024))) char sm[P_SLOTS * W64_STAGE_B];            // 66 KiB
    typedef __attribute__((address_space(3))) void* lds_vp;
    typedef const __attribute__((address_space(1))) void* glb_vp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, wn = wave;
    const int ntile = g.tiles_n;                            // one tile row (M <= 64)
    const int t0 = acr_xcd_remap(blockIdx.x, ntile * g.nsplit);
    const int split = t0 / ntile, tn = t0 - split * ntile;
    const int n0 = tn * W64_BN;
    const int zs = split / g.ksplit;
    const int kbeg = (split - zs * g.ksplit) * g.k_zs, kend = min(g.K, kbeg + g.kps);      // host: multiples of 16
    const int qa = wave < 3 ? 2 * wave : 4;
    const char* __restrict__ pa = reinterpret_cast<const char*>(g.a) + (int64_t)(kbeg / P_BK) * (3 * P_TILE_B) + lane * 16;
    const float* __restrict__ pb = g.b + (int64_t)zs * g.b_zs + (int64_t)kbeg * g.ldb + min(n0 + 4 * lane, g.N - 4);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nst = (kend - kbeg) / P_BK;
    auto issue = [&](int st, int slot) {
        char* d = sm + slot * W64_STAGE_B;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = qa + i;
            __builtin_amdgcn_global_load_lds((glb_vp)(pa + (int64_t)st * (3 * P_TILE_B) + (q >> 1) * P_TILE_B + (q & 1) * 1024), (lds_vp)(d + q * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_vp)(pb + (int64_t)(st * P_BK + 4 * wave + i) * g.ldb), (lds_vp)(d + W64_A_B + (4 * wave + i) * 1024), 16, 0, 0);
    };
    const uint32_t lbase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)sm;
    const uint32_t fa = lbase + r * 32 + (h ^ ((r >> 3) & 1)) * 16;
    const uint32_t fb = lbase + W64_A_B + ((8 * h) * W64_BN + wn * 64 + r) * 4;
    issue(0, 0);
    issue(min(1, nst - 1), 1);
    bf16x8 ap[2][2][3], bp[2][2][3];
    float rb[2][8];
#define W64_MFMA6(SET, I, J)                                                                                                 \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][2], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][2], bp[SET][J][0], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][1], bp[SET][J][1], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][1], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][1], bp[SET][J][0], acc[I][J], 0, 0, 0);                   \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[SET][I][0], bp[SET][J][0], acc[I][J], 0, 0, 0);
#define W64_RD32(dst, addr, OFF) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "i"(OFF))
    auto step = [&](int st, int slot, auto set_tag, auto first_tag) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // younger: the 6 pieces of stage st + 1
        acr_barrier_nofence();
        const int rslot = slot == 0 ? 2 : slot - 1;
        issue(min(st + 2, nst - 1), rslot);
        const uint32_t fas = fa + slot * W64_STAGE_B, fbs = fb + slot * W64_STAGE_B;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            W64_RD32(rb[j][0], fbs, 0 * W64_BN * 4 + j * 128); W64_RD32(rb[j][1], fbs, 1 * W64_BN * 4 + j * 128);
            W64_RD32(rb[j][2], fbs, 2 * W64_BN * 4 + j * 128); W64_RD32(rb[j][3], fbs, 3 * W64_BN * 4 + j * 128);
            W64_RD32(rb[j][4], fbs, 4 * W64_BN * 4 + j * 128); W64_RD32(rb[j][5], fbs, 5 * W64_BN * 4 + j * 128);
            W64_RD32(rb[j][6], fbs, 6 * W64_BN * 4 + j * 128); W64_RD32(rb[j][7], fbs, 7 * W64_BN * 4 + j * 128);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            PL_RD(ap[SET][i][0], fas, 0 * 2048 + i * 1024); PL_RD(ap[SET][i][1], fas, 1 * 2048 + i * 1024); PL_RD(ap[SET][i][2], fas, 2 * 2048 + i * 1024);
        }
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[0][2]), "+v"(rb[0][3]), "+v"(rb[0][4]), "+v"(rb[0][5]), "+v"(rb[0][6]),
                     "+v"(rb[0][7]), "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[1][2]), "+v"(rb[1][3]), "+v"(rb[1][4]), "+v"(rb[1][5]), "+v"(rb[1][6]), "+v"(rb[1][7]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 lo = {rb[j][0], rb[j][1], rb[j][2], rb[j][3]}, hi = {rb[j][4], rb[j][5], rb[j][6], rb[j][7]};
            split3_bf16(lo, hi, bp[SET][j][0], bp[SET][j][1], bp[SET][j][2]);
        }
        if (!FIRST) {
            W64_MFMA6(SET ^ 1, 0, 0) W64_MFMA6(SET ^ 1, 0, 1) W64_MFMA6(SET ^ 1, 1, 0) W64_MFMA6(SET ^ 1, 1, 1)
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
    if (nst & 1) { W64_MFMA6(0, 0, 0) W64_MFMA6(0, 0, 1) W64_MFMA6(0, 1, 0) W64_MFMA6(0, 1, 1) }
    else { W64_MFMA6(1, 0, 0) W64_MFMA6(1, 0, 1) W64_MFMA6(1, 1, 0) W64_MFMA6(1, 1, 1) }
#undef W64_MFMA6
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the refills past the end
    if (ACT == 3) {                                         // K-split small launch: raw part sums into slab `split`
        float* slab = g.c + (int64_t)split * g.M * g.ldc;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = i * 32 + acr_krow(e, h);
                    if (row < g.M && col < g.N) slab[(int64_t)row * g.ldc + col] = acc[i][j][e];
                }
            }
        return;
    }
    GemmF32Args gz = g;
    gz.c += (int64_t)zs * g.c_zs;
    if (gz.aux) gz.aux += (int64_t)zs * g.aux_zs;
    epilogue_f32<0, true>(gz, acc, 0, n0 + wn * 64, r, h);
}

// ---- the split passes (HBM-bound: 4 bytes read, 6 written per element) ----------------------------------------------------------
__device__ __forceinline__ void planes_split8(const float (&x)[8], bf16x8& p0, bf16x8& p1, bf16x8& p2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h0 = (__bf16)x[e];
        const float r1 = x[e] - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const float r2 = r1 - (float)h1;
        p0[e] = h0; p1[e] = h1; p2[e] = (__bf16)r2;
    }
}
// byte offset of the 16-byte chunk (row rr of the tile, contraction half kh) inside a plane of the tiled image
__device__ __forceinline__ int planes_chunk_off(int rr, int kh) { return rr * 32 + ((kh ^ ((rr >> 3) & 1)) << 4); }

// tiled image of x[row][k] (pitch ld floats; the operand's rows are x's rows).  Workgroup = row block rb x 4 stages (64 k);
// thread -> 4 chunks of 8 k: a row's 256 bytes are read by 8 neighbouring threads, a stage's 8 rows x 32 bytes written by 16.
__global__ __launch_bounds__(256) void planes_tile_kernel(const float* __restrict__ x, int64_t ld, int rows, int K, int nkb, char* __restrict__ img,
                                                          float* __restrict__ colpart) {
    __shared__ float red[32 * 64];
    const int kq = (nkb + 3) >> 2;
    const int rb = blockIdx.x / kq, k0 = (blockIdx.x - rb * kq) << 6;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = j * 256 + threadIdx.x, rr = c >> 3, k8 = c & 7;
        const int row = rb * 128 + rr, k = k0 + k8 * 8;
        if (k >= nkb * P_BK) continue;
        float v[8];
        if (row < rows && k + 8 <= K) {
            const float* src = x + (int64_t)row * ld + k;
            const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
            const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + 4));
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (row < rows && k + e < K) ? x[(int64_t)row * ld + k + e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] += v[e];
        bf16x8 p0, p1, p2;
        planes_split8(v, p0, p1, p2);
        char* dst = img + ((int64_t)rb * nkb + (k >> 4)) * (3 * P_TILE_B) + planes_chunk_off(rr, k8 & 1);
        *reinterpret_cast<bf16x8*>(dst) = p0;
        *reinterpret_cast<bf16x8*>(dst + P_TILE_B) = p1;
        *reinterpret_cast<bf16x8*>(dst + 2 * P_TILE_B) = p2;
    }
    if (colpart) {                                          // column sums of this row block (bias gradient part): thread = rows
        const int tid = threadIdx.x;                        // (tid >> 3) + 32 j of the 8 columns 8 (tid & 7) ..; fixed summation order
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 3) * 64 + (tid & 7) * 8 + e] = cs[e];
        __syncthreads();
        if (tid < 64 && k0 + tid < K) {
            float t = red[tid];
            for (int q = 1; q < 32; ++q) t += red[q * 64 + tid];
            colpart[(int64_t)rb * K + k0 + tid] = t;
        }
    }
}
// tiled image of the TRANSPOSE of x[rw][c] (pitch ld): operand rows = x's columns, contraction = x's rows (R of them).  64 x 64
// blocks through an fp32 LDS tile (pitch 65: the column reads are conflict-free); thread (c = tid & 63, q = tid >> 6) then
// holds the 16 contraction elements 16 q .. 16 q + 15 of operand row c0 + c = one whole stage row (32 bytes per plane).
// colpart (or null): per 64-row block of x the column sums of the block (bias gradient parts, summed in block order by
// gemm_f32_reduce1_kernel: deterministic).
__global__ __launch_bounds__(256) void planes_tile_t_kernel(const float* __restrict__ x, int64_t ld, int R, int C, int nkb, char* __restrict__ img,
                                                            float* __restrict__ colpart) {
    __shared__ float tile[64 * 65];
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int cblocks = ((C + 127) >> 7) << 1;             // whole 128-row blocks of the operand (zeros past C)
    const int rb = blockIdx.x / cblocks, cb = blockIdx.x - rb * cblocks;
    const int r0 = rb << 6, c0 = cb << 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                            // 64 rows x 16 float4
        const int e = i * 256 + tid, rr = e >> 4, c4 = (e & 15) << 2;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + rr < R) {
            const float* src = x + (int64_t)(r0 + rr) * ld + c0 + c4;
            if (c0 + c4 + 4 <= C) v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
            else
#pragma unroll
                for (int q = 0; q < 4; ++q) if (c0 + c4 + q < C) v[q] = src[q];
        }
        float* d = tile + rr * 65 + c4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    const int c = tid & 63, gq = tid >> 6;
    float v0[8], v1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { v0[e] = tile[(gq * 16 + e) * 65 + c]; v1[e] = tile[(gq * 16 + 8 + e) * 65 + c]; }
    if (colpart) {
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += v0[e];
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += v1[e];
        red[tid] = sum;
        __syncthreads();
        if (tid < 64 && c0 + tid < C) colpart[(int64_t)rb * C + c0 + tid] = (red[tid] + red[tid + 64]) + (red[tid + 128] + red[tid + 192]);
    }
    const int kb = (r0 >> 4) + gq;                          // stage of these 16 contraction elements
    if (kb >= nkb) return;
    const int orow = c0 + c;                                // operand row (rows past C inside the last 128-row block: zeros from the loads above)
    bf16x8 p0, p1, p2;
    char* dst = img + ((int64_t)(orow >> 7) * nkb + kb) * (3 * P_TILE_B);
    const int rr = orow & 127;
    planes_split8(v0, p0, p1, p2);
    char* d0 = dst + planes_chunk_off(rr, 0);
    *reinterpret_cast<bf16x8*>(d0) = p0; *reinterpret_cast<bf16x8*>(d0 + P_TILE_B) = p1; *reinterpret_cast<bf16x8*>(d0 + 2 * P_TILE_B) = p2;
    planes_split8(v1, p0, p1, p2);
    char* d1 = dst + planes_chunk_off(rr, 1);
    *reinterpret_cast<bf16x8*>(d1) = p0; *reinterpret_cast<bf16x8*>(d1 + P_TILE_B) = p1; *reinterpret_cast<bf16x8*>(d1 + 2 * P_TILE_B) = p2;
}

// MANY small images in one launch (round 5): the stem's 52 standardised convolution weights need up to two images each per step
// (W for the forward, W^T resp. the flipped / role-swapped pack for the input gradient) -- as ~130 launches of a few microseconds
// (planes_tile / planes_tile_t plus the permute copies that packed the 3x3 weights) they cost more than the passes move.  Every
// image is described by a strided view of its source: element (r, k) of the rows x K operand is
//     src[r * sr + (k / kin) * sko + (k % kin) * ski]            (kin % 8 == 0: a chunk of 8 k never straddles an outer index)
// which covers W (co x ci: sr = ci, kin = K, ski = 1), W^T (sr = 1, ski = ci), the packed 3x3 weight w[co][t * ci + c] of
// w (co, ci, 3, 3) (sr = 9 ci, kin = ci, sko = 1, ski = 9) and its input-gradient pack w[o][c][8 - t'] as (ci x 9 co)
// (src + 8, sr = 9, kin = co, sko = -1, ski = 9 ci).  Workgroup = (image, row block, 64 k) as in planes_tile_kernel; `blk` maps a
// workgroup to its image.  Reads are strided (the weights are a few MB: L2-resident), writes are the image's contiguous chunks.
struct X3ManyDesc {
    const float* src;
    char* dst;
    int32_t rows, K, sr, kin, sko, ski, wg0, nkb;
};
__global__ __launch_bounds__(256) void planes_tile_many_kernel(const X3ManyDesc* __restrict__ descs, const int32_t* __restrict__ blk) {
    const X3ManyDesc d = descs[blk[blockIdx.x]];
    const int local = (int)blockIdx.x - d.wg0;
    const int kq = (d.nkb + 3) >> 2;
    const int rb = local / kq, k0 = (local - rb * kq) << 6;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = j * 256 + threadIdx.x, rr = c >> 3, k8 = c & 7;
        const int row = rb * 128 + rr, k = k0 + k8 * 8;
        if (k >= d.nkb * P_BK) continue;
        float v[8];
        if (row < d.rows && k < d.K) {                      // K % 8 == 0 (host): the chunk is whole
            const int outer = k / d.kin, inner = k - outer * d.kin;
            const float* sp = d.src + (int64_t)row * d.sr + (int64_t)outer * d.sko + (int64_t)inner * d.ski;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = sp[(int64_t)e * d.ski];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        bf16x8 p0, p1, p2;
        planes_split8(v, p0, p1, p2);
        char* dst = d.dst + ((int64_t)rb * d.nkb + (k >> 4)) * (3 * P_TILE_B) + planes_chunk_off(rr, k8 & 1);
        *reinterpret_cast<bf16x8*>(dst) = p0;
        *reinterpret_cast<bf16x8*>(dst + P_TILE_B) = p1;
        *reinterpret_cast<bf16x8*>(dst + 2 * P_TILE_B) = p2;
    }
}
extern "C" int acr_x3_image_many(const void* descs, const int32_t* blk, int32_t nwg, void* stream) {
    ACR_CHECK_ARG(descs && blk && nwg > 0, "acr_x3_image_many: null table or empty launch");
    ACR_CHECK_ARG(((uintptr_t)descs & 7) == 0 && ((uintptr_t)blk & 3) == 0, "acr_x3_image_many: table alignment");
    hipLaunchKernelGGL(planes_tile_many_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, (const X3ManyDesc*)descs, blk);
    return acr_check_launch("acr_x3_image_many");
}

// out[i] = sum_s slab[s][i] in split order (deterministic), float4 per thread; n4 = elements / 4
__global__ __launch_bounds__(256) void gemm_f32_reduce_kernel(const float* __restrict__ ws, int nsplit, int64_t n4,
                                                              float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 s = reinterpret_cast<const f32x4*>(ws)[i];
    for (int k = 1; k < nsplit; ++k) {
        const f32x4 v = reinterpret_cast<const f32x4*>(ws)[(int64_t)k * n4 + i];
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    reinterpret_cast<f32x4*>(out)[i] = s;
}

__global__ __launch_bounds__(256) void gemm_f32_reduce1_kernel(const float* __restrict__ ws, int nsplit, int n,
                                                               float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = ws[i];
    for (int k = 1; k < nsplit; ++k) s += ws[(int64_t)k * n + i];
    out[i] = s;
}

// Tail tiles of an NT / NN product (see gemm_tail_plan): sum the `nsplit` K-parts of every tail tile in part order
// (deterministic) and apply the epilogue of epilogue_f32<ACT> -- the same expressions, so a split tile differs from an
// unsplit one only by the grouping of its fp32 sum.  One thread = 4 consecutive columns of one row.
template <int ACT>
__global__ __launch_bounds__(256) void gemm_f32_tail_epilogue_kernel(const GemmF32Args g, const float* __restrict__ ws, int ntail,
                                                                     int nsplit) {
    const int tix = blockIdx.x >> 4;                        // 16 blocks of 256 threads per 128 x 128 tile
    const int e4 = ((blockIdx.x & 15) << 8) + threadIdx.x;  // float4 index inside the tile
    const int row_t = e4 >> 5, col_t = (e4 & 31) << 2;
    const int tt = g.tile0 + tix;
    int tm, tn;
    tile_coords(tt, g.tiles_m, g.tiles_n, tm, tn);
    const int row = tm * F_BM + row_t, col = tn * F_BN + col_t;
    if (row >= g.M || col >= g.N) return;                   // host: N % 4 == 0
    const float* p = ws + (int64_t)tix * (F_BM * F_BN) + row_t * F_BN + col_t;
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    for (int k = 1; k < nsplit; ++k) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(p + (int64_t)k * ntail * (F_BM * F_BN));
        v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
    }
    if (ACT != 2 && g.bias) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(g.bias + col);
        v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
    }
    float* cp = g.c + (int64_t)row * g.ldc + col;
    if (ACT == 0) {
        if (g.aux) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(g.aux + (int64_t)row * g.ldaux + col);
            v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
        }
        *reinterpret_cast<f32x4*>(cp) = v;
    } else if (ACT == 1) {
        f32x4 d, a;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float er = erff(v[e] * 0.70710678118654752440f);
            a[e] = v[e] * 0.5f * (1.0f + er);
            d[e] = 0.5f * (1.0f + er) + v[e] * (expf(-0.5f * v[e] * v[e]) * 0.39894228040143267794f);
        }
        *reinterpret_cast<f32x4*>(g.c2 + (int64_t)row * g.ldc + col) = a;
        *reinterpret_cast<f32x4*>(cp) = d;
    } else {
        const f32x4 x = *reinterpret_cast<const f32x4*>(g.aux + (int64_t)row * g.ldaux + col);
        v[0] *= x[0]; v[1] *= x[1]; v[2] *= x[2]; v[3] *= x[3];
        *reinterpret_cast<f32x4*>(cp) = v;
    }
}

// Tile quantisation of the NT / NN products (measured, scripts/lab/gemm_tail.py): the chip holds 512 workgroups (two per CU)
// and a launch's time is a step function of its tile count in units of 256 -- 1182 tiles (every 25 120 x 768 output of the
// step) cost 2.5 rounds for 2.31 rounds of work.  Plan: the leading multiple of 256 tiles runs as usual; the R remaining tiles
// are split s ways along K (R * s <= 512, all resident at once) into fp32 slabs, and one small kernel sums the parts in order
// and applies the epilogue.  Only worth it from s = 3 on (two halves at two per CU take what R tiles at one per CU take).
struct TailPlan { int ntail, nsplit, kps; };
// x3 (products on images, gemm_f32_planes_kernel): a workgroup alone on its CU leaves every SIMD with ONE wave of six-term
// MFMA chains and nothing to cover its LDS reads, so launches of up to 256 tiles are split two ways as well.
static TailPlan gemm_tail_plan(int M, int N, int K, bool x3 = false) {
    TailPlan p = {0, 1, 0};
    if (acr_opt(ACR_OPT_GEMM_F32_NOTAIL) != 0 || (K % F_BK) != 0 || (N % 4) != 0) return p;
    const int tiles = ((M + F_BM - 1) / F_BM) * ((N + F_BN - 1) / F_BN);
    // A product of at most a third of the chip's 512 workgroup slots (CAM generation at batch 2: 18-170 tiles) is all tail:
    // every tile is K-split, up to 16 ways, so that the launch fills the CUs instead of running 24-96 chunks on a few of them.
    const bool small = x3 ? tiles <= 256 : tiles * 3 <= 512;
    const int R = small ? tiles : tiles % 256;
    if ((!small && tiles < 512) || R == 0) return p;
    const int smin = x3 && small ? 2 : 3;
    int s = 512 / R;
    if (s > (small ? 16 : 8)) s = small ? 16 : 8;
    const int maxs = K / (2 * F_BK);                         // at least two chunks per part
    if (s > maxs) s = maxs;
    if (s < smin) return p;
    const int kps = ((K + s - 1) / s + F_BK - 1) / F_BK * F_BK;
    s = (K + kps - 1) / kps;                                // every part non-empty
    if (s < smin) return p;
    p.ntail = R; p.nsplit = s; p.kps = kps;
    return p;
}

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// the LDS-DMA kernels address operands with 32-bit element offsets inside one operand
static bool off32_ok(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int mode) {
    const int64_t lim = (1ll << 31) - (1 << 20);
    const int64_t ea = (mode == ACR_GEMM_TN ? K : M) * lda, eb = (mode == ACR_GEMM_NT ? N : K) * ldb;
    return ea < lim && eb < lim;
}
struct TnPlan { int nsplit, kps; };
// Weight gradient: the token contraction is split over workgroups.  The chip holds 512 workgroups at a time (two per CU),
// so tiles x splits should fill whole rounds of 512: 576 workgroups take as long as 1024 (measured: fc1's dW with 4
// splits = 576 workgroups ran at 87 TF, the loop itself at the same 8.4k cycles per chunk as NT).  Pick the split count
// that minimises  rounds x chunks-per-split x t_chunk  +  slab traffic  (t_chunk = 3.5 us per 32-token chunk with two
// workgroups sharing a CU; slabs are written and read once at ~4 TB/s), at least 256 tokens per split.
static TnPlan tn_plan(int M, int N, int K) {
    const int tiles = ((M + F_BM - 1) / F_BM) * ((N + F_BN - 1) / F_BN);
    const int maxs = (K + 255) / 256;
    double best = 1e30;
    int bns = 1;
    for (int ns = 1; ns <= maxs && ns <= 64; ++ns) {
        const int kps = ((K + ns - 1) / ns + F_BK - 1) / F_BK * F_BK;
        const int rounds = (tiles * ns + 511) / 512;
        const double t = rounds * (kps / (double)F_BK) * 3.5e-6 + (ns > 1 ? ns * (double)M * N * 8.0 / 4e12 : 0.0);
        if (t < best * 0.999) { best = t; bns = ns; }
    }
    int kps = ((K + bns - 1) / bns + F_BK - 1) / F_BK * F_BK;
    const int ns = (K + kps - 1) / kps;
    return {ns, kps};
}

static size_t gemm_ws_base_floats(int mode, int M, int N, int K, bool x3 = false) {
    if (mode != ACR_GEMM_TN) {
        const TailPlan tp = gemm_tail_plan(M, N, K, x3);
        return (size_t)tp.ntail * tp.nsplit * (F_BM * F_BN);
    }
    const TnPlan p = tn_plan(M, N, K);
    return (size_t)p.nsplit * ((size_t)M * N + (size_t)M);
}
// Workspace of the pre-split operands (gemm_f32_planes_kernel): behind the slabs, [A planes | B planes | column-sum parts],
// every region a multiple of 16 bytes.  Off when the plane offsets would not fit 32 bits (the kernel's DMA offsets are ints).
struct PlanesPlan { bool on; int nkb; size_t a_fl, b_fl, cs_fl; };
static PlanesPlan planes_plan(int mode, int math, int M, int N, int K) {
    PlanesPlan p = {false, 0, 0, 0, 0};
    if (math != ACR_MATH_BF16X3 || acr_opt(ACR_OPT_GEMM_X3_INKERNEL) != 0) return p;
    p.on = true; p.nkb = (K + P_BK - 1) / P_BK;
    if (mode == ACR_GEMM_TN) {                              // images of a[K][M] and b[K][N] as stored: rows = the K tokens
        const size_t nrb = (size_t)(K + 127) / 128;
        p.a_fl = nrb * ((M + P_BK - 1) / P_BK) * (3 * P_TILE_B / 4);
        p.b_fl = nrb * ((N + P_BK - 1) / P_BK) * (3 * P_TILE_B / 4);
        p.cs_fl = (nrb * M + 3) / 4 * 4;
        return p;
    }
    p.a_fl = (size_t)((M + F_BM - 1) / F_BM) * p.nkb * (3 * P_TILE_B / 4);
    p.b_fl = (size_t)((N + F_BN - 1) / F_BN) * p.nkb * (3 * P_TILE_B / 4);
    return p;
}
extern "C" size_t acr_gemm_f32_ws_floats(int32_t mode, int32_t math, int32_t M, int32_t N, int32_t K) {
    const PlanesPlan pl = planes_plan(mode, math, M, N, K);
    return (gemm_ws_base_floats(mode, M, N, K, pl.on) + 3) / 4 * 4 + pl.a_fl + pl.b_fl + pl.cs_fl;
}
// out[c] = sum over the nparts row-block parts of planes_tile_t_kernel, 16 columns per workgroup, 16 threads per column each
// summing every 16th part (independent loads in flight), combined through LDS in a fixed order (deterministic)
__global__ __launch_bounds__(256) void planes_colsum_kernel(const float* __restrict__ parts, int nparts, int C, float* __restrict__ out) {
    __shared__ float red[256];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), q = threadIdx.x >> 4;
    float s = 0.f;
    if (c < C) {
#pragma unroll 8
        for (int k = q; k < nparts; k += 16) s += parts[(int64_t)k * C + c];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 16 && c < C) {
        float t = red[threadIdx.x];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k * 16 + threadIdx.x];
        out[c] = t;
    }
}
// operand rows = x's rows
static void launch_planes_tile(const float* x, int64_t ld, int rows, int K, int nkb, float* img, float* colpart, hipStream_t st) {
    const int64_t nb = (int64_t)((rows + F_BM - 1) / F_BM) * ((nkb + 3) / 4);
    hipLaunchKernelGGL(planes_tile_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, ld, rows, K, nkb, reinterpret_cast<char*>(img), colpart);
}
// operand rows = x's C columns, contraction = x's R rows; the row blocks of x cover whole stages up to nkb * 16
static void launch_planes_tile_t(const float* x, int64_t ld, int R, int C, int nkb, float* img, float* colpart, hipStream_t st) {
    const int cpad = (C + F_BM - 1) / F_BM * F_BM;            // all 128 rows of the last row block are written (zeros past C)
    const int64_t nb = (int64_t)((nkb * P_BK + 63) / 64) * (cpad / 64);
    hipLaunchKernelGGL(planes_tile_t_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, ld, R, C, nkb, reinterpret_cast<char*>(img), colpart);
}

// ---- the image API: split-product operands made once, used by several products (include/acr_hip.h "split-product images") ------
// ---------------------------------------------------------------------------------------------------------------------------------
// LayerNorm whose output LEAVES AS AN IMAGE (models/vision_transformer.py:219-222: norm1 -> attn.qkv, norm2 -> mlp.fc1): in the
// blocks LN(x) is read by exactly one consumer, a Linear that wants it as a split-product image (and keeps that image for its
// weight gradient) -- written in fp32 and re-read by an image pass it cost 4 + 4 + 6 bytes per element on top of the 4 read here;
// now 4 read + 6 written.  Thread mapping = planes_tile_kernel's: 8 lanes per row, lane k8 holds the 8 columns 64 j + 8 k8 .. of
// every 64-column group j (one 16-byte chunk per plane), 32 rows per workgroup; the row statistics are two-pass sums over the
// registers, reduced over the row's 8 lanes.  Rows >= M of the last 128-row block are written as zeros (the weight gradient
// contracts over image rows).  y = fma((x - mean) * rstd, gamma, beta) as ln_fwd_kernel; stats = (mean, rstd) per row.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int NC> __global__ __launch_bounds__(256) void ln_image_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                          const float* __restrict__ beta, char* __restrict__ img,
                                                                          float* __restrict__ stats, int M, int nkb, float eps) {
    constexpr int C = NC * 64;
    const int tid = threadIdx.x, rr = tid >> 3, k8 = tid & 7;
    const int row = blockIdx.x * 32 + rr;
    const bool live = row < M;
    float v[NC][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
        if (live) {
            const float* src = x + (int64_t)row * C + j * 64 + k8 * 8;
            a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
            b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + 4));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[j][e] = a[e]; v[j][4 + e] = b[e]; }
        s += (a[0] + a[1] + a[2] + a[3]) + (b[0] + b[1] + b[2] + b[3]);
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    const float mean = s * (1.f / (float)C);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[j][e] - mean; ss = fmaf(d, d, ss); }
    ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
    const float rstd = rsqrtf(ss * (1.f / (float)C) + eps);
    const int rb = row >> 7, rt = row & 127;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int k = j * 64 + k8 * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + k), g1 = *reinterpret_cast<const f32x4*>(gamma + k + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + k), b1 = *reinterpret_cast<const f32x4*>(beta + k + 4);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = live ? fmaf((v[j][e] - mean) * rstd, e < 4 ? g0[e] : g1[e - 4], e < 4 ? b0[e] : b1[e - 4]) : 0.f;
        bf16x8 p0, p1, p2;
        planes_split8(o, p0, p1, p2);
        char* dst = img + ((int64_t)rb * nkb + (k >> 4)) * (3 * P_TILE_B) + planes_chunk_off(rt, k8 & 1);
        *reinterpret_cast<bf16x8*>(dst) = p0;
        *reinterpret_cast<bf16x8*>(dst + P_TILE_B) = p1;
        *reinterpret_cast<bf16x8*>(dst + 2 * P_TILE_B) = p2;
    }
    if (live && k8 == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

extern "C" int acr_layernorm_image_f32(const float* x, const float* gamma, const float* beta, float* image, float* stats, int32_t M, int32_t C,
                                       float eps, void* stream) {
    ACR_CHECK_ARG(x && gamma && beta && image && stats, "acr_layernorm_image_f32: null pointer");
    ACR_CHECK_ARG(M > 0 && C > 0 && (C % 64) == 0, "acr_layernorm_image_f32: need M > 0, C %% 64 == 0 (M=%d C=%d)", M, C);
    ACR_CHECK_ARG(al16(x) && al16(gamma) && al16(beta) && al16(image), "acr_layernorm_image_f32: x / gamma / beta / image must be 16-byte aligned");
    const int nkb = C / P_BK;
    const dim3 grid((unsigned)(((M + F_BM - 1) / F_BM) * (F_BM / 32)));
    hipStream_t st = (hipStream_t)stream;
    char* img = reinterpret_cast<char*>(image);
#define LNI(NC) case NC: hipLaunchKernelGGL(ln_image_kernel<NC>, grid, dim3(256), 0, st, x, gamma, beta, img, stats, M, nkb, eps); break;
    switch (C / 64) {
        LNI(4) LNI(8) LNI(12) LNI(16)
        default:
            acr_set_error("acr_layernorm_image_f32: C = %d not instantiated (256, 512, 768, 1024: what acr_layernorm_bwd_f32 takes)", C);
            return ACR_ERR_UNSUPPORTED;
    }
#undef LNI
    return acr_check_launch("acr_layernorm_image_f32");
}

extern "C" size_t acr_x3_image_floats(int32_t rows, int32_t cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (size_t)((rows + F_BM - 1) / F_BM) * ((cols + P_BK - 1) / P_BK) * (3 * P_TILE_B / 4);
}
extern "C" size_t acr_x3_colsum_ws_floats(int32_t rows, int32_t cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (size_t)((rows + F_BM - 1) / F_BM) * cols;
}
extern "C" int acr_x3_image(const float* x, int64_t ld, int32_t rows, int32_t cols, float* image, float* colsum, float* colsum_ws, void* stream) {
    ACR_CHECK_ARG(x && image, "acr_x3_image: null pointer");
    ACR_CHECK_ARG(rows > 0 && cols > 0 && ld >= cols, "acr_x3_image: bad shape (rows=%d cols=%d ld=%lld)", rows, cols, (long long)ld);
    ACR_CHECK_ARG(al16(x) && al16(image) && (ld % 4) == 0, "acr_x3_image: x and image must be 16-byte aligned, ld %% 4 == 0");
    ACR_CHECK_ARG(!colsum || colsum_ws, "acr_x3_image: colsum needs colsum_ws (acr_x3_colsum_ws_floats)");
    hipStream_t st = (hipStream_t)stream;
    launch_planes_tile(x, ld, rows, cols, (cols + P_BK - 1) / P_BK, image, colsum ? colsum_ws : nullptr, st);
    if (colsum)
        hipLaunchKernelGGL(planes_colsum_kernel, dim3((cols + 15) / 16), dim3(256), 0, st, (const float*)colsum_ws, (rows + F_BM - 1) / F_BM, cols, colsum);
    return acr_check_launch("acr_x3_image");
}
extern "C" int acr_x3_image_t(const float* x, int64_t ld, int32_t rows, int32_t cols, float* image, void* stream) {
    ACR_CHECK_ARG(x && image, "acr_x3_image_t: null pointer");
    ACR_CHECK_ARG(rows > 0 && cols > 0 && ld >= cols, "acr_x3_image_t: bad shape (rows=%d cols=%d ld=%lld)", rows, cols, (long long)ld);
    ACR_CHECK_ARG(al16(x) && al16(image) && (ld % 4) == 0, "acr_x3_image_t: x and image must be 16-byte aligned, ld %% 4 == 0");
    launch_planes_tile_t(x, ld, rows, cols, (rows + P_BK - 1) / P_BK, image, nullptr, (hipStream_t)stream);
    return acr_check_launch("acr_x3_image_t");
}
extern "C" size_t acr_gemm_x3_ws_floats(int32_t mode, int32_t act, int32_t M, int32_t N, int32_t K) {
    if (mode == ACR_GEMM_NN) return 0;
    const size_t base = (gemm_ws_base_floats(mode, M, N, K, true) + 3) / 4 * 4;
    return base + (act == 4 ? ((size_t)((M + F_BM - 1) / F_BM) * N + 3) / 4 * 4 : 0);        // act 4: column-sum parts per tile row
}
extern "C" int acr_gemm_x3(int32_t mode, int32_t act, const float* a_img, const float* b_img, const float* bias, const float* aux, int64_t ldaux,
                           float* c, int64_t ldc, float* c2, float* colsum, int32_t M, int32_t N, int32_t K, float* ws, void* stream) {
    ACR_CHECK_ARG(a_img && b_img && (c || act == 4), "acr_gemm_x3: null pointer");
    ACR_CHECK_ARG(M > 0 && N > 0 && K > 0, "acr_gemm_x3: empty problem (M=%d N=%d K=%d)", M, N, K);
    ACR_CHECK_ARG((mode == ACR_GEMM_NT || mode == ACR_GEMM_TN) && act >= 0 && act <= 4, "acr_gemm_x3: mode must be ACR_GEMM_NT or ACR_GEMM_TN (got %d), act 0..4 (got %d)", mode, act);
    ACR_CHECK_ARG(!colsum || act == 4, "acr_gemm_x3: colsum comes with act 4 only (the image passes give it otherwise)");
    ACR_CHECK_ARG(al16(a_img) && al16(b_img) && al16(c) && (ldc % 4) == 0 && (!bias || al16(bias)) && (!aux || (al16(aux) && (ldaux % 4) == 0)) && (!c2 || al16(c2)),
                  "acr_gemm_x3: pointers must be 16-byte aligned, pitches %% 4 == 0");
    ACR_CHECK_ARG(!ws || al16(ws), "acr_gemm_x3: ws must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    GemmF32Args g;
    g.a = a_img; g.lda = 0; g.b = b_img; g.ldb = 0; g.bias = bias; g.aux = aux; g.ldaux = ldaux; g.c = c; g.ldc = ldc; g.c2 = c2;
    g.cs = nullptr; g.M = M; g.N = N;
    g.K = (K + P_BK - 1) / P_BK * P_BK;                     // the images are zero past K
    g.tiles_m = (M + F_BM - 1) / F_BM; g.tiles_n = (N + F_BN - 1) / F_BN; g.nsplit = 1; g.kps = g.K;
    g.a_zs = g.b_zs = g.c_zs = g.aux_zs = 0; g.k_zs = g.kps; g.ksplit = 1 << 30;
    g.tile0 = 0; g.tiles_launch = g.tiles_m * g.tiles_n;
    g.nkb_a = (M + P_BK - 1) / P_BK; g.nkb_b = (N + P_BK - 1) / P_BK;
    if (mode == ACR_GEMM_TN) {                              // a_img = image of a[K][M], b_img = image of b[K][N] (rows = the K tokens)
        ACR_CHECK_ARG(act == 0 && !bias && !aux, "acr_gemm_x3: TN takes no epilogue");
        ACR_CHECK_ARG(ws, "acr_gemm_x3: TN needs the acr_gemm_x3_ws_floats workspace");
        ACR_CHECK_ARG((M % 4) == 0 && (N % 4) == 0 && ldc == N, "acr_gemm_x3: TN needs M, N %% 4 == 0 and a dense output (ldc == N)");
        const TnPlan p = tn_plan(M, N, K);
        g.nsplit = p.nsplit; g.kps = p.kps; g.k_zs = p.kps;
        g.c = ws; g.ldc = N;
        hipLaunchKernelGGL(gemm_f32_planes_tn_kernel, dim3((unsigned)(g.tiles_m * g.tiles_n * p.nsplit)), dim3(256), 0, st, g);
        const int64_t n4 = (int64_t)M * N / 4;
        hipLaunchKernelGGL(gemm_f32_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, p.nsplit, n4, c);
        return acr_check_launch("acr_gemm_x3(TN)");
    }
    ACR_CHECK_ARG((act != 1 && act != 3 && act != 4) || c2, "acr_gemm_x3: act 1 / 3 / 4 need c2");
    ACR_CHECK_ARG((act != 2 && act != 4) || aux, "acr_gemm_x3: act 2 / 4 (GELU') need the saved derivative in aux");
    ACR_CHECK_ARG(act < 3 || ((N % 8) == 0 && (!aux || (ldaux % 4) == 0)), "acr_gemm_x3: image epilogues need N %% 8 == 0");
    ACR_CHECK_ARG(act != 4 || !colsum || ws, "acr_gemm_x3: act 4 with colsum needs ws");
    TailPlan tp = gemm_tail_plan(M, N, K, true);
    if (!ws) tp.ntail = 0;
    float* parts = (act == 4 && colsum) ? ws + (gemm_ws_base_floats(mode, M, N, K, true) + 3) / 4 * 4 : nullptr;
    g.cs = parts; g.img_nkb = (N + P_BK - 1) / P_BK;
    g.tiles_launch -= tp.ntail;
    if (g.tiles_launch > 0) {
        const dim3 grid((unsigned)g.tiles_launch);
        if (act == 0) hipLaunchKernelGGL((gemm_f32_planes_kernel<0>), grid, dim3(256), 0, st, g);
        else if (act == 1) hipLaunchKernelGGL((gemm_f32_planes_kernel<1>), grid, dim3(256), 0, st, g);
        else if (act == 2) hipLaunchKernelGGL((gemm_f32_planes_kernel<2>), grid, dim3(256), 0, st, g);
        else if (act == 3) hipLaunchKernelGGL((gemm_f32_planes_kernel<5>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gemm_f32_planes_kernel<6>), grid, dim3(256), 0, st, g);
    }
    if (tp.ntail) {                                         // the tail tiles, K-split into slabs, and their epilogue (gemm_tail_plan)
        GemmF32Args gt = g;
        gt.tile0 = g.tiles_launch; gt.tiles_launch = tp.ntail; gt.nsplit = tp.nsplit; gt.kps = tp.kps; gt.k_zs = tp.kps; gt.c = ws;
        hipLaunchKernelGGL((gemm_f32_planes_kernel<4>), dim3((unsigned)(tp.ntail * tp.nsplit)), dim3(256), 0, st, gt);
        GemmF32Args ge = g;
        ge.tile0 = gt.tile0;
        const dim3 egrid((unsigned)(tp.ntail * 16));
        if (act == 0) hipLaunchKernelGGL((gemm_f32_tail_epilogue_kernel<0>), egrid, dim3(256), 0, st, ge, (const float*)ws, tp.ntail, tp.nsplit);
        else if (act == 1) hipLaunchKernelGGL((gemm_f32_tail_epilogue_kernel<1>), egrid, dim3(256), 0, st, ge, (const float*)ws, tp.ntail, tp.nsplit);
        else if (act == 2) hipLaunchKernelGGL((gemm_f32_tail_epilogue_kernel<2>), egrid, dim3(256), 0, st, ge, (const float*)ws, tp.ntail, tp.nsplit);
        else if (act == 3) hipLaunchKernelGGL((gemm_x3_tail_image_kernel<5>), dim3((unsigned)tp.ntail), dim3(256), 0, st, ge, (const float*)ws, tp.ntail, tp.nsplit);
        else hipLaunchKernelGGL((gemm_x3_tail_image_kernel<6>), dim3((unsigned)tp.ntail), dim3(256), 0, st, ge, (const float*)ws, tp.ntail, tp.nsplit);
    }
    if (parts) hipLaunchKernelGGL(planes_colsum_kernel, dim3((N + 15) / 16), dim3(256), 0, st, (const float*)parts, g.tiles_m, N, colsum);
    return acr_check_launch("acr_gemm_x3");
}

extern "C" int acr_gemm_f32(int32_t mode, int32_t math, int32_t act, const float* a, int64_t lda, const float* b, int64_t ldb, const float* bias,
                            const float* aux, int64_t ldaux, float* c, int64_t ldc, float* c2, float* colsum, int32_t M, int32_t N,
                            int32_t K, float* ws, void* stream) {
    ACR_CHECK_ARG(a && b && c, "acr_gemm_f32: null pointer");
    ACR_CHECK_ARG(M > 0 && N > 0 && K > 0, "acr_gemm_f32: empty problem (M=%d N=%d K=%d)", M, N, K);
    ACR_CHECK_ARG(mode >= ACR_GEMM_NT && mode <= ACR_GEMM_TN && act >= 0 && act <= 2, "acr_gemm_f32: bad mode %d / act %d", mode, act);
    ACR_CHECK_ARG(math == ACR_MATH_F32 || math == ACR_MATH_BF16X3, "acr_gemm_f32: bad math %d", math);
    ACR_CHECK_ARG(al16(a) && al16(b) && (lda % 4) == 0 && (ldb % 4) == 0, "acr_gemm_f32: operands must be 16-byte aligned with pitches %% 4 == 0");
    hipStream_t st = (hipStream_t)stream;
    GemmF32Args g;
    g.a = a; g.lda = lda; g.b = b; g.ldb = ldb; g.bias = bias; g.aux = aux; g.ldaux = ldaux; g.c = c; g.ldc = ldc; g.c2 = c2;
    g.cs = nullptr; g.M = M; g.N = N; g.K = K;
    g.tiles_m = (M + F_BM - 1) / F_BM; g.tiles_n = (N + F_BN - 1) / F_BN; g.nsplit = 1; g.kps = (K + F_BK - 1) / F_BK * F_BK;
    g.a_zs = g.b_zs = g.c_zs = g.aux_zs = 0; g.k_zs = g.kps; g.ksplit = 1 << 30;
    g.tile0 = 0; g.tiles_launch = g.tiles_m * g.tiles_n;
    dim3 grid((unsigned)(g.tiles_m * g.tiles_n));
    if (mode == ACR_GEMM_TN) {
        // c[M,N] = a[K,M]^T b[K,N]: both operands contraction-strided; M, N are the weight's dims, K the token count
        ACR_CHECK_ARG(act == 0 && !bias && !aux, "acr_gemm_f32: TN takes no epilogue");
        ACR_CHECK_ARG(ws, "acr_gemm_f32: TN needs the acr_gemm_f32_ws_floats workspace");
        ACR_CHECK_ARG((M % 4) == 0 && (N % 4) == 0 && M >= 4 && N >= 4 && ldc == N && al16(c), "acr_gemm_f32: TN needs M, N %% 4 == 0 and a dense output (ldc == N)");
        const TnPlan p = tn_plan(M, N, K);
        g.nsplit = p.nsplit; g.kps = p.kps; g.k_zs = p.kps;
        g.c = ws; g.ldc = N;
        g.cs = colsum ? ws + (size_t)p.nsplit * M * N : nullptr;
        const PlanesPlan pl = planes_plan(mode, math, M, N, K);
        if (pl.on) {                                        // both operands split once into images, then the product on the images
            float* wp = ws + (gemm_ws_base_floats(mode, M, N, K, true) + 3) / 4 * 4;
            float* pa = wp;
            float* pb = wp + pl.a_fl;
            int rc = acr_x3_image(a, lda, K, M, pa, colsum, colsum ? wp + pl.a_fl + pl.b_fl : nullptr, stream);
            if (rc == ACR_OK) rc = acr_x3_image(b, ldb, K, N, pb, nullptr, nullptr, stream);
            if (rc == ACR_OK) rc = acr_gemm_x3(ACR_GEMM_TN, 0, pa, pb, nullptr, nullptr, 0, c, ldc, nullptr, nullptr, M, N, K, ws, stream);
            return rc;
        }
        if ((K % F_BK) == 0 && off32_ok(M, N, K, lda, ldb, mode) && acr_opt(ACR_OPT_GEMM_F32_REGSTAGE) == 0 && math == ACR_MATH_BF16X3)
            hipLaunchKernelGGL((gemm_f32_split_kernel<false, false, 3>), dim3((unsigned)(g.tiles_m * g.tiles_n * p.nsplit)), dim3(256), 0, st, g);
        else if ((K % F_BK) == 0 && off32_ok(M, N, K, lda, ldb, mode) && acr_opt(ACR_OPT_GEMM_F32_REGSTAGE) == 0)
            hipLaunchKernelGGL((gemm_f32_dma_kernel<false, false, 3>), dim3((unsigned)(g.tiles_m * g.tiles_n * p.nsplit)), dim3(256), 0, st, g);
        else
            hipLaunchKernelGGL((gemm_f32_kernel<false, false, 3>), dim3((unsigned)(g.tiles_m * g.tiles_n * p.nsplit)), dim3(256), 0, st, g);
        const int64_t n4 = (int64_t)M * N / 4;
        hipLaunchKernelGGL(gemm_f32_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, p.nsplit, n4, c);
        if (colsum)
            hipLaunchKernelGGL(gemm_f32_reduce1_kernel, dim3((M + 255) / 256), dim3(256), 0, st, (const float*)g.cs, p.nsplit, M, colsum);
        return acr_check_launch("acr_gemm_f32(TN)");
    }
    ACR_CHECK_ARG((K % 4) == 0 && K >= 4, "acr_gemm_f32: NT / NN need K %% 4 == 0 (K=%d)", K);
    if (mode == ACR_GEMM_NN) ACR_CHECK_ARG((N % 4) == 0 && N >= 4, "acr_gemm_f32: NN needs N %% 4 == 0 (N=%d)", N);
    ACR_CHECK_ARG(act != 1 || c2, "acr_gemm_f32: act 1 (GELU) needs c2");
    ACR_CHECK_ARG(act != 2 || aux, "acr_gemm_f32: act 2 (GELU') needs the saved pre-activation in aux");
#define ACR_F32_LAUNCH(AK, BK_, ACTV)                                                                              \
    do {                                                                                                            \
        if (dma && split) hipLaunchKernelGGL((gemm_f32_split_kernel<AK, BK_, ACTV>), grid, dim3(256), 0, st, g);   \
        else if (dma) hipLaunchKernelGGL((gemm_f32_dma_kernel<AK, BK_, ACTV>), grid, dim3(256), 0, st, g);          \
        else hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, ACTV>), grid, dim3(256), 0, st, g);                      \
    } while (0)
    const bool split = math == ACR_MATH_BF16X3;                   // products as six bf16 MFMA terms of a three-way split
    const bool dma = (K % F_BK) == 0 && acr_opt(ACR_OPT_GEMM_F32_REGSTAGE) == 0 && off32_ok(M, N, K, lda, ldb, mode);
    TailPlan tp = gemm_tail_plan(M, N, K);
    const bool vec_ok = al16(c) && (ldc % 4) == 0 && (!bias || al16(bias)) && (!aux || (al16(aux) && (ldaux % 4) == 0)) &&
                        (!c2 || al16(c2));
    if (!dma || !ws || !al16(ws) || !vec_ok) tp.ntail = 0;
    if (tp.ntail) {                                         // leading whole half-rounds as usual ...
        g.tiles_launch -= tp.ntail;
        grid = dim3((unsigned)g.tiles_launch);
    }
    const PlanesPlan pl = planes_plan(mode, math, M, N, K);
    if (pl.on && ws && al16(ws) && vec_ok) {                // both operands split once into images, then the product on the images
        float* wp = ws + (gemm_ws_base_floats(mode, M, N, K, true) + 3) / 4 * 4;
        float* pa = wp;
        float* pb = wp + pl.a_fl;
        int rc = acr_x3_image(a, lda, M, K, pa, nullptr, nullptr, stream);
        if (rc == ACR_OK) rc = mode == ACR_GEMM_NT ? acr_x3_image(b, ldb, N, K, pb, nullptr, nullptr, stream) : acr_x3_image_t(b, ldb, K, N, pb, stream);
        if (rc == ACR_OK) rc = acr_gemm_x3(ACR_GEMM_NT, act, pa, pb, bias, aux, ldaux, c, ldc, c2, nullptr, M, N, K, ws, stream);
        return rc;
    }
    if (g.tiles_launch == 0) {                              // a small product: every tile goes the K-split way
    } else if (mode == ACR_GEMM_NT) {
        if (act == 0) ACR_F32_LAUNCH(true, true, 0);
        else if (act == 1) ACR_F32_LAUNCH(true, true, 1);
        else ACR_F32_LAUNCH(true, true, 2);
    } else {
        if (act == 0) ACR_F32_LAUNCH(true, false, 0);
        else if (act == 1) ACR_F32_LAUNCH(true, false, 1);
        else ACR_F32_LAUNCH(true, false, 2);
    }
#undef ACR_F32_LAUNCH
    if (tp.ntail) {                                         // ... then the tail tiles, K-split into slabs, and their epilogue
        GemmF32Args gt = g;
        gt.tile0 = g.tiles_launch; gt.tiles_launch = tp.ntail; gt.nsplit = tp.nsplit; gt.kps = tp.kps; gt.k_zs = tp.kps;
        gt.c = ws;
        const dim3 tgrid((unsigned)(tp.ntail * tp.nsplit));
        if (mode == ACR_GEMM_NT && split) hipLaunchKernelGGL((gemm_f32_split_kernel<true, true, 4>), tgrid, dim3(256), 0, st, gt);
        else if (mode == ACR_GEMM_NT) hipLaunchKernelGGL((gemm_f32_dma_kernel<true, true, 4>), tgrid, dim3(256), 0, st, gt);
        else if (split) hipLaunchKernelGGL((gemm_f32_split_kernel<true, false, 4>), tgrid, dim3(256), 0, st, gt);
        else hipLaunchKernelGGL((gemm_f32_dma_kernel<true, false, 4>), tgrid, dim3(256), 0, st, gt);
        GemmF32Args ge = g;
        ge.tile0 = gt.tile0;
        const dim3 egrid((unsigned)(tp.ntail * 16));
        if (act == 0) hipLaunchKernelGGL((gemm_f32_tail_epilogue_kernel<0>), egrid, dim3(256), 0, st, ge, (const float*)ws, tp.ntail, tp.nsplit);
        else if (act == 1) hipLaunchKernelGGL((gemm_f32_tail_epilogue_kernel<1>), egrid, dim3(256), 0, st, ge, (const float*)ws, tp.ntail, tp.nsplit);
        else hipLaunchKernelGGL((gemm_f32_tail_epilogue_kernel<2>), egrid, dim3(256), 0, st, ge, (const float*)ws, tp.ntail, tp.nsplit);
    }
    return acr_check_launch("acr_gemm_f32");
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 convolutions of the ResNetV2 stem at the reference precision (models/resnetv2.py:186-190, fp32 NCHW, stride 1) on the
// same kernels, one z-slice per sample, no layout change:
//   forward  y[n] (co x hw) = W (co x ci) . x[n] (ci x hw)            A = W  [i][k] contiguous in k, B = x[n]  [k][i]
//   input    dx[n] (ci x hw) = W^T . dy[n] (co x hw) (+ addend[n])     A = W  read as [k = co][i = ci],  B = dy[n] [k][i]
//   weight   dW (co x ci) = sum_n dy[n] (co x hw) . x[n]^T            A = dy[n], B = x[n], both contiguous in the contraction
//            (hw): one fp32 slab per sample, summed in sample order (deterministic)
// ---------------------------------------------------------------------------------------------------------------
static void conv_args(GemmF32Args& g, int M, int N, int K) {
    g.bias = nullptr; g.aux = nullptr; g.ldaux = 0; g.c2 = nullptr; g.cs = nullptr; g.M = M; g.N = N; g.K = K;
    g.tiles_m = (M + F_BM - 1) / F_BM; g.tiles_n = (N + F_BN - 1) / F_BN; g.kps = (K + F_BK - 1) / F_BK * F_BK; g.k_zs = 0;
    g.a_zs = g.b_zs = g.c_zs = g.aux_zs = 0; g.ksplit = 1;
    g.tile0 = 0; g.tiles_launch = g.tiles_m * g.tiles_n;
}

// Small launches (CAM generation: two views of one image, 8-72 workgroups) under split products: the contraction is split into
// parts of at least 64 channels so that the launch fills the chip; raw part sums go to slabs [sample][part], summed in part
// order (+ addend) by conv1x1_ksum_kernel.
static int conv1x1_ksplit(int nsamp, int cout, int cin, int hw, int* kps_out, bool wide64 = false) {
    const int tiles = wide64 ? ((hw + 255) / 256) * nsamp : ((cout + F_BM - 1) / F_BM) * ((hw + F_BN - 1) / F_BN) * nsamp;
    *kps_out = (cin + S_BK - 1) / S_BK * S_BK;
    if (tiles >= 192 || (cin % F_BK) != 0) return 1;
    int ks = 512 / tiles;
    if (ks > cin / 64) ks = cin / 64;
    if (ks < 2) return 1;
    const int kps = ((cin + ks - 1) / ks + S_BK - 1) / S_BK * S_BK;
    *kps_out = kps;
    return (cin + kps - 1) / kps;
}
extern "C" size_t acr_conv1x1_ws_floats(int32_t math, int32_t nsamp, int32_t cout, int32_t cin, int32_t hw) {
    int kps;
    if (math != ACR_MATH_BF16X3) return 0;
    int ks = conv1x1_ksplit(nsamp, cout, cin, hw, &kps);
    if (cout <= 64) ks = max(ks, conv1x1_ksplit(nsamp, cout, cin, hw, &kps, true));      // acr_conv1x1_x3's 64 x 256 tiling
    return ks > 1 ? (size_t)ks * nsamp * cout * hw : 0;
}
__global__ __launch_bounds__(256) void conv1x1_ksum_kernel(const float* __restrict__ ws, int ks, int64_t per4, const float* __restrict__ addend,
                                                           float* __restrict__ y, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int64_t n = i / per4, e = i - n * per4;
    const f32x4* p = reinterpret_cast<const f32x4*>(ws) + n * ks * per4 + e;
    f32x4 s = p[0];
    for (int k = 1; k < ks; ++k) {
        const f32x4 v = p[(int64_t)k * per4];
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    if (addend) {
        const f32x4 v = reinterpret_cast<const f32x4*>(addend)[i];
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    reinterpret_cast<f32x4*>(y)[i] = s;
}

extern "C" int acr_conv1x1_f32(int32_t math, const float* w, int32_t w_transposed, const float* x, const float* addend, float* y, int32_t nsamp,
                               int32_t cout, int32_t cin, int32_t hw, float* ws, void* stream) {
    // cout / cin are the channel counts of THIS product.  w_transposed = 0: w is (cout, cin).  w_transposed = 1: w is stored
    // (cin, cout) -- the forward convolution's weight handed over as is for the input gradient, where the roles swap.
    ACR_CHECK_ARG(w && x && y, "acr_conv1x1_f32: null pointer");
    ACR_CHECK_ARG(nsamp > 0 && cout > 0 && cin > 0 && hw > 0 && (hw % 4) == 0 && (cin % 4) == 0 && (cout % 4) == 0,
                  "acr_conv1x1_f32: need hw, cin, cout %% 4 == 0 (n=%d co=%d ci=%d hw=%d)", nsamp, cout, cin, hw);
    ACR_CHECK_ARG(al16(w) && al16(x) && al16(y) && al16(addend), "acr_conv1x1_f32: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    GemmF32Args g;
    conv_args(g, cout, hw, cin);
    g.a = w; g.b = x; g.ldb = hw; g.b_zs = (int64_t)cin * hw;
    g.c = y; g.ldc = hw; g.c_zs = (int64_t)cout * hw;
    g.aux = addend; g.ldaux = hw; g.aux_zs = (int64_t)cout * hw;
    g.nsplit = nsamp;
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n * nsamp));
    const bool dma = (cin % F_BK) == 0 && acr_opt(ACR_OPT_GEMM_F32_REGSTAGE) == 0;
    ACR_CHECK_ARG(math == ACR_MATH_F32 || math == ACR_MATH_BF16X3, "acr_conv1x1_f32: bad math %d", math);
    const bool split = dma && math == ACR_MATH_BF16X3;
    int kps = 0;
    const int ks = (split && ws && al16(ws)) ? conv1x1_ksplit(nsamp, cout, cin, hw, &kps) : 1;
    if (ks > 1) {                           // K-split small launch: slabs [sample][part] of raw sums, then the part sum (+ addend)
        g.lda = w_transposed ? cout : cin;
        g.nsplit = nsamp * ks; g.ksplit = ks; g.kps = kps; g.k_zs = kps;
        g.c = ws; g.ldc = hw; g.aux = nullptr;
        const dim3 kgrid((unsigned)(g.tiles_m * g.tiles_n * nsamp * ks));
        if (!w_transposed) hipLaunchKernelGGL((gemm_f32_split_kernel<true, false, 3>), kgrid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gemm_f32_split_kernel<false, false, 3>), kgrid, dim3(256), 0, st, g);
        const int64_t per4 = (int64_t)cout * hw / 4, n4 = per4 * nsamp;
        hipLaunchKernelGGL(conv1x1_ksum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, ks, per4, addend, y, n4);
        return acr_check_launch("acr_conv1x1_f32(K-split)");
    }
    if (!w_transposed) {                    // w = (cout, cin): rows = output channels, k contiguous
        g.lda = cin;
        if (split) hipLaunchKernelGGL((gemm_f32_split_kernel<true, false, 0>), grid, dim3(256), 0, st, g);
        else if (dma) hipLaunchKernelGGL((gemm_f32_dma_kernel<true, false, 0>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gemm_f32_kernel<true, false, 0>), grid, dim3(256), 0, st, g);
    } else {                                                // w = (cin, cout) as stored by the forward conv: A[i][k] = w[k][i]
        g.lda = cout;
        if (split) hipLaunchKernelGGL((gemm_f32_split_kernel<false, false, 0>), grid, dim3(256), 0, st, g);
        else if (dma) hipLaunchKernelGGL((gemm_f32_dma_kernel<false, false, 0>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gemm_f32_kernel<false, false, 0>), grid, dim3(256), 0, st, g);
    }
    return acr_check_launch("acr_conv1x1_f32");
}

// The same convolution with the weight given as a split-product image (acr_x3_image of W (cout x cin) for the forward;
// acr_x3_image_t of the forward's W for the input gradient, where cout / cin are THIS product's): gemm_f32_wimg_kernel.
extern "C" int acr_conv1x1_x3(const float* w_img, const float* x, const float* addend, float* y, int32_t nsamp, int32_t cout, int32_t cin,
                              int32_t hw, float* ws, void* stream) {
    ACR_CHECK_ARG(w_img && x && y, "acr_conv1x1_x3: null pointer");
    ACR_CHECK_ARG(nsamp > 0 && cout > 0 && cin > 0 && hw >= 4 && (hw % 4) == 0 && (cin % P_BK) == 0 && (cout % 4) == 0,
                  "acr_conv1x1_x3: need hw, cout %% 4 == 0, cin %% 16 == 0 (n=%d co=%d ci=%d hw=%d)", nsamp, cout, cin, hw);
    ACR_CHECK_ARG(al16(w_img) && al16(x) && al16(y) && al16(addend), "acr_conv1x1_x3: 16-byte alignment");
    ACR_CHECK_ARG((int64_t)cin * hw < (1ll << 30), "acr_conv1x1_x3: sample too large for 32-bit offsets");
    hipStream_t st = (hipStream_t)stream;
    GemmF32Args g;
    conv_args(g, cout, hw, cin);
    g.a = w_img; g.lda = 0; g.b = x; g.ldb = hw; g.b_zs = (int64_t)cin * hw;
    g.c = y; g.ldc = hw; g.c_zs = (int64_t)cout * hw;
    g.aux = addend; g.ldaux = hw; g.aux_zs = (int64_t)cout * hw;
    g.nsplit = nsamp; g.kps = cin; g.k_zs = 0;
    const bool wide64 = cout <= 64 && hw >= 4;              // 64 x 256 tiles: all four waves compute (gemm_f32_wimg64_kernel)
    if (wide64) { g.tiles_m = 1; g.tiles_n = (hw + W64_BN - 1) / W64_BN; g.tiles_launch = g.tiles_n; }
    int kps = 0;
    const int ks = (ws && al16(ws)) ? conv1x1_ksplit(nsamp, cout, cin, hw, &kps, wide64) : 1;
    if (ks > 1) {                                           // K-split small launch (conv1x1_ksplit): slabs, then the part sum (+ addend)
        g.nsplit = nsamp * ks; g.ksplit = ks; g.kps = kps; g.k_zs = kps;
        g.c = ws; g.aux = nullptr;
        if (wide64) hipLaunchKernelGGL((gemm_f32_wimg64_kernel<3>), dim3((unsigned)(g.tiles_n * nsamp * ks)), dim3(256), 0, st, g);
        else
        hipLaunchKernelGGL((gemm_f32_wimg_kernel<3>), dim3((unsigned)(g.tiles_m * g.tiles_n * nsamp * ks)), dim3(256), 0, st, g);
        const int64_t per4 = (int64_t)cout * hw / 4, n4 = per4 * nsamp;
        hipLaunchKernelGGL(conv1x1_ksum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, ks, per4, addend, y, n4);
        return acr_check_launch("acr_conv1x1_x3(K-split)");
    }
    if (wide64) hipLaunchKernelGGL((gemm_f32_wimg64_kernel<0>), dim3((unsigned)(g.tiles_n * nsamp)), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_wimg_kernel<0>), dim3((unsigned)(g.tiles_m * g.tiles_n * nsamp)), dim3(256), 0, st, g);
    return acr_check_launch("acr_conv1x1_x3");
}

// pixels of a sample are additionally split so that tiles x samples x parts fills the chip's 512 workgroup slots (a 64x64
// weight at 112^2 is ONE tile per sample: 32 workgroups of 392 chunks each otherwise); at least 512 pixels per part
static int conv_wgrad_ksplit(int nsamp, int cout, int cin, int hw) {
    const int tiles = ((cout + F_BM - 1) / F_BM) * ((cin + F_BN - 1) / F_BN) * nsamp;
    int ks = 512 / tiles;
    const int maxs = hw / 512;
    if (ks > maxs) ks = maxs;
    if (ks < 1) ks = 1;
    const int kps = ((hw + ks - 1) / ks + F_BK - 1) / F_BK * F_BK;
    return (hw + kps - 1) / kps;                            // every part non-empty
}
extern "C" size_t acr_conv1x1_wgrad_f32_ws_floats(int32_t nsamp, int32_t cout, int32_t cin, int32_t hw) {
    return (size_t)nsamp * conv_wgrad_ksplit(nsamp, cout, cin, hw) * cout * cin;
}

extern "C" int acr_conv1x1_wgrad_f32(int32_t math, const float* dy, const float* x, int32_t nsamp, int32_t cout, int32_t cin, int32_t hw, float* ws,
                                     float* dw, void* stream) {
    ACR_CHECK_ARG(dy && x && ws && dw, "acr_conv1x1_wgrad_f32: null pointer");
    ACR_CHECK_ARG(nsamp > 0 && cout > 0 && cin > 0 && hw > 0 && (hw % 4) == 0 && (cin % 4) == 0 && (cout % 4) == 0,
                  "acr_conv1x1_wgrad_f32: need hw, cin, cout %% 4 == 0");
    ACR_CHECK_ARG(al16(dy) && al16(x) && al16(dw) && al16(ws), "acr_conv1x1_wgrad_f32: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    GemmF32Args g;
    conv_args(g, cout, cin, hw);
    g.a = dy; g.lda = hw; g.a_zs = (int64_t)cout * hw;
    g.b = x; g.ldb = hw; g.b_zs = (int64_t)cin * hw;
    g.c = ws; g.ldc = cin;
    const int ks = conv_wgrad_ksplit(nsamp, cout, cin, hw);
    g.ksplit = ks;
    g.kps = ((hw + ks - 1) / ks + F_BK - 1) / F_BK * F_BK;
    ACR_CHECK_ARG((int64_t)(ks - 1) * g.kps < hw, "acr_conv1x1_wgrad_f32: internal split plan");
    g.k_zs = g.kps;
    g.nsplit = nsamp * ks;
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n * g.nsplit));
    ACR_CHECK_ARG(math == ACR_MATH_F32 || math == ACR_MATH_BF16X3, "acr_conv1x1_wgrad_f32: bad math %d", math);
    // the split-product kernel advances in 16-deep stages: pixel counts that are multiples of 16 suffice (28 x 28 = 784 = 49 x 16
    // took the register-staged exact kernel before: 2.6 ms of the f32_split step)
    if ((hw % S_BK) == 0 && acr_opt(ACR_OPT_GEMM_F32_REGSTAGE) == 0 && math == ACR_MATH_BF16X3)
        hipLaunchKernelGGL((gemm_f32_split_kernel<true, true, 3>), grid, dim3(256), 0, st, g);
    else if ((hw % F_BK) == 0 && acr_opt(ACR_OPT_GEMM_F32_REGSTAGE) == 0)
        hipLaunchKernelGGL((gemm_f32_dma_kernel<true, true, 3>), grid, dim3(256), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<true, true, 3>), grid, dim3(256), 0, st, g);
    const int64_t n4 = (int64_t)cout * cin / 4;
    if (!acr_slab_sum_wide(ws, g.nsplit, n4, dw, st))
        hipLaunchKernelGGL(gemm_f32_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, g.nsplit, n4, dw);
    return acr_check_launch("acr_conv1x1_wgrad_f32");
}
