// fp32 attention with SPLIT PRODUCTS (acr_dtype ACR_F32_BF16X3): the resident-score generation (attn_f32_sres.hip -- read its
// header first) with every matrix product evaluated on v_mfma_f32_32x32x16_bf16 as six exact terms of a three-way operand
// split,
//     a = a0 + a1 + a2  (a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1): 3 x 8 = 24 mantissa bits),
//     a b ~ a0 b2 + a2 b0 + a1 b1 + a0 b1 + a1 b0 + a0 b0     (summed small to large; the dropped terms are <= 2^-24 |a b|),
// fp32 accumulate, fp32 softmax / lse2 / delta / head mean: the same numbers as the exact-fp32 MFMA chain up to fp32 rounding
// (tests/test_kernels_gpu.py::test_attention_f32[split]), at 6/16 of its matrix time.  models/vision_transformer.py:203-211 and
// its autograd backward, like the kernels it replaces.
//
// What is split where.  q, k, v (and dO in the backward) are split ONCE per call into three bf16 planes in HBM by a streaming
// kernel (x3_split_kernel: every workgroup of a sweep re-reads the K / V resp. Q / dO tiles -- splitting them inside the sweeps
// would repeat 5.5 VALU instructions per element 7 times over); P and dS are split in registers right where the accumulator
// tile becomes the next MFMA's operand (cdna_hip_programming.md "An accumulator tile as the next MFMA's operand": element j of
// lane half h of k-step s is accumulator row 16s + 8(j>>2) + 4h + (j&3)).  The planes of q, k, v live behind the score blocks
// in the caller's `scores` buffer (acr_attn_scores_floats), those of dO behind delta in `delta_ws` (acr_attn_bwd_ws_floats).
//
// LDS image of a 32-row x 64-column bf16 tile plane: unpadded 128-byte rows filled by LDS-DMA (global_load_lds_dwordx4, 8 rows
// per wave-instruction), 16-byte chunk c of row r stored in slot c ^ f(r), f(r) = x ^ ((x & 1) << 2) with x = (r >> 1) & 7, applied
// on the DMA's SOURCE address.  Conflict-free for both kinds of read (banks per MI355X_MICROARCH.md "LDS"):
//   row read   (ds_read_b128, lane (r, h) takes chunk 2s + h): the 16 lanes of a group hold 8 even + 8 odd rows whose x are
//              all different, and x -> f is a bijection;
//   transposed (ds_read_b64_tr_b16, 4 rows x 16 columns per 16-lane group): rows R, R+2 of a block share the bank half and
//              take the 16-byte slots C ^ f(R) and C ^ f(R) ^ 5 -- different aligned groups of four.
// One image serves rowop (contraction over the 64 columns) and accop (contraction over the 32 rows).
#include <type_traits>

#include "acr_common.h"
#include "attn_f32.h"

typedef __bf16 bf16_t;
typedef __attribute__((address_space(3))) void* x3_lds_vp;
typedef const __attribute__((address_space(1))) void* x3_glb_vp;

#define X3_PLANE_B 4096                    // bytes of one plane image of a 32-row tile
#define X3_TILE_B (3 * X3_PLANE_B)         // one operand tile: three planes
#define X3_SLOT_B (2 * X3_TILE_B)          // one ring slot: two operand tiles
#define X3_SB_FLOATS 1024                  // one 32 x 32 score block (layout: attn_f32_sres.hip)

#define X3_STORE_NT(p, v) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p))
#define X3_LOAD_NT(p) __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p))

struct X3Geom {
    int B, H, T, D;                        // D = H * 64: row pitch (elements) of every plane
    float scale;
    int64_t plane;                         // elements between two planes of one operand (B * T * D)
    int64_t osb, ost, osh;                 // fp32 o / dq-dk-dv strides are passed separately where needed
    int64_t sb, st, sh;                    // fp32 q / k / v (dq / dk / dv) strides
};

__device__ __forceinline__ int x3_swz(int row) {
    const int x = (row >> 1) & 7;
    return x ^ ((x & 1) << 2);
}

// ---- a = a0 + a1 + a2 --------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void x3_split1(float x, bf16_t& h0, bf16_t& h1, bf16_t& h2) {
    h0 = (bf16_t)x;
    const float r1 = x - (float)h0;
    h1 = (bf16_t)r1;
    const float r2 = r1 - (float)h1;
    h2 = (bf16_t)r2;
}
// accumulator registers 8S .. 8S+7 -> the three fragments of k-step S
template <int S>
__device__ __forceinline__ void x3_split_acc(const f32x16& z, bf16x8 (&p)[3]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bf16_t h0, h1, h2;
        x3_split1(z[8 * S + e], h0, h1, h2);
        p[0][e] = h0; p[1][e] = h1; p[2][e] = h2;
    }
}
#define X3_MFMA6(ACC, A, Bv)                                                         \
    do {                                                                             \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], Bv[2], ACC, 0, 0, 0);    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], Bv[0], ACC, 0, 0, 0);    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], Bv[1], ACC, 0, 0, 0);    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], Bv[1], ACC, 0, 0, 0);    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], Bv[0], ACC, 0, 0, 0);    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], Bv[0], ACC, 0, 0, 0);    \
    } while (0)

// ---- split kernel: up to three fp32 (B, T, H, 64) operands -> 3 bf16 planes each, dense (B, T, D) ------------------------------
struct X3SplitArgs {
    const float* src[3];
    int64_t sb, st, sh;
    bf16_t* dst;                            // operand w, plane p at dst + (3 w + p) * plane
    int64_t plane;
    int B, T, H;
};
__global__ __launch_bounds__(256) void x3_split_kernel(const X3SplitArgs a) {
    const int64_t n8 = (int64_t)a.B * a.T * a.H * 8;       // 8-element groups per operand
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int w = blockIdx.y;
    const int c8 = (int)(i & 7);
    int64_t t = i >> 3;
    const int h = (int)(t % a.H); t /= a.H;
    const int tok = (int)(t % a.T);
    const int b = (int)(t / a.T);
    const float* s = a.src[w] + (int64_t)b * a.sb + (int64_t)tok * a.st + (int64_t)h * a.sh + c8 * 8;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(s), hi = *reinterpret_cast<const f32x4*>(s + 4);
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bf16_t h0, h1, h2;
        x3_split1(e < 4 ? lo[e] : hi[e - 4], h0, h1, h2);
        p0[e] = h0; p1[e] = h1; p2[e] = h2;
    }
    bf16_t* d = a.dst + (int64_t)(3 * w) * a.plane + i * 8;
    *reinterpret_cast<bf16x8*>(d) = p0;
    *reinterpret_cast<bf16x8*>(d + a.plane) = p1;
    *reinterpret_cast<bf16x8*>(d + 2 * a.plane) = p2;
}

// generic form behind the C ABI (acr_split3_bf16): x (rows, cols) with pitch ld -> 3 dense (rows, cols) planes
__global__ __launch_bounds__(256) void x3_split2d_kernel(const float* __restrict__ x, int64_t rows, int64_t cols, int64_t ld,
                                                        bf16_t* __restrict__ dst, int64_t plane) {
    const int64_t c8n = cols >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c8n) return;
    const int64_t row = i / c8n, c8 = i - row * c8n;
    const float* s = x + row * ld + c8 * 8;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(s), hi = *reinterpret_cast<const f32x4*>(s + 4);
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bf16_t h0, h1, h2;
        x3_split1(e < 4 ? lo[e] : hi[e - 4], h0, h1, h2);
        p0[e] = h0; p1[e] = h1; p2[e] = h2;
    }
    bf16_t* d = dst + row * cols + c8 * 8;
    *reinterpret_cast<bf16x8*>(d) = p0;
    *reinterpret_cast<bf16x8*>(d + plane) = p1;
    *reinterpret_cast<bf16x8*>(d + 2 * plane) = p2;
}

// ---- tile DMA ------------------------------------------------------------------------------------------------------------------
// One operand tile = rows row0 .. row0+31 of the three planes of a (T, D) operand of one (b, h) (`base` points at token 0,
// column 0 of plane 0 for that head).  Wave w moves rows 8w .. 8w+7 of every plane: lane l -> row 8w + (l >> 3), slot l & 7.
__device__ __forceinline__ int x3_dma_off(int D, int wave, int lane) {         // element offset of the lane's 16 bytes, tile-relative
    const int row = 8 * wave + (lane >> 3);
    return row * D + (((lane & 7) ^ x3_swz(row)) << 3);
}
__device__ __forceinline__ void x3_dma_tile_i(char* lds, const bf16_t* __restrict__ row0ptr, int64_t plane, int off, int wave) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
        __builtin_amdgcn_global_load_lds((x3_glb_vp)(row0ptr + p * plane + off), (x3_lds_vp)(lds + p * X3_PLANE_B + wave * 1024), 16, 0, 0);
}
// edge form: rows clamped to Tn - 1 (rows past the end alias the last valid one: finite, and every consumer masks them)
__device__ __forceinline__ void x3_dma_tile(char* lds, const bf16_t* __restrict__ base, int64_t plane, int D, int row0, int Tn, int wave,
                                            int lane) {
    const int row = 8 * wave + (lane >> 3);
    const bf16_t* src = base + (int64_t)min(row0 + row, Tn - 1) * D + (((lane & 7) ^ x3_swz(row)) << 3);
#pragma unroll
    for (int p = 0; p < 3; ++p)
        __builtin_amdgcn_global_load_lds((x3_glb_vp)(src + p * plane), (x3_lds_vp)(lds + p * X3_PLANE_B + wave * 1024), 16, 0, 0);
}

// ---- fragment addresses (byte offsets inside one plane image; lane-dependent part, computed once per wave) -------------------------
struct X3Lane { int rowb[4]; int trb[2][2]; };
__device__ __forceinline__ X3Lane x3_lane(int lane) {
    X3Lane lb;
    const int r = lane & 31, h = lane >> 5;
    const int fr = x3_swz(r);
#pragma unroll
    for (int s = 0; s < 4; ++s) lb.rowb[s] = r * 128 + (((2 * s + h) ^ fr) << 4);
    const int i = lane & 15, g1 = (lane >> 4) & 1, q = i >> 2, p = i & 3;
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
        const int row = 8 * hi + 4 * h + q;                // + 16 s rows per k-step: f is unchanged by it
        const int f = x3_swz(row);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) lb.trb[hi][blk] = row * 128 + (((4 * blk + 2 * g1 + (p >> 1)) ^ f) << 4) + 8 * (p & 1);
    }
    return lb;
}
// rows of the tile as an MFMA operand: element j of lane (r, h) = tile[r][16 S + 8 h + j], planes 0..2
template <int TILE_OFF, int S>
__device__ __forceinline__ void x3_rowfrag(bf16x8 (&a)[3], const char* sm, const X3Lane& lb) {
#pragma unroll
    for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(sm + lb.rowb[S] + (TILE_OFF + p * X3_PLANE_B));
}
// the tile transposed: element j of lane (r, h) = tile[16 S + 8 (j >> 2) + 4 h + (j & 3)][32 BLK + r]
template <int TILE_OFF, int S, int BLK>
__device__ __forceinline__ void x3_trfrag(bf16x8 (&a)[3], const char* sm, const X3Lane& lb) {
    typedef __attribute__((address_space(3))) bf16x4* lds_p;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(sm + lb.trb[0][BLK] + (TILE_OFF + p * X3_PLANE_B + S * 2048)));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(sm + lb.trb[1][BLK] + (TILE_OFF + p * X3_PLANE_B + S * 2048)));
        a[p] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
}
// acc[krow(reg,h)][r] += sum_d tile[krow][d] * y[r][d]: tile rows = A operand, y = the lane's row held as fragments y[plane][k-step]
template <int TILE_OFF>
__device__ __forceinline__ void x3_rowop(f32x16& acc, const char* sm, const X3Lane& lb, const bf16x8 (&y)[3][4]) {
    bf16x8 a[3];
#define X3_ROWSTEP(S)                                          \
    {                                                          \
        x3_rowfrag<TILE_OFF, S>(a, sm, lb);                    \
        const bf16x8 b_[3] = {y[0][S], y[1][S], y[2][S]};      \
        X3_MFMA6(acc, a, b_);                                  \
    }
    X3_ROWSTEP(0) X3_ROWSTEP(1) X3_ROWSTEP(2) X3_ROWSTEP(3)
#undef X3_ROWSTEP
}
// z as B operand: acc[i = tile column 32 BLK + krow][j = z-lane] += sum over z's rows
template <int TILE_OFF, int BLK>
__device__ __forceinline__ void x3_accop_b(f32x16& acc, const bf16x8 (&z0)[3], const bf16x8 (&z1)[3], const char* sm, const X3Lane& lb) {
    bf16x8 a[3];
    x3_trfrag<TILE_OFF, 0, BLK>(a, sm, lb);
    X3_MFMA6(acc, a, z0);
    x3_trfrag<TILE_OFF, 1, BLK>(a, sm, lb);
    X3_MFMA6(acc, a, z1);
}
// z as A operand: acc[i = z-lane][j = tile column 32 BLK + r]
template <int TILE_OFF, int BLK>
__device__ __forceinline__ void x3_accop_a(f32x16& acc, const bf16x8 (&z0)[3], const bf16x8 (&z1)[3], const char* sm, const X3Lane& lb) {
    bf16x8 b[3];
    x3_trfrag<TILE_OFF, 0, BLK>(b, sm, lb);
    X3_MFMA6(acc, z0, b);
    x3_trfrag<TILE_OFF, 1, BLK>(b, sm, lb);
    X3_MFMA6(acc, z1, b);
}
// the lane's row (row0 + r, clamped) of an operand's planes as fragments y[plane][k-step]
__device__ __forceinline__ void x3_rows_from_global(bf16x8 (&y)[3][4], const bf16_t* __restrict__ base, int64_t plane, int D, int row0,
                                                    int Tn, int r, int h) {
    const bf16_t* p0 = base + (int64_t)min(row0 + r, Tn - 1) * D + 8 * h;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int s = 0; s < 4; ++s) y[p][s] = *reinterpret_cast<const bf16x8*>(p0 + p * plane + 16 * s);
}

__device__ __forceinline__ int64_t x3_block(int H, int NB, int b, int hd, int qb, int kb) {
    return ((((int64_t)b * H + hd) * NB + qb) * NB + kb) * X3_SB_FLOATS;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward: workgroup = (b, h, 128 queries), wave = 32 queries; K / V tile planes stream through a two-slot LDS ring; every
// 32 x 32 logit tile is written to `sres` (scaled base-2 logits, keys >= T = -inf) before the softmax consumes it
//   S^T = K Q^T (24 MFMAs)    P^T = exp2(S^T - m)    O^T += V^T P^T (24 MFMAs)
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_fwd_x3_kernel(X3Geom g, const bf16_t* __restrict__ qp, const bf16_t* __restrict__ kp,
                                                             const bf16_t* __restrict__ vp, float* __restrict__ o, float* __restrict__ lse2,
                                                             float* __restrict__ sres) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * X3_SLOT_B];        // [slot][K planes | V planes]
    const int NB = (g.T + 31) >> 5, nqt = (NB + 3) >> 2;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int qt = id % nqt; id /= nqt;
    const int hd = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = (qt * 4 + wave) * 32;
    const bool live = q0 < g.T;                            // wave-uniform: waves past the end only help with the DMA
    const int64_t pbase = (int64_t)b * g.T * g.D + (int64_t)hd * 64;
    const bf16_t* kb = kp + pbase;
    const bf16_t* vb = vp + pbase;
    x3_dma_tile(smem, kb, g.plane, g.D, 0, g.T, wave, lane);
    x3_dma_tile(smem + X3_TILE_B, vb, g.plane, g.D, 0, g.T, wave, lane);
    bf16x8 qf[3][4];
    x3_rows_from_global(qf, qp + pbase, g.plane, g.D, q0, g.T, r, h);
    float m = -INFINITY, l = 0.f;
    f32x16 o0 = {0}, o1 = {0};
    const X3Lane lb = x3_lane(lane);
    const int doff = x3_dma_off(g.D, wave, lane);
    const float c2 = g.scale * ACR_LOG2E;
    float* sblk = sres + x3_block(g.H, NB, b, hd, min(q0 >> 5, NB - 1), 0) + lane * 4;
    auto step = [&](int k0, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int KOFF = SLOT * X3_SLOT_B, VOFF = KOFF + X3_TILE_B;
        acr_dma_barrier();                                 // slot SLOT has landed; the other slot is free
        if (k0 + 64 <= g.T) {                              // next tile fully inside: precomputed lane offset, uniform base
            x3_dma_tile_i(smem + (SLOT ^ 1) * X3_SLOT_B, kb + (int64_t)(k0 + 32) * g.D, g.plane, doff, wave);
            x3_dma_tile_i(smem + (SLOT ^ 1) * X3_SLOT_B + X3_TILE_B, vb + (int64_t)(k0 + 32) * g.D, g.plane, doff, wave);
        } else if (k0 + 32 < g.T) {                        // partial last tile: clamped rows
            x3_dma_tile(smem + (SLOT ^ 1) * X3_SLOT_B, kb, g.plane, g.D, k0 + 32, g.T, wave, lane);
            x3_dma_tile(smem + (SLOT ^ 1) * X3_SLOT_B + X3_TILE_B, vb, g.plane, g.D, k0 + 32, g.T, wave, lane);
        }
        if (!live) return;
        f32x16 s = {0};
        x3_rowop<KOFF>(s, smem, lb, qf);                   // s[reg] = q.k of key k0 + krow, query q0 + r
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) s[reg] *= c2;   // scaled base-2 logits
        if (k0 + 32 > g.T) {                               // only the last key tile has keys beyond T (uniform branch)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (k0 + acr_krow(reg, h) >= g.T) s[reg] = -INFINITY;
        }
        float* sp = sblk + (int64_t)(k0 >> 5) * X3_SB_FLOATS;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 t = {s[4 * gq], s[4 * gq + 1], s[4 * gq + 2], s[4 * gq + 3]};
            X3_STORE_NT(sp + gq * 256, t);
        }
        float mx = s[0];
#pragma unroll
        for (int reg = 1; reg < 16; ++reg) mx = fmaxf(mx, s[reg]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (__any(mx > m + 8.f)) {                         // deferred rescale (attn_f32_dma.hip)
            const float mn = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            l *= alpha;
            o0 *= alpha; o1 *= alpha;
            m = mn;
        }
        float rs = 0.f;
        f32x16 p;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) { p[reg] = __builtin_amdgcn_exp2f(s[reg] - m); rs += p[reg]; }
        rs += __shfl_xor(rs, 32);
        l += rs;
        bf16x8 p0[3], p1[3];
        x3_split_acc<0>(p, p0);
        x3_split_acc<1>(p, p1);
        x3_accop_b<VOFF, 0>(o0, p0, p1, smem, lb);         // o[reg] = O^T[d = 32*blk + krow][query = r]
        x3_accop_b<VOFF, 1>(o1, p0, p1, smem, lb);
    };
    for (int k0 = 0; k0 < g.T; k0 += 64) {
        step(k0, std::integral_constant<int, 0>{});
        if (k0 + 32 < g.T) step(k0 + 32, std::integral_constant<int, 1>{});
    }
    if (live && q0 + r < g.T) {
        const float inv = 1.f / l;
        float* ob = o + (int64_t)b * g.osb + (int64_t)(q0 + r) * g.ost + (int64_t)hd * g.osh;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            f32x4 a = {o0[4 * grp] * inv, o0[4 * grp + 1] * inv, o0[4 * grp + 2] * inv, o0[4 * grp + 3] * inv};
            f32x4 c = {o1[4 * grp] * inv, o1[4 * grp + 1] * inv, o1[4 * grp + 2] * inv, o1[4 * grp + 3] * inv};
            *reinterpret_cast<f32x4*>(ob + 8 * grp + 4 * h) = a;
            *reinterpret_cast<f32x4*>(ob + 32 + 8 * grp + 4 * h) = c;
        }
        if (h == 0) lse2[((int64_t)b * g.H + hd) * g.T + q0 + r] = m + log2f(l);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dQ: workgroup = (b, h, 128 queries); the dO planes of the wave's 32 queries in registers; K / V tile planes stream through the
// ring; the wave's score block and G rows for the NEXT step are in flight (registers) while this step's MFMAs run.
//   dP^T = V dO^T (24 MFMAs)   dS^T = exp2(S - lse2) (dP^T + G/H - delta)   dQ += dS K (24 MFMAs)
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void attn_dq_x3_body(char* smem, int bid, int nblk, const X3Geom& g, const bf16_t* __restrict__ kp,
                                                const bf16_t* __restrict__ vp, const bf16_t* __restrict__ dop,
                                                const float* __restrict__ lse2, const float* __restrict__ delta,
                                                const float* __restrict__ sres, const float* __restrict__ gm, int64_t gm_sb, int64_t gm_st,
                                                float* __restrict__ dq) {
    const int NB = (g.T + 31) >> 5, nqt = (NB + 3) >> 2;
    int id = acr_xcd_remap(bid, nblk);
    const int qt = id % nqt; id /= nqt;
    const int hd = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = qt * 128 + wave * 32;
    const bool live = q0 < g.T;
    const int64_t pbase = (int64_t)b * g.T * g.D + (int64_t)hd * 64;
    const bf16_t* kb = kp + pbase;
    const bf16_t* vb = vp + pbase;
    x3_dma_tile(smem, kb, g.plane, g.D, 0, g.T, wave, lane);
    x3_dma_tile(smem + X3_TILE_B, vb, g.plane, g.D, 0, g.T, wave, lane);
    bf16x8 dof[3][4];
    x3_rows_from_global(dof, dop + pbase, g.plane, g.D, q0, g.T, r, h);
    const bool qok = q0 + r < g.T;
    const float l2q = qok ? lse2[((int64_t)b * g.H + hd) * g.T + q0 + r] : INFINITY;    // queries beyond T: p = exp2(-inf) = 0
    const float dl = qok ? delta[((int64_t)b * g.H + hd) * g.T + q0 + r] : 0.f;
    const float invH = 1.f / (float)g.H;
    const float* grow = gm ? gm + (int64_t)b * gm_sb + (int64_t)min(q0 + r, g.T - 1) * gm_st : nullptr;
    const float* sblk = sres + x3_block(g.H, NB, b, hd, min(q0 >> 5, NB - 1), 0) + lane * 4;
    f32x16 dq0 = {0}, dq1 = {0};
    const X3Lane lb = x3_lane(lane);
    const int doff = x3_dma_off(g.D, wave, lane);
    f32x4 sbuf[2][4], gbuf[2][4];                          // [ring slot][register quad]
    auto load_sg = [&](int k0, f32x4 (&s4)[4], f32x4 (&g4)[4]) {
        const float* sp = sblk + (int64_t)(k0 >> 5) * X3_SB_FLOATS;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) s4[gq] = X3_LOAD_NT(sp + gq * 256);
        if (grow == nullptr) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) g4[gq] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else if (k0 + 32 <= g.T) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) g4[gq] = *reinterpret_cast<const f32x4*>(grow + k0 + 8 * gq + 4 * h);
        } else {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int e = 0; e < 4; ++e) g4[gq][e] = grow[min(k0 + 8 * gq + 4 * h + e, g.T - 1)];
        }
    };
    if (live) load_sg(0, sbuf[0], gbuf[0]);
    auto step = [&](int k0, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int KOFF = SLOT * X3_SLOT_B, VOFF = KOFF + X3_TILE_B;
        acr_dma_barrier();
        if (k0 + 64 <= g.T) {
            x3_dma_tile_i(smem + (SLOT ^ 1) * X3_SLOT_B, kb + (int64_t)(k0 + 32) * g.D, g.plane, doff, wave);
            x3_dma_tile_i(smem + (SLOT ^ 1) * X3_SLOT_B + X3_TILE_B, vb + (int64_t)(k0 + 32) * g.D, g.plane, doff, wave);
        } else if (k0 + 32 < g.T) {
            x3_dma_tile(smem + (SLOT ^ 1) * X3_SLOT_B, kb, g.plane, g.D, k0 + 32, g.T, wave, lane);
            x3_dma_tile(smem + (SLOT ^ 1) * X3_SLOT_B + X3_TILE_B, vb, g.plane, g.D, k0 + 32, g.T, wave, lane);
        }
        if (!live) return;
        if (k0 + 32 < g.T) load_sg(k0 + 32, sbuf[SLOT ^ 1], gbuf[SLOT ^ 1]);
        f32x16 dp = {0};
        x3_rowop<VOFF>(dp, smem, lb, dof);                 // dP^T[key = krow][query = r]
        f32x16 ds;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const float gv = gbuf[SLOT][reg >> 2][reg & 3] * invH;
            ds[reg] = __builtin_amdgcn_exp2f(sbuf[SLOT][reg >> 2][reg & 3] - l2q) * (dp[reg] + gv - dl);
        }
        bf16x8 z0[3], z1[3];
        x3_split_acc<0>(ds, z0);
        x3_split_acc<1>(ds, z1);
        x3_accop_a<KOFF, 0>(dq0, z0, z1, smem, lb);        // dQ[query = krow][d = 32*blk + r]
        x3_accop_a<KOFF, 1>(dq1, z0, z1, smem, lb);
    };
    for (int k0 = 0; k0 < g.T; k0 += 64) {
        step(k0, std::integral_constant<int, 0>{});
        if (k0 + 32 < g.T) step(k0 + 32, std::integral_constant<int, 1>{});
    }
    if (!live) return;
    const int64_t base = (int64_t)b * g.sb + (int64_t)hd * g.sh;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int qq = q0 + acr_krow(reg, h);
        if (qq < g.T) {
            float* p = dq + base + (int64_t)qq * g.st;
            p[r] = dq0[reg] * g.scale;
            p[32 + r] = dq1[reg] * g.scale;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dK, dV: workgroup = (b, h, 128 keys); the V planes of the wave's 32 keys in registers; Q / dO tile planes stream through the
// shared ring; every wave also streams ITS score blocks (q-block j x its key block) by LDS-DMA into a private two-slot ring and
// reads them transposed (key on the lane), exactly as attn_dkdv_sres_body does; lse2 / delta of the step's 32 queries sit one
// per lane and reach the accumulator rows by ds_bpermute (no LDS bytes: the two workgroups of a CU use all 160 KiB).
//   dP = dO V^T (24 MFMAs)   P = exp2(S - lse2)   dS = P (dP + G/H - delta)   dV += P^T dO (24)   dK += dS^T Q (24)
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void attn_dkdv_x3_body(char* smem, float* ssm, int bid, int nblk, const X3Geom& g, const bf16_t* __restrict__ qp,
                                                  const bf16_t* __restrict__ vp, const bf16_t* __restrict__ dop,
                                                  const float* __restrict__ lse2, const float* __restrict__ delta,
                                                  const float* __restrict__ sres, const float* __restrict__ gm, int64_t gm_sb,
                                                  int64_t gm_st, float* __restrict__ dk, float* __restrict__ dv) {
    const int NB = (g.T + 31) >> 5, nkt = (NB + 3) >> 2;
    int id = acr_xcd_remap(bid, nblk);
    const int ktile = id % nkt; id /= nkt;
    const int hd = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int key0 = ktile * 128 + wave * 32;
    const bool live = key0 < g.T;
    const int64_t pbase = (int64_t)b * g.T * g.D + (int64_t)hd * 64;
    const bf16_t* qb = qp + pbase;
    const bf16_t* dob = dop + pbase;
    const float* lrow = lse2 + ((int64_t)b * g.H + hd) * g.T;
    const float* drow = delta + ((int64_t)b * g.H + hd) * g.T;
    // score blocks of this wave: (qb = step, kb = key0 / 32); lane c of DMA piece gq fetches global chunk c ^ (2 gq + (c >> 5))
    const float* scol = sres + x3_block(g.H, NB, b, hd, 0, min(key0 >> 5, NB - 1));
    const int64_t sstep = (int64_t)NB * X3_SB_FLOATS;
    int soff[4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) soff[gq] = gq * 256 + 4 * (lane ^ (2 * gq + (lane >> 5)));
    float* sw = ssm + wave * 2 * X3_SB_FLOATS;
    auto dma_scores = [&](int qblk, int slot) {
        const float* src = scol + (int64_t)qblk * sstep;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            __builtin_amdgcn_global_load_lds((x3_glb_vp)(src + soff[gq]), (x3_lds_vp)(sw + slot * X3_SB_FLOATS + gq * 256), 16, 0, 0);
    };
    x3_dma_tile(smem, qb, g.plane, g.D, 0, g.T, wave, lane);
    x3_dma_tile(smem + X3_TILE_B, dob, g.plane, g.D, 0, g.T, wave, lane);
    if (live) dma_scores(0, 0);
    bf16x8 vf[3][4];
    x3_rows_from_global(vf, vp + pbase, g.plane, g.D, key0, g.T, r, h);
    const int key = key0 + r;
    const float invH = 1.f / (float)g.H;
    const float* gb0 = gm ? gm + (int64_t)b * gm_sb : nullptr;          // uniform
    const int glane = min(key, g.T - 1) + 4 * h * (int)gm_st;            // lane part of a G address (krow = c_reg + 4h)
    const int gcl = min(key, g.T - 1);
    f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
    const X3Lane lb = x3_lane(lane);
    const int doff = x3_dma_off(g.D, wave, lane);
    // transposed score reads: lane (kappa = r, h): byte address = tb[reg & 3] + slot*4096 + 128*(reg >> 2)
    int tb[4];
    {
        const int gk = r >> 3, hk = (r >> 2) & 1, ek = r & 3, mm = 2 * gk + hk;
#pragma unroll
        for (int j = 0; j < 4; ++j) tb[j] = ((wave * 2 * X3_SB_FLOATS) + gk * 256 + 128 * hk + ek + 4 * ((j + 4 * h) ^ mm)) * 4;
    }
    const char* ssb = reinterpret_cast<const char*>(ssm);
    float gbuf[2][16];                                      // [ring slot][register]: raw G[b][q0 + krow][key]
    float lbuf[2], dbuf[2];                                 // [ring slot]: lse2 / delta of query q0 + (lane & 31)
    auto load_g = [&](int q0, float (&gv)[16]) {
        if (gb0 == nullptr) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) gv[reg] = 0.f;
        } else if (q0 + 32 <= g.T) {                       // uniform row pointer + lane offset (saddr form loads)
            const float* gq0 = gb0 + (int64_t)q0 * gm_st;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int c_reg = (reg & 3) + 8 * (reg >> 2);
                gv[reg] = (gq0 + (int64_t)c_reg * gm_st)[glane];
            }
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) gv[reg] = gb0[(int64_t)min(q0 + acr_krow(reg, h), g.T - 1) * gm_st + gcl];
        }
    };
    auto load_ld = [&](int q0, float& lv, float& dv_) {
        const int qi = min(q0 + r, g.T - 1);
        lv = lrow[qi];
        dv_ = drow[qi];
    };
    if (live) { load_g(0, gbuf[0]); load_ld(0, lbuf[0], dbuf[0]); }
    auto step = [&](int q0, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int QOFF = SLOT * X3_SLOT_B, DOOFF = QOFF + X3_TILE_B;
        acr_dma_barrier();
        if (q0 + 32 < g.T) {
            if (q0 + 64 <= g.T) {                          // next tile fully inside: precomputed lane offset, uniform base
                x3_dma_tile_i(smem + (SLOT ^ 1) * X3_SLOT_B, qb + (int64_t)(q0 + 32) * g.D, g.plane, doff, wave);
                x3_dma_tile_i(smem + (SLOT ^ 1) * X3_SLOT_B + X3_TILE_B, dob + (int64_t)(q0 + 32) * g.D, g.plane, doff, wave);
            } else {
                x3_dma_tile(smem + (SLOT ^ 1) * X3_SLOT_B, qb, g.plane, g.D, q0 + 32, g.T, wave, lane);
                x3_dma_tile(smem + (SLOT ^ 1) * X3_SLOT_B + X3_TILE_B, dob, g.plane, g.D, q0 + 32, g.T, wave, lane);
            }
            if (live) dma_scores((q0 >> 5) + 1, SLOT ^ 1);
        }
        if (!live) return;
        // G, lse2 and delta of the NEXT query block go in flight now and are consumed a whole step later
        if (q0 + 32 < g.T) { load_g(q0 + 32, gbuf[SLOT ^ 1]); load_ld(q0 + 32, lbuf[SLOT ^ 1], dbuf[SLOT ^ 1]); }
        f32x16 dp = {0};
        x3_rowop<DOOFF>(dp, smem, lb, vf);                 // dP[query = krow][key = r]
        f32x16 s;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            s[reg] = *reinterpret_cast<const float*>(ssb + tb[reg & 3] + (SLOT * X3_SB_FLOATS * 4 + 128 * (reg >> 2)));
        if (q0 + 32 > g.T) {                               // last query block: rows beyond T are junk, P = 0 there
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (q0 + acr_krow(reg, h) >= g.T) s[reg] = -INFINITY;
        }
        f32x16 p, ds;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int src = acr_krow(reg, h);              // lane that holds this accumulator row's query
            const float lq = __shfl(lbuf[SLOT], src), dq_ = __shfl(dbuf[SLOT], src);
            const float pv = __builtin_amdgcn_exp2f(s[reg] - lq);
            p[reg] = pv;
            ds[reg] = pv * (dp[reg] + gbuf[SLOT][reg] * invH - dq_);
        }
        {
            bf16x8 z0[3], z1[3];
            x3_split_acc<0>(p, z0);
            x3_split_acc<1>(p, z1);
            x3_accop_a<DOOFF, 0>(dv0, z0, z1, smem, lb);   // dV[key = krow][d = 32*blk + r]
            x3_accop_a<DOOFF, 1>(dv1, z0, z1, smem, lb);
        }
        {
            bf16x8 z0[3], z1[3];
            x3_split_acc<0>(ds, z0);
            x3_split_acc<1>(ds, z1);
            x3_accop_a<QOFF, 0>(dk0, z0, z1, smem, lb);
            x3_accop_a<QOFF, 1>(dk1, z0, z1, smem, lb);
        }
    };
    for (int q0 = 0; q0 < g.T; q0 += 64) {
        step(q0, std::integral_constant<int, 0>{});
        if (q0 + 32 < g.T) step(q0 + 32, std::integral_constant<int, 1>{});
    }
    if (!live) return;
    const int64_t base = (int64_t)b * g.sb + (int64_t)hd * g.sh;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int kk = key0 + acr_krow(reg, h);
        if (kk < g.T) {
            float* pk = dk + base + (int64_t)kk * g.st;
            float* pv = dv + base + (int64_t)kk * g.st;
            pk[r] = dk0[reg] * g.scale;
            pk[32 + r] = dk1[reg] * g.scale;
            pv[r] = dv0[reg];
            pv[32 + r] = dv1[reg];
        }
    }
}

// dK/dV and dQ in ONE launch (one partly filled last round instead of two): first half of the grid dK/dV, second half dQ.
__global__ __launch_bounds__(256, 2) void attn_bwd_x3_kernel(X3Geom g, const bf16_t* __restrict__ qp, const bf16_t* __restrict__ kp,
                                                             const bf16_t* __restrict__ vp, const bf16_t* __restrict__ dop,
                                                             const float* __restrict__ lse2, const float* __restrict__ delta,
                                                             const float* __restrict__ sres, const float* __restrict__ gm, int64_t gm_sb,
                                                             int64_t gm_st, float* __restrict__ dq, float* __restrict__ dk,
                                                             float* __restrict__ dv) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * X3_SLOT_B];                 // [slot][Q | dO planes]  resp.  [slot][K | V planes]
    __shared__ __attribute__((aligned(1024))) float ssm[4 * 2 * X3_SB_FLOATS];          // dK/dV: [wave][slot] score blocks
    const int half = (int)gridDim.x >> 1;
    const int bid = (int)blockIdx.x;
    if (bid < half)
        attn_dkdv_x3_body(smem, ssm, bid, half, g, qp, vp, dop, lse2, delta, sres, gm, gm_sb, gm_st, dk, dv);
    else
        attn_dq_x3_body(smem, bid - half, half, g, kp, vp, dop, lse2, delta, sres, gm, gm_sb, gm_st, dq);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// launchers (called from attn_f32.hip)
// ---------------------------------------------------------------------------------------------------------------------------------
static int64_t x3_score_floats(const AttnGeom& g) {
    const int64_t nb = (g.T + 31) / 32;
    return (int64_t)g.B * g.H * nb * nb * X3_SB_FLOATS;
}
int64_t acr_attn_x3_scores_floats(const AttnGeom& g) {                  // score blocks + the 9 planes of q, k, v
    return x3_score_floats(g) + 9 * ((int64_t)g.B * g.T * g.H * 64) / 2;
}
int64_t acr_attn_x3_bwd_ws_floats(const AttnGeom& g) {                  // delta (rounded to 16 bytes) + the 3 planes of dO
    return (((int64_t)g.B * g.H * g.T + 3) & ~(int64_t)3) + 3 * ((int64_t)g.B * g.T * g.H * 64) / 2;
}
static X3Geom x3_geom(const AttnGeom& g) {
    X3Geom x;
    x.B = g.B; x.H = g.H; x.T = g.T; x.D = g.H * 64; x.scale = g.scale;
    x.plane = (int64_t)g.B * g.T * x.D;
    x.osb = g.osb; x.ost = g.ost; x.osh = g.osh;
    x.sb = g.sb; x.st = g.st; x.sh = g.sh;
    return x;
}

void acr_attn_fwd_f32_x3(const AttnGeom& g, const float* q, const float* k, const float* v, float* o, float* lse2, float* scores,
                         float* pmean, int64_t pmean_sb, int64_t pmean_st, hipStream_t st) {
    const X3Geom x = x3_geom(g);
    bf16_t* planes = reinterpret_cast<bf16_t*>(scores + x3_score_floats(g));
    X3SplitArgs a;
    a.src[0] = q; a.src[1] = k; a.src[2] = v;
    a.sb = g.sb; a.st = g.st; a.sh = g.sh; a.dst = planes; a.plane = x.plane; a.B = g.B; a.T = g.T; a.H = g.H;
    const int64_t n8 = (int64_t)g.B * g.T * g.H * 8;
    hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)((n8 + 255) / 256), 3), dim3(256), 0, st, a);
    const int NB = (g.T + 31) / 32;
    hipLaunchKernelGGL(attn_fwd_x3_kernel, dim3(g.B * g.H * ((NB + 3) / 4)), dim3(256), 0, st, x, (const bf16_t*)planes,
                       (const bf16_t*)(planes + 3 * x.plane), (const bf16_t*)(planes + 6 * x.plane), o, lse2, scores);
    if (pmean) acr_attn_pmean_sres(g, scores, lse2, pmean, pmean_sb, pmean_st, st);
}

void acr_attn_bwd_f32_x3(const AttnGeom& g, const float* o, const float* d_o, const float* lse2, const float* scores, const float* gm,
                         int64_t gm_sb, int64_t gm_st, float* dq, float* dk, float* dv, float* delta_ws, hipStream_t st) {
    const X3Geom x = x3_geom(g);
    const bf16_t* planes = reinterpret_cast<const bf16_t*>(scores + x3_score_floats(g));
    bf16_t* dop = reinterpret_cast<bf16_t*>(delta_ws + (((int64_t)g.B * g.H * g.T + 3) & ~(int64_t)3));
    X3SplitArgs a;
    a.src[0] = d_o; a.src[1] = a.src[2] = nullptr;
    a.sb = g.osb; a.st = g.ost; a.sh = g.osh; a.dst = dop; a.plane = x.plane; a.B = g.B; a.T = g.T; a.H = g.H;
    const int64_t n8 = (int64_t)g.B * g.T * g.H * 8;
    hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)((n8 + 255) / 256), 1), dim3(256), 0, st, a);
    acr_attn_delta_sres(g, scores, o, d_o, lse2, gm, gm_sb, gm_st, delta_ws, st);
    const int NB = (g.T + 31) / 32;
    const int nmain = g.B * g.H * ((NB + 3) / 4);
    hipLaunchKernelGGL(attn_bwd_x3_kernel, dim3(2 * nmain), dim3(256), 0, st, x, planes, planes + 3 * x.plane, planes + 6 * x.plane,
                       (const bf16_t*)dop, lse2, (const float*)delta_ws, scores, gm, gm_sb, gm_st, dq, dk, dv);
}

extern "C" int acr_split3_bf16(const float* x, int64_t rows, int64_t cols, int64_t ld, void* planes, int64_t plane_stride, void* stream) {
    ACR_CHECK_ARG(x && planes, "acr_split3_bf16: null pointer");
    ACR_CHECK_ARG(rows > 0 && cols > 0 && (cols % 8) == 0 && (ld % 4) == 0 && ld >= cols, "acr_split3_bf16: need cols %% 8 == 0, ld %% 4 == 0, ld >= cols");
    ACR_CHECK_ARG(plane_stride >= rows * cols && (plane_stride % 8) == 0, "acr_split3_bf16: plane stride must cover rows * cols and be a multiple of 8");
    ACR_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)planes & 15) == 0, "acr_split3_bf16: 16-byte alignment");
    const int64_t n = rows * (cols >> 3);
    ACR_CHECK_ARG((n + 255) / 256 < (1ll << 31), "acr_split3_bf16: too large");
    hipLaunchKernelGGL(x3_split2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld,
                       (bf16_t*)planes, plane_stride);
    return acr_check_launch("acr_split3_bf16");
}
