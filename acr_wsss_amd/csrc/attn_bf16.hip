// Attention kernels on the gfx950 bf16 MFMA (v_mfma_f32_32x32x16_bf16): the training-throughput precision.
//
// Same kernel set and the same fragment algebra as attn_f32.hip (read its header first); what changes:
//   * operands are 8 x bf16 per lane (k = 8*(l>>5) + j), fp32 accumulate; 4 MFMAs per 32x32x64 product
//     instead of 32, so VALU (softmax) and LDS traffic matter and tiles are 64 keys / 64 queries per step;
//   * rowop reads its LDS operand with ds_read_b128 (row pitch 144 B: 16 consecutive rows hit 16 distinct
//     16-byte slots -> conflict-free for every ds_read_b128 lane group);
//   * accop ("accumulator tile is the next MFMA's operand"): P / dS leave the accumulator as bf16x8 fragments
//     whose k-slot j of lane-half h is row 16s + 8(j>>2) + 4h + (j&3) of the tile, so the other operand
//     (V, K, Q or dO rows -- contraction index = row, i.e. k-strided in memory) is fetched with the hardware
//     transpose read ds_read_b64_tr_b16 from the SAME row-major LDS image (2-way conflict at pitch 144 B:
//     measured irrelevant next to the softmax VALU work);
//   * the softmax scale rides in the exp2 argument: p = exp2(fma(s, scale*log2e, -m)) -- q is not re-rounded;
//   * P and dS are rounded to bf16 for the second-stage products (standard flash-attention numerics), row
//     sums / log-sum-exp / delta / the head-mean output stay fp32.
// The fp32-accumulated head-mean map (B,T,T) and lse2 have the same meaning as on the fp32 path.
#include <type_traits>

#include "acr_common.h"

typedef __bf16 bf16_t;
#define BP 72                       // LDS row pitch in bf16 elements (144 B)
#define GP 68                       // LDS row pitch of the staged fp32 G tile (floats)

struct AttnGeomB {
    int B, H, T;
    float scale;
    int64_t sb, st, sh;
    int64_t osb, ost, osh;
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }


// Compiler-level fence: global loads issued before it stay before it, LDS stores after it stay after it.  Without
// it hipcc sinks every staging load next to its ds_write and drains vmcnt(0) per 16-byte chunk.
#define ACR_MEMBAR() asm volatile("" ::: "memory")

// 64-row x 64-col bf16 tile, two-phase staging (global -> registers now, registers -> LDS one iteration later)
template <int NT>
struct TileRegs { bf16x8 v[512 / NT]; };

// EDGE = false: the whole 64-row tile is inside [0, Tn) -- plain pointer arithmetic, no clamps (the clamp + select
// versions cost ~190 of 640 instructions per step of the dQ loop).  EDGE = true: rows are clamped to Tn-1, i.e. rows
// past the end alias the last valid row; every consumer masks them (key >= T -> p = 0 / -inf, query >= T -> lse = +inf
// -> p = 0, or the output row is simply not stored), so no zero-fill is needed and garbage stays finite.
template <int NT, bool EDGE = true>
__device__ __forceinline__ void tile_gload(TileRegs<NT>& t, const bf16_t* g, int64_t st, int row0, int Tn, int tid) {
    if (!EDGE) {
        // uniform base (row0 and the per-instruction row step are wave-uniform -> SGPR pair) + ONE 32-bit per-thread byte
        // offset: saddr-form loads.  The per-slot 64-bit form cost 7 VALU instructions of address arithmetic per load,
        // ~56 of a sweep step's ~300.
        const unsigned off0 = (unsigned)(((int64_t)(tid >> 3) * st + (tid & 7) * 8) * 2);
        const char* ub = reinterpret_cast<const char*>(g + (int64_t)row0 * st);
#pragma unroll
        for (int i = 0; i < 512 / NT; ++i)
            t.v[i] = *reinterpret_cast<const bf16x8*>(ub + (int64_t)(i * (NT >> 3)) * st * 2 + off0);
        return;
    }
#pragma unroll
    for (int i = 0; i < 512 / NT; ++i) {
        const int slot = tid + i * NT;
        const int row = row0 + (slot >> 3);
        const int rc = min(row, Tn - 1);
        t.v[i] = *reinterpret_cast<const bf16x8*>(g + (int64_t)rc * st + (slot & 7) * 8);
    }
}
template <int NT>
__device__ __forceinline__ void tile_lstore(bf16_t* lds, const TileRegs<NT>& t, int row0, int Tn, int tid) {
    (void)row0; (void)Tn;
#pragma unroll
    for (int i = 0; i < 512 / NT; ++i) {
        const int slot = tid + i * NT;
        *reinterpret_cast<bf16x8*>(lds + (slot >> 3) * BP + (slot & 7) * 8) = t.v[i];
    }
}
template <int ROWS, int NTHREADS>
__device__ __forceinline__ void stage_tile_bf(bf16_t* lds, const bf16_t* g, int64_t st, int row0, int Tn, int tid) {
    static_assert(ROWS == 64, "64-row tiles");
    TileRegs<NTHREADS> t;
    tile_gload<NTHREADS>(t, g, st, row0, Tn, tid);
    ACR_MEMBAR();
    tile_lstore<NTHREADS>(lds, t, row0, Tn, tid);
}

// lane (r, h) owns row (row0 + r), k-slots 16s + 8h + j  (s = 0..3, j = 0..7)
template <bool EDGE = true>
__device__ __forceinline__ void load_rows_bf(bf16x8 (&reg)[4], const bf16_t* g, int64_t st, int row0, int Tn, int lane) {
    const int r = lane & 31, h = lane >> 5;
    const bool ok = !EDGE || row0 + r < Tn;
    const bf16_t* p = g + (int64_t)(EDGE ? min(row0 + r, Tn - 1) : row0 + r) * st + 8 * h;
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + 16 * s);
        reg[s] = ok ? v : z;
    }
}
__device__ __forceinline__ void load_rows_lds_bf(bf16x8 (&reg)[4], const bf16_t* tile, int lane) {
    const bf16_t* p = tile + (lane & 31) * BP + 8 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < 4; ++s) reg[s] = *reinterpret_cast<const bf16x8*>(p + 16 * s);
}

// acc[reg] += sum_d tile[krow(reg,h)][d] * Y[l&31][d]
__device__ __forceinline__ void mma_rowop_bf(f32x16& acc, const bf16_t* tile, const bf16x8 (&y)[4], int lane) {
    const bf16_t* ap = tile + (lane & 31) * BP + 8 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + 16 * s);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, y[s], acc, 0, 0, 0);
    }
}

// Fragment of X^T for k-step s of a 32-row tile: element j = X[16s + 8(j>>2) + 4h + (j&3)][32*blk + (l&31)]
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int s, int blk, int lane) {
    const int i = lane & 15, h = lane >> 5;
    const bf16_t* a0 = tile + (16 * s + 4 * h + (i >> 2)) * BP + 32 * blk + 16 * ((lane >> 4) & 1) + 4 * (i & 3);
    typedef __attribute__((address_space(3))) bf16x4* lds_p;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(a0 + 8 * BP));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}
template <int S>
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& z) {
    bf16x8 r = {(bf16_t)z[8 * S + 0], (bf16_t)z[8 * S + 1], (bf16_t)z[8 * S + 2], (bf16_t)z[8 * S + 3],
                (bf16_t)z[8 * S + 4], (bf16_t)z[8 * S + 5], (bf16_t)z[8 * S + 6], (bf16_t)z[8 * S + 7]};
    return r;
}
// z (lane index = output row) x tile (rows = contraction, cols 32*blk.. = output column)
__device__ __forceinline__ void mma_accop_a_bf(f32x16& acc, const f32x16& z, const bf16_t* tile, int blk, int lane) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag<0>(z), tr_frag(tile, 0, blk, lane), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acc_frag<1>(z), tr_frag(tile, 1, blk, lane), acc, 0, 0, 0);
}
// tile columns 32*blk.. = output row, z (lane index = output column)
__device__ __forceinline__ void mma_accop_b_bf(f32x16& acc, const f32x16& z, const bf16_t* tile, int blk, int lane) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(tile, 0, blk, lane), acc_frag<0>(z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(tile, 1, blk, lane), acc_frag<1>(z), acc, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// forward: 2 waves x 32 query rows, 64 keys per step, K/V tiles double-buffered in LDS with the next tile's
// global loads in flight during the current tile's MFMA + softmax work (one barrier per step)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128, 2) void attn_fwd_bf16_kernel(AttnGeomB g, const bf16_t* __restrict__ q,
                                                            const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                            bf16_t* __restrict__ o, float* __restrict__ lse2) {
    __shared__ __attribute__((aligned(16))) bf16_t kt[2][64 * BP];
    __shared__ __attribute__((aligned(16))) bf16_t vt[2][64 * BP];
    const int nqt = (g.T + 63) >> 6;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int qt = id % nqt; id /= nqt;
    const int h = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int q0 = qt * 64 + wave * 32;
    const int64_t base = (int64_t)b * g.sb + (int64_t)h * g.sh;
    const float c = g.scale * ACR_LOG2E;
    bf16x8 qreg[4];
    load_rows_bf(qreg, q + base, g.st, q0, g.T, lane);
    TileRegs<128> kr, vr;
    tile_gload<128>(kr, k + base, g.st, 0, g.T, tid);
    tile_gload<128>(vr, v + base, g.st, 0, g.T, tid);
    ACR_MEMBAR();
    tile_lstore<128>(kt[0], kr, 0, g.T, tid);
    tile_lstore<128>(vt[0], vr, 0, g.T, tid);
    __syncthreads();
    float m = -INFINITY, l = 0.f;
    f32x16 o0 = {0}, o1 = {0};
    int cur = 0;
    auto step = [&](int k0, auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        tile_gload<128, EDGE>(kr, k + base, g.st, k0 + 64, g.T, tid);      // next tile
        tile_gload<128, EDGE>(vr, v + base, g.st, k0 + 64, g.T, tid);
        ACR_MEMBAR();
        const bf16_t* ktc = kt[cur];
        const bf16_t* vtc = vt[cur];
        f32x16 s0 = {0}, s1 = {0};
        mma_rowop_bf(s0, ktc, qreg, lane);                 // s[reg] = q.k (raw) [key = k0 + 32*kb + krow][query = r]
        mma_rowop_bf(s1, ktc + 32 * BP, qreg, lane);
        if (EDGE) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                if (k0 + acr_krow(reg, hh) >= g.T) s0[reg] = -INFINITY;
                if (k0 + 32 + acr_krow(reg, hh) >= g.T) s1[reg] = -INFINITY;
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) mx = fmaxf(mx, fmaxf(s0[reg], s1[reg]));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(m, mx * c);
        const float alpha = fast_exp2(m - mn);
        float rs = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            s0[reg] = fast_exp2(fmaf(s0[reg], c, -mn));
            s1[reg] = fast_exp2(fmaf(s1[reg], c, -mn));
            rs += s0[reg] + s1[reg];
        }
        rs += __shfl_xor(rs, 32);
        l = fmaf(l, alpha, rs);
        m = mn;
        o0 *= alpha; o1 *= alpha;
        mma_accop_b_bf(o0, s0, vtc, 0, lane);              // o[reg] = O^T[d = 32*blk + krow][query = r]
        mma_accop_b_bf(o1, s0, vtc, 1, lane);
        mma_accop_b_bf(o0, s1, vtc + 32 * BP, 0, lane);
        mma_accop_b_bf(o1, s1, vtc + 32 * BP, 1, lane);
        ACR_MEMBAR();
        tile_lstore<128>(kt[cur ^ 1], kr, k0 + 64, g.T, tid);
        tile_lstore<128>(vt[cur ^ 1], vr, k0 + 64, g.T, tid);
        __syncthreads();
        cur ^= 1;
    };
    {
        int k0 = 0;
        for (; k0 + 128 <= g.T; k0 += 64) step(k0, std::false_type{});
        for (; k0 < g.T; k0 += 64) step(k0, std::true_type{});
    }
    if (q0 + r < g.T) {
        const float inv = 1.f / l;
        bf16_t* ob = o + (int64_t)b * g.osb + (int64_t)(q0 + r) * g.ost + (int64_t)h * g.osh;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            f32x4 a = {o0[4 * grp] * inv, o0[4 * grp + 1] * inv, o0[4 * grp + 2] * inv, o0[4 * grp + 3] * inv};
            f32x4 cc = {o1[4 * grp] * inv, o1[4 * grp + 1] * inv, o1[4 * grp + 2] * inv, o1[4 * grp + 3] * inv};
            acr_store4<bf16_t>(ob + 8 * grp + 4 * hh, a);
            acr_store4<bf16_t>(ob + 32 + 8 * grp + 4 * hh, cc);
        }
        if (hh == 0) lse2[((int64_t)b * g.H + h) * g.T + q0 + r] = m + log2f(l);
    }
}

// ---------------------------------------------------------------------------------------------
// 64x64 tiles of P / dO V^T (PMEAN over heads, PROBS / DPROBS per head); 4 waves as 2 x 2; the head loop of
// PMEAN is double-buffered like the key loop of the forward
// ---------------------------------------------------------------------------------------------
enum { TQKB_PMEAN = 0, TQKB_PROBS = 1, TQKB_DPROBS = 2 };

template <int MODE, bool EDGE>
__device__ __forceinline__ void tile_qk_body(const AttnGeomB& g, const bf16_t* __restrict__ xq, const bf16_t* __restrict__ xk,
                                             const float* __restrict__ lse2, float* __restrict__ out, int64_t out_sb,
                                             int64_t out_st, bf16_t (*qs)[64 * BP], bf16_t (*ks)[64 * BP], int b, int hsel,
                                             int q0, int k0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wq = wave >> 1, wk = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const bool is_q = (MODE != TQKB_DPROBS);
    const float c = g.scale * ACR_LOG2E;
    const int64_t qsb = is_q ? g.sb : g.osb, qsh = is_q ? g.sh : g.osh, qst = is_q ? g.st : g.ost;
    const int h_lo = (MODE == TQKB_PMEAN) ? 0 : hsel, h_hi = (MODE == TQKB_PMEAN) ? g.H : hsel + 1;
    TileRegs<256> qr, kr;
    tile_gload<256, EDGE>(qr, xq + (int64_t)b * qsb + (int64_t)h_lo * qsh, qst, q0, g.T, tid);
    tile_gload<256, EDGE>(kr, xk + (int64_t)b * g.sb + (int64_t)h_lo * g.sh, g.st, k0, g.T, tid);
    ACR_MEMBAR();
    tile_lstore<256>(qs[0], qr, q0, g.T, tid);
    tile_lstore<256>(ks[0], kr, k0, g.T, tid);
    __syncthreads();
    // query on the lane (S^T[key = krow][query = r]): the row log-sum-exp is ONE coalesced load per head and lane
    const int qlane = EDGE ? min(q0 + wq * 32 + r, g.T - 1) : q0 + wq * 32 + r;
    f32x16 acc = {0};
    int cur = 0;
    for (int h = h_lo; h < h_hi; ++h, cur ^= 1) {
        const int hn = min(h + 1, g.H - 1);
        if (MODE == TQKB_PMEAN) {
            tile_gload<256, EDGE>(qr, xq + (int64_t)b * qsb + (int64_t)hn * qsh, qst, q0, g.T, tid);
            tile_gload<256, EDGE>(kr, xk + (int64_t)b * g.sb + (int64_t)hn * g.sh, g.st, k0, g.T, tid);
        }
        float lv = 0.f;
        if (MODE != TQKB_DPROBS) lv = lse2[((int64_t)b * g.H + h) * g.T + qlane];
        ACR_MEMBAR();
        bf16x8 qreg[4];
        load_rows_lds_bf(qreg, qs[cur] + wq * 32 * BP, lane);
        f32x16 s = {0};
        mma_rowop_bf(s, ks[cur] + wk * 32 * BP, qreg, lane);    // s[reg] = X^T[key = krow][query = r]
        if (MODE == TQKB_DPROBS) {
            acc = s;
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[reg] += fast_exp2(fmaf(s[reg], c, -lv));
        }
        if (MODE == TQKB_PMEAN) {
            ACR_MEMBAR();
            tile_lstore<256>(qs[cur ^ 1], qr, q0, g.T, tid);
            tile_lstore<256>(ks[cur ^ 1], kr, k0, g.T, tid);
        }
        __syncthreads();
    }
    // transpose the wave's 32x32 tile through LDS (free now) so every store instruction writes 128 contiguous bytes
    float* tt = reinterpret_cast<float*>(&qs[0][0]) + wave * (32 * 33);
    const float mul = (MODE == TQKB_PMEAN) ? 1.f / (float)g.H : 1.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) tt[r * 33 + acr_krow(reg, hh)] = acc[reg] * mul;
    float* ob = (MODE == TQKB_PMEAN) ? out + (int64_t)b * out_sb : out + ((int64_t)b * g.H + hsel) * (int64_t)g.T * g.T;
    const int64_t ost = (MODE == TQKB_PMEAN) ? out_st : (int64_t)g.T;
    const int key = k0 + wk * 32 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ql = 2 * i + hh;
        const int qq = q0 + wq * 32 + ql;
        const float v = tt[ql * 33 + r];
        if (!EDGE || (qq < g.T && key < g.T)) ob[(int64_t)qq * ost + key] = v;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void attn_tile_qk_bf16_kernel(AttnGeomB g, const bf16_t* __restrict__ xq,
                                                                const bf16_t* __restrict__ xk,
                                                                const float* __restrict__ lse2,
                                                                float* __restrict__ out, int64_t out_sb, int64_t out_st) {
    __shared__ __attribute__((aligned(16))) bf16_t qs[2][64 * BP];
    __shared__ __attribute__((aligned(16))) bf16_t ks[2][64 * BP];
    const int nt = (g.T + 63) >> 6;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int kti = id % nt; id /= nt;
    const int qti = id % nt; id /= nt;
    int b, hsel;
    if (MODE == TQKB_PMEAN) { b = id; hsel = 0; } else { hsel = id % g.H; b = id / g.H; }
    const int q0 = qti * 64, k0 = kti * 64;
    if (q0 + 64 <= g.T && k0 + 64 <= g.T)
        tile_qk_body<MODE, false>(g, xq, xk, lse2, out, out_sb, out_st, qs, ks, b, hsel, q0, k0);
    else
        tile_qk_body<MODE, true>(g, xq, xk, lse2, out, out_sb, out_st, qs, ks, b, hsel, q0, k0);
}

// ---------------------------------------------------------------------------------------------
// delta[b,h,i] = rowsum(dO*O) + (1/H) sum_j P_h[i,j] G[b,i,j]
// Same skeleton as the dQ sweep (query on the lane, K tiles double-buffered in LDS, G pulled as 16-byte groups of
// the lane's own row): the row sum over keys is then an in-lane accumulation + one cross-half add, and a 64-key step
// needs 8 G loads instead of 32 coalesced scalar ones.
// ---------------------------------------------------------------------------------------------
template <bool HAS_G>
__global__ __launch_bounds__(128, 2) void attn_delta_bf16_kernel(AttnGeomB g, const bf16_t* __restrict__ q,
                                                                 const bf16_t* __restrict__ k, const bf16_t* __restrict__ o,
                                                                 const bf16_t* __restrict__ d_o,
                                                                 const float* __restrict__ lse2,
                                                                 const float* __restrict__ gm, int64_t gm_sb, int64_t gm_st,
                                                                 float* __restrict__ delta) {
    __shared__ __attribute__((aligned(16))) bf16_t kt[2][64 * BP];
    const int nqt = (g.T + 63) >> 6;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int qt = id % nqt; id /= nqt;
    const int h = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int q0 = qt * 64 + wave * 32;
    const int64_t base = (int64_t)b * g.sb + (int64_t)h * g.sh;
    const int64_t obase = (int64_t)b * g.osb + (int64_t)h * g.osh;
    const float c = g.scale * ACR_LOG2E;
    const bool qok = q0 + r < g.T;
    const int qc = min(q0 + r, g.T - 1);
    float part = 0.f;
    {
        const bf16_t* op = o + obase + (int64_t)qc * g.ost + 32 * hh;
        const bf16_t* dp = d_o + obase + (int64_t)qc * g.ost + 32 * hh;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(op + 8 * i);
            const bf16x8 d8 = *reinterpret_cast<const bf16x8*>(dp + 8 * i);
#pragma unroll
            for (int e = 0; e < 8; ++e) part = fmaf((float)a[e], (float)d8[e], part);
        }
    }
    float rho = 0.f;
    if (HAS_G) {
        bf16x8 qreg[4];
        load_rows_bf(qreg, q + base, g.st, q0, g.T, lane);
        const float l2v = lse2[((int64_t)b * g.H + h) * g.T + qc];
        const float l2 = qok ? l2v : INFINITY;              // rows beyond T: p = exp2(-inf) = 0
        const float* grow = gm + (int64_t)b * gm_sb + (int64_t)qc * gm_st + 4 * hh;
        const int gmax = (int)gm_st - 4 - 4 * hh;
        TileRegs<128> kr;
        tile_gload<128>(kr, k + base, g.st, 0, g.T, tid);
        ACR_MEMBAR();
        tile_lstore<128>(kt[0], kr, 0, g.T, tid);
        __syncthreads();
        int cur = 0;
        auto step = [&](int k0, auto edge_tag) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            tile_gload<128, EDGE>(kr, k + base, g.st, k0 + 64, g.T, tid);
            f32x4 gq[2][4];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    const int go = k0 + 32 * kb + 8 * grp;
                    __builtin_memcpy(&gq[kb][grp], grow + (EDGE ? min(go, gmax) : go), 16);
                }
            ACR_MEMBAR();
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 s = {0};
                mma_rowop_bf(s, kt[cur] + kb * 32 * BP, qreg, lane);     // S^T raw [key = krow][query = r]
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    float p = fast_exp2(fmaf(s[reg], c, -l2));
                    float gv = gq[kb][reg >> 2][reg & 3];
                    if (EDGE) {
                        const bool kv = k0 + 32 * kb + acr_krow(reg, hh) < g.T;
                        p = kv ? p : 0.f;
                        gv = kv ? gv : 0.f;
                    }
                    rho = fmaf(p, gv, rho);
                }
            }
            ACR_MEMBAR();
            tile_lstore<128>(kt[cur ^ 1], kr, k0 + 64, g.T, tid);
            __syncthreads();
            cur ^= 1;
        };
        int k0 = 0;
        for (; k0 + 128 <= g.T; k0 += 64) step(k0, std::false_type{});
        for (; k0 < g.T; k0 += 64) step(k0, std::true_type{});
    }
    // both halves of a lane pair (r, r+32) hold partial sums of the same query row
    part += __shfl_xor(part, 32);
    rho += __shfl_xor(rho, 32);
    if (hh == 0 && qok) delta[((int64_t)b * g.H + h) * g.T + q0 + r] = part + rho * (1.f / (float)g.H);
}

// ---------------------------------------------------------------------------------------------
// dQ: 2 waves x 32 query rows, 64 keys per step (K/V tiles double-buffered); each lane pulls the G values of its
// own query row as 16-byte groups (row pitch gm_st is a multiple of 4 floats) at the top of the step
// ---------------------------------------------------------------------------------------------
template <bool HAS_G>
__global__ __launch_bounds__(128, 2) void attn_dq_bf16_kernel(AttnGeomB g, const bf16_t* __restrict__ q,
                                                           const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                           const bf16_t* __restrict__ d_o,
                                                           const float* __restrict__ lse2,
                                                           float* __restrict__ delta,
                                                           const float* __restrict__ gm, int64_t gm_sb, int64_t gm_st,
                                                           bf16_t* __restrict__ dq) {
    // With G (the head-mean gradient) the flash-backward row term is delta_i = D_i + rho_i/H, D_i = rowsum(dO o O),
    // rho_i = sum_j P_ij G_ij.  rho needs a full sweep over the keys -- a separate kernel cost 120 us per layer.  Here the
    // sweep is this one: dS_ij = P_ij (dP_ij + G_ij/H - D_i) - (rho_i/H) P_ij is linear in rho_i, so
    //     dQ_i = X_i - (rho_i/H) Y_i,   X_i = sum_j P_ij (dP_ij + G_ij/H - D_i) K_j,   Y_i = sum_j P_ij K_j,
    // with X, Y and rho accumulated side by side (one extra product) and combined in fp32 at the end; the kernel takes
    // D in `delta` and overwrites it with the full delta for the dK/dV sweep that follows.  rho/H is ~1e-6 of the
    // other terms (G = +-alpha/count), so the subtraction cancels nothing and the correction stays in fp32.
    __shared__ __attribute__((aligned(16))) bf16_t kt[2][64 * BP];
    __shared__ __attribute__((aligned(16))) bf16_t vt[2][64 * BP];
    const int nqt = (g.T + 63) >> 6;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int qt = id % nqt; id /= nqt;
    const int h = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int q0 = qt * 64 + wave * 32;
    const int64_t base = (int64_t)b * g.sb + (int64_t)h * g.sh;
    const int64_t obase = (int64_t)b * g.osb + (int64_t)h * g.osh;
    const float c = g.scale * ACR_LOG2E;
    bf16x8 qreg[4], doreg[4];
    load_rows_bf(qreg, q + base, g.st, q0, g.T, lane);
    load_rows_bf(doreg, d_o + obase, g.ost, q0, g.T, lane);
    const bool qok = q0 + r < g.T;
    const int qc = min(q0 + r, g.T - 1);
    const float l2v = lse2[((int64_t)b * g.H + h) * g.T + qc];
    const float dlv = delta[((int64_t)b * g.H + h) * g.T + qc];
    const float l2 = qok ? l2v : INFINITY;                 // rows beyond T: p = exp2(-inf) = 0
    const float dl = qok ? dlv : 0.f;
    const float invH = 1.f / (float)g.H;
    const float* grow = HAS_G ? gm + (int64_t)b * gm_sb + (int64_t)qc * gm_st + 4 * hh : nullptr;
    const int gmax = (int)gm_st - 4 - 4 * hh;              // last in-row 16-byte group start (relative to grow)
    TileRegs<128> kr, vr;
    tile_gload<128>(kr, k + base, g.st, 0, g.T, tid);
    tile_gload<128>(vr, v + base, g.st, 0, g.T, tid);
    ACR_MEMBAR();
    tile_lstore<128>(kt[0], kr, 0, g.T, tid);
    tile_lstore<128>(vt[0], vr, 0, g.T, tid);
    __syncthreads();
    f32x16 dq0 = {0}, dq1 = {0};
    f32x16 y0 = {0}, y1 = {0};                              // HAS_G only
    float rho = 0.f;
    int cur = 0;
    // one step = 64 keys.  EDGE steps (the partial last tile, and the step that prefetches it) carry the clamps and
    // the key masks; all other steps are straight-line code without a single compare/select.
    auto step = [&](int k0, auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        tile_gload<128, EDGE>(kr, k + base, g.st, k0 + 64, g.T, tid);
        tile_gload<128, EDGE>(vr, v + base, g.st, k0 + 64, g.T, tid);
        f32x4 gq[2][4];
        if (HAS_G) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    const int go = k0 + 32 * kb + 8 * grp;
                    __builtin_memcpy(&gq[kb][grp], grow + (EDGE ? min(go, gmax) : go), 16);
                }
        }
        ACR_MEMBAR();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const bf16_t* ktc = kt[cur] + kb * 32 * BP;
            const bf16_t* vtc = vt[cur] + kb * 32 * BP;
            f32x16 s = {0}, dp = {0};
            mma_rowop_bf(s, ktc, qreg, lane);               // S^T raw [key = krow][query = r]
            mma_rowop_bf(dp, vtc, doreg, lane);             // dP^T
            // in place: s becomes dS and dp becomes P.  (Filling fresh f32x16 values element by element makes hipcc
            // initialise each 16-register tuple with 16 v_mov first: 64 of a step's ~250 VALU instructions.)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                float p = fast_exp2(fmaf(s[reg], c, -l2));
                float gv = HAS_G ? gq[kb][reg >> 2][reg & 3] : 0.f;
                float t = dp[reg];
                if (EDGE) {                                 // pad columns of G may hold anything: select, don't multiply
                    const bool kv = k0 + 32 * kb + acr_krow(reg, hh) < g.T;
                    p = kv ? p : 0.f;
                    t = kv ? t : 0.f;
                    gv = kv ? gv : 0.f;
                }
                if (HAS_G) {
                    t = fmaf(gv, invH, t);
                    rho = fmaf(p, gv, rho);
                    dp[reg] = p;
                }
                s[reg] = p * (t - dl);
            }
            mma_accop_a_bf(dq0, s, ktc, 0, lane);           // dQ[query = krow][d = 32*blk + r]
            mma_accop_a_bf(dq1, s, ktc, 1, lane);
            if (HAS_G) {
                mma_accop_a_bf(y0, dp, ktc, 0, lane);       // Y[query = krow][d]
                mma_accop_a_bf(y1, dp, ktc, 1, lane);
            }
        }
        ACR_MEMBAR();
        tile_lstore<128>(kt[cur ^ 1], kr, k0 + 64, g.T, tid);
        tile_lstore<128>(vt[cur ^ 1], vr, k0 + 64, g.T, tid);
        __syncthreads();
        cur ^= 1;
    };
    int k0 = 0;
    for (; k0 + 128 <= g.T; k0 += 64) step(k0, std::false_type{});
    for (; k0 < g.T; k0 += 64) step(k0, std::true_type{});
    if (HAS_G) {
        rho += __shfl_xor(rho, 32);                         // both halves of a lane pair hold keys of the same query row
        rho *= invH;                                        // lane r (either half): rho_{q0+r} / H
        if (hh == 0 && qok) delta[((int64_t)b * g.H + h) * g.T + q0 + r] = dl + rho;      // full delta for dK/dV
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int qq = q0 + acr_krow(reg, hh);
        if (HAS_G) {                                        // accumulator rows are krow(reg, hh): fetch that row's rho
            const float rr = __shfl(rho, acr_krow(reg, hh));
            dq0[reg] = fmaf(-rr, y0[reg], dq0[reg]);
            dq1[reg] = fmaf(-rr, y1[reg], dq1[reg]);
        }
        if (qq < g.T) {
            bf16_t* p = dq + base + (int64_t)qq * g.st;
            p[r] = (bf16_t)(dq0[reg] * g.scale);
            p[32 + r] = (bf16_t)(dq1[reg] * g.scale);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dQ, 4-wave variant: 4 waves x 32 query rows share every K/V tile.  The 2-wave sweep above is latency-bound -- a step
// is 32 MFMAs + ~250 VALU instructions (~1.1 k cycles) but takes ~8 k cycles per workgroup, one exposed memory round trip
// per step: the next tile's global loads are issued at the top of the step and consumed (register -> LDS) at its end,
// and G is consumed in the step that loads it.  With 256 threads a 64x64 tile is 8 registers per thread, so TWO tiles in
// flight (loaded two steps ahead, stored to LDS one step ahead) cost the same 32 registers as one tile did with 128
// threads; LDS stays two slots.  G is prefetched per half: as soon as a 32-key half has consumed its 16 values, the same
// registers are reloaded with the next step's values.  Every load is now issued about one step before it is needed.
// ---------------------------------------------------------------------------------------------
template <bool HAS_G>
__global__ __launch_bounds__(256, 2) void attn_dq4_bf16_kernel(AttnGeomB g, const bf16_t* __restrict__ q,
                                                            const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                            const bf16_t* __restrict__ d_o,
                                                            const float* __restrict__ lse2, float* __restrict__ delta,
                                                            const float* __restrict__ gm, int64_t gm_sb, int64_t gm_st,
                                                            bf16_t* __restrict__ dq) {
    __shared__ __attribute__((aligned(16))) bf16_t kt[2][64 * BP];
    __shared__ __attribute__((aligned(16))) bf16_t vt[2][64 * BP];
    const int nqt = (g.T + 127) >> 7;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int qt = id % nqt; id /= nqt;
    const int h = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int q0 = qt * 128 + wave * 32;
    const int64_t base = (int64_t)b * g.sb + (int64_t)h * g.sh;
    const int64_t obase = (int64_t)b * g.osb + (int64_t)h * g.osh;
    const float c = g.scale * ACR_LOG2E;
    bf16x8 qreg[4], doreg[4];
    load_rows_bf(qreg, q + base, g.st, q0, g.T, lane);
    load_rows_bf(doreg, d_o + obase, g.ost, q0, g.T, lane);
    const bool qok = q0 + r < g.T;
    const int qc = min(q0 + r, g.T - 1);
    const float l2v = lse2[((int64_t)b * g.H + h) * g.T + qc];
    const float dlv = delta[((int64_t)b * g.H + h) * g.T + qc];
    const float l2 = qok ? l2v : INFINITY;                 // rows beyond T: p = exp2(-inf) = 0
    const float dl = qok ? dlv : 0.f;
    const float invH = 1.f / (float)g.H;
    const float* grow = HAS_G ? gm + (int64_t)b * gm_sb + (int64_t)qc * gm_st + 4 * hh : nullptr;
    const int gmax = (int)gm_st - 4 - 4 * hh;              // last in-row 16-byte group start (relative to grow)
    // two tiles in flight: set (t & 1) holds key tile t between its global load (step t-2) and its LDS store (step t-1)
    TileRegs<256> kr[2], vr[2];
    tile_gload<256>(kr[0], k + base, g.st, 0, g.T, tid);
    tile_gload<256>(vr[0], v + base, g.st, 0, g.T, tid);
    tile_gload<256>(kr[1], k + base, g.st, 64, g.T, tid);
    tile_gload<256>(vr[1], v + base, g.st, 64, g.T, tid);
    f32x4 gq[2][4];
    if (HAS_G) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int grp = 0; grp < 4; ++grp)
                __builtin_memcpy(&gq[kb][grp], grow + min(32 * kb + 8 * grp, gmax), 16);
    }
    ACR_MEMBAR();
    tile_lstore<256>(kt[0], kr[0], 0, g.T, tid);
    tile_lstore<256>(vt[0], vr[0], 0, g.T, tid);
    __syncthreads();
    f32x16 dq0 = {0}, dq1 = {0};
    f32x16 y0 = {0}, y1 = {0};                              // HAS_G only
    float rho = 0.f;
    // step t (keys k0 = 64 t .. +63) reads LDS slot t & 1; PAR = t & 1.  EDGE steps carry the clamps and key masks: the
    // step's own tile, the tile stored at its end (t+1) or the tile loaded at its top (t+2) is partial or out of range.
    auto step = [&](int k0, auto edge_tag, auto par_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        constexpr int PAR = decltype(par_tag)::value;
        tile_gload<256, EDGE>(kr[PAR], k + base, g.st, k0 + 128, g.T, tid);        // tile t+2
        tile_gload<256, EDGE>(vr[PAR], v + base, g.st, k0 + 128, g.T, tid);
        ACR_MEMBAR();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const bf16_t* ktc = kt[PAR] + kb * 32 * BP;
            const bf16_t* vtc = vt[PAR] + kb * 32 * BP;
            f32x16 s = {0}, dp = {0};
            mma_rowop_bf(s, ktc, qreg, lane);               // S^T raw [key = krow][query = r]
            mma_rowop_bf(dp, vtc, doreg, lane);             // dP^T
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                float p = fast_exp2(fmaf(s[reg], c, -l2));
                float gv = HAS_G ? gq[kb][reg >> 2][reg & 3] : 0.f;
                float t = dp[reg];
                if (EDGE) {                                 // pad columns of G may hold anything: select, don't multiply
                    const bool kv = k0 + 32 * kb + acr_krow(reg, hh) < g.T;
                    p = kv ? p : 0.f;
                    t = kv ? t : 0.f;
                    gv = kv ? gv : 0.f;
                }
                if (HAS_G) {
                    t = fmaf(gv, invH, t);
                    rho = fmaf(p, gv, rho);
                    dp[reg] = p;
                }
                s[reg] = p * (t - dl);
            }
            if (HAS_G) {                                    // this half's G registers are free: fetch the next step's
                ACR_MEMBAR();
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    const int go = k0 + 64 + 32 * kb + 8 * grp;
                    __builtin_memcpy(&gq[kb][grp], grow + (EDGE ? min(go, gmax) : go), 16);
                }
                ACR_MEMBAR();
            }
            mma_accop_a_bf(dq0, s, ktc, 0, lane);           // dQ[query = krow][d = 32*blk + r]
            mma_accop_a_bf(dq1, s, ktc, 1, lane);
            if (HAS_G) {
                mma_accop_a_bf(y0, dp, ktc, 0, lane);       // Y[query = krow][d]
                mma_accop_a_bf(y1, dp, ktc, 1, lane);
            }
        }
        ACR_MEMBAR();
        tile_lstore<256>(kt[PAR ^ 1], kr[PAR ^ 1], k0 + 64, g.T, tid);             // tile t+1, loaded one step ago
        tile_lstore<256>(vt[PAR ^ 1], vr[PAR ^ 1], k0 + 64, g.T, tid);
        __syncthreads();
    };
    {
        int k0 = 0;
        for (; k0 + 256 <= g.T; k0 += 128) {                // tiles t, t+1 and the tiles they load (t+2, t+3) all full
            step(k0, std::false_type{}, std::integral_constant<int, 0>{});
            step(k0 + 64, std::false_type{}, std::integral_constant<int, 1>{});
        }
        for (; k0 < g.T; k0 += 128) {
            step(k0, std::true_type{}, std::integral_constant<int, 0>{});
            if (k0 + 64 < g.T) step(k0 + 64, std::true_type{}, std::integral_constant<int, 1>{});
        }
    }
    if (HAS_G) {
        rho += __shfl_xor(rho, 32);                         // both halves of a lane pair hold keys of the same query row
        rho *= invH;                                        // lane r (either half): rho_{q0+r} / H
        if (hh == 0 && qok) delta[((int64_t)b * g.H + h) * g.T + q0 + r] = dl + rho;      // full delta for dK/dV
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int qq = q0 + acr_krow(reg, hh);
        if (HAS_G) {                                        // accumulator rows are krow(reg, hh): fetch that row's rho
            const float rr = __shfl(rho, acr_krow(reg, hh));
            dq0[reg] = fmaf(-rr, y0[reg], dq0[reg]);
            dq1[reg] = fmaf(-rr, y1[reg], dq1[reg]);
        }
        if (qq < g.T) {
            bf16_t* p = dq + base + (int64_t)qq * g.st;
            p[r] = (bf16_t)(dq0[reg] * g.scale);
            p[32 + r] = (bf16_t)(dq1[reg] * g.scale);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dK, dV: 2 waves x 32 keys (K, V fragments in registers), 64 queries per step (Q/dO tiles double-buffered)
// ---------------------------------------------------------------------------------------------
template <bool HAS_G>
__global__ __launch_bounds__(128, 2) void attn_dkdv_bf16_kernel(AttnGeomB g, const bf16_t* __restrict__ q,
                                                             const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                             const bf16_t* __restrict__ d_o,
                                                             const float* __restrict__ lse2,
                                                             const float* __restrict__ delta,
                                                             const float* __restrict__ gm, int64_t gm_sb, int64_t gm_st,
                                                             bf16_t* __restrict__ dk, bf16_t* __restrict__ dv) {
    __shared__ __attribute__((aligned(16))) bf16_t qtile[2][64 * BP];
    __shared__ __attribute__((aligned(16))) bf16_t dotile[2][64 * BP];
    __shared__ float l2s[2][64], dls[2][64];
    const int nkt = (g.T + 63) >> 6;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int ktile = id % nkt; id /= nkt;
    const int h = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int key0 = ktile * 64 + wave * 32;
    const int64_t base = (int64_t)b * g.sb + (int64_t)h * g.sh;
    const int64_t obase = (int64_t)b * g.osb + (int64_t)h * g.osh;
    const float c = g.scale * ACR_LOG2E;
    bf16x8 kreg[4], vreg[4];
    load_rows_bf(kreg, k + base, g.st, key0, g.T, lane);
    load_rows_bf(vreg, v + base, g.st, key0, g.T, lane);
    const int key = key0 + r;
    const float kmask = (key < g.T) ? 1.f : 0.f;
    const float invH = 1.f / (float)g.H;
    const float* lrow = lse2 + ((int64_t)b * g.H + h) * g.T;
    const float* drow = delta + ((int64_t)b * g.H + h) * g.T;
    const float* gbase = HAS_G ? gm + (int64_t)b * gm_sb + min(key, g.T - 1) : nullptr;
    const int gst = (int)gm_st;
    const unsigned gloff = 4u * (unsigned)(min(key, g.T - 1) + 4 * hh * gst);  // lane part of a G address (bytes, 32 bit)
    TileRegs<128> qr, dr;
    float lnext = 0.f, dnext = 0.f;
    tile_gload<128>(qr, q + base, g.st, 0, g.T, tid);
    tile_gload<128>(dr, d_o + obase, g.ost, 0, g.T, tid);
    if (tid < 64) { lnext = lrow[min(tid, g.T - 1)]; dnext = drow[min(tid, g.T - 1)]; }
    ACR_MEMBAR();
    tile_lstore<128>(qtile[0], qr, 0, g.T, tid);
    tile_lstore<128>(dotile[0], dr, 0, g.T, tid);
    if (tid < 64) { l2s[0][tid] = (tid < g.T) ? lnext : INFINITY; dls[0][tid] = (tid < g.T) ? dnext : 0.f; }
    __syncthreads();
    f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
    int cur = 0;
    // Register diet (this kernel sat at 318 VGPR+AGPR = 1 wave/SIMD): the next Q tile and the G values of the first
    // 32 queries are fetched before the first half's MFMAs, the next dO tile and the second half's G values only
    // before the second half.  EDGE steps (partial last query tile / the step prefetching it) carry clamps and masks.
    auto step = [&](int q0, auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (qb == 0) {
                tile_gload<128, EDGE>(qr, q + base, g.st, q0 + 64, g.T, tid);
                if (tid < 64) {
                    const int qn = EDGE ? min(q0 + 64 + tid, g.T - 1) : q0 + 64 + tid;
                    lnext = lrow[qn]; dnext = drow[qn];
                }
            } else {
                tile_gload<128, EDGE>(dr, d_o + obase, g.ost, q0 + 64, g.T, tid);
            }
            float gv[16];
            if (HAS_G) {
                if (EDGE) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int qq = q0 + 32 * qb + acr_krow(reg, hh);
                        gv[reg] = gbase[min(qq, g.T - 1) * gst];
                    }
                } else {
                    // uniform row base (SGPR pair) + one per-lane 32-bit offset: sixteen saddr-form loads without any
                    // per-load 64-bit address arithmetic (the per-lane-pointer form cost ~3 VALU per load)
                    const float* gu = gm + (int64_t)b * gm_sb + (int64_t)(q0 + 32 * qb) * gst;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        gv[reg] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(gu + ((reg & 3) + 8 * (reg >> 2)) * gst) + gloff);
                }
            }
            ACR_MEMBAR();
            const bf16_t* qtc = qtile[cur] + qb * 32 * BP;
            const bf16_t* dtc = dotile[cur] + qb * 32 * BP;
            f32x16 s = {0}, dp = {0};
            mma_rowop_bf(s, qtc, kreg, lane);               // S raw [query = krow][key = r]
            mma_rowop_bf(dp, dtc, vreg, lane);              // dP
            // in place: s becomes P and dp becomes dS (see the dQ sweep: fresh element-wise vectors cost 16 v_mov each)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int kr = 32 * qb + acr_krow(reg, hh);
                const float pv = fast_exp2(fmaf(s[reg], c, -l2s[cur][kr])) * kmask;   // lse = +inf beyond T -> 0
                s[reg] = pv;
                const float t = HAS_G ? fmaf(gv[reg], invH, dp[reg]) : dp[reg];
                dp[reg] = pv * (t - dls[cur][kr]);
            }
            mma_accop_a_bf(dv0, s, dtc, 0, lane);           // dV[key = krow][d = 32*blk + r]
            mma_accop_a_bf(dv1, s, dtc, 1, lane);
            mma_accop_a_bf(dk0, dp, qtc, 0, lane);
            mma_accop_a_bf(dk1, dp, qtc, 1, lane);
            if (qb == 0) {                                   // Q tile of the next step can go to LDS already
                ACR_MEMBAR();
                tile_lstore<128>(qtile[cur ^ 1], qr, q0 + 64, g.T, tid);
            }
        }
        ACR_MEMBAR();
        tile_lstore<128>(dotile[cur ^ 1], dr, q0 + 64, g.T, tid);
        if (tid < 64) {
            const bool ok = !EDGE || q0 + 64 + tid < g.T;
            l2s[cur ^ 1][tid] = ok ? lnext : INFINITY;
            dls[cur ^ 1][tid] = ok ? dnext : 0.f;
        }
        __syncthreads();
        cur ^= 1;
    };
    {
        int q0 = 0;
        for (; q0 + 128 <= g.T; q0 += 64) step(q0, std::false_type{});
        for (; q0 < g.T; q0 += 64) step(q0, std::true_type{});
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int kk = key0 + acr_krow(reg, hh);
        if (kk < g.T) {
            bf16_t* pk = dk + base + (int64_t)kk * g.st;
            bf16_t* pv = dv + base + (int64_t)kk * g.st;
            pk[r] = (bf16_t)(dk0[reg] * g.scale);
            pk[32 + r] = (bf16_t)(dk1[reg] * g.scale);
            pv[r] = (bf16_t)dv0[reg];
            pv[32 + r] = (bf16_t)dv1[reg];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers (called from the dtype dispatch in attn_f32.hip)
// ---------------------------------------------------------------------------------------------
static AttnGeomB geom_b(const acr_attn_desc* d) {
    AttnGeomB g;
    g.B = d->B; g.H = d->H; g.T = d->T; g.scale = d->scale;
    g.sb = d->qkv_sb; g.st = d->qkv_st; g.sh = d->qkv_sh;
    g.osb = d->o_sb; g.ost = d->o_st; g.osh = d->o_sh;
    return g;
}

bool acr_bf16_mfma_ok(const acr_attn_desc* d, const void* const* ptrs, int n) {
    if ((d->qkv_sb | d->qkv_st | d->qkv_sh | d->o_sb | d->o_st | d->o_sh) & 7) return false;
    for (int i = 0; i < n; ++i)
        if (reinterpret_cast<uintptr_t>(ptrs[i]) & 15) return false;
    return true;
}

void acr_attn_fwd_bf16(const acr_attn_desc* d, const void* q, const void* k, const void* v, void* o, float* lse2,
                       float* pmean, int64_t pmean_sb, int64_t pmean_st, hipStream_t st) {
    AttnGeomB g = geom_b(d);
    const int nqt = (d->T + 63) / 64;
    hipLaunchKernelGGL(attn_fwd_bf16_kernel, dim3(d->B * d->H * nqt), dim3(128), 0, st, g, (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, lse2);
    if (pmean)
        hipLaunchKernelGGL((attn_tile_qk_bf16_kernel<TQKB_PMEAN>), dim3(d->B * nqt * nqt), dim3(256), 0, st, g,
                           (const bf16_t*)q, (const bf16_t*)k, (const float*)lse2, pmean, pmean_sb, pmean_st);
}

void acr_attn_bwd_bf16(const acr_attn_desc* d, const void* q, const void* k, const void* v, const void* o,
                       const void* d_o, const float* lse2, const float* gm, int64_t gm_sb, int64_t gm_st, void* dq, void* dk,
                       void* dv, float* delta, hipStream_t st) {
    AttnGeomB g = geom_b(d);
    const int nqt = (d->T + 63) / 64;
    const dim3 grid(d->B * d->H * nqt);
    // delta kernel: D_i = rowsum(dO o O) only; with G the dQ sweep adds rho_i/H and rewrites delta before dK/dV reads it
    hipLaunchKernelGGL((attn_delta_bf16_kernel<false>), grid, dim3(128), 0, st, g, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)o, (const bf16_t*)d_o, lse2, (const float*)nullptr, (int64_t)0, (int64_t)0, delta);
    // 2: 2-wave sweep, 4: 4-wave sweep.  Measured at B=32, H=12, T=785: with G both take 217 us (the sweep is bound by its
    // MFMA -> softmax -> MFMA dependency chain at 2 waves/SIMD, not by load latency); without G (CAM inference, plain
    // attention) the 4-wave sweep is 9 % faster (127 vs 139 us) and uses 168 registers.  Default: by HAS_G.
    const int dq_env = acr_opt(ACR_OPT_DQ_VARIANT);
    const int dq_variant = dq_env ? dq_env : (gm ? 2 : 4);
    const dim3 grid4(d->B * d->H * ((d->T + 127) / 128));
#define ACR_BWD_LAUNCH(HG)                                                                                          \
    if (dq_variant == 4)                                                                                             \
        hipLaunchKernelGGL((attn_dq4_bf16_kernel<HG>), grid4, dim3(256), 0, st, g, (const bf16_t*)q, (const bf16_t*)k, \
                           (const bf16_t*)v, (const bf16_t*)d_o, lse2, delta, gm, gm_sb, gm_st, (bf16_t*)dq);         \
    else                                                                                                             \
        hipLaunchKernelGGL((attn_dq_bf16_kernel<HG>), grid, dim3(128), 0, st, g, (const bf16_t*)q, (const bf16_t*)k, \
                           (const bf16_t*)v, (const bf16_t*)d_o, lse2, delta, gm, gm_sb, gm_st, (bf16_t*)dq);         \
    hipLaunchKernelGGL((attn_dkdv_bf16_kernel<HG>), grid, dim3(128), 0, st, g, (const bf16_t*)q, (const bf16_t*)k,   \
                       (const bf16_t*)v, (const bf16_t*)d_o, lse2, (const float*)delta, gm, gm_sb, gm_st, (bf16_t*)dk, \
                       (bf16_t*)dv);
    if (gm) { ACR_BWD_LAUNCH(true) } else { ACR_BWD_LAUNCH(false) }
#undef ACR_BWD_LAUNCH
}

void acr_attn_probs_bf16(const acr_attn_desc* d, const void* q, const void* k, const float* lse2, float* probs,
                         hipStream_t st) {
    AttnGeomB g = geom_b(d);
    const int nt = (d->T + 63) / 64;
    hipLaunchKernelGGL((attn_tile_qk_bf16_kernel<TQKB_PROBS>), dim3(d->B * d->H * nt * nt), dim3(256), 0, st, g,
                       (const bf16_t*)q, (const bf16_t*)k, lse2, probs, (int64_t)0, (int64_t)0);
}

void acr_attn_dprobs_bf16(const acr_attn_desc* d, const void* d_o, const void* v, float* dprobs, hipStream_t st) {
    AttnGeomB g = geom_b(d);
    const int nt = (d->T + 63) / 64;
    hipLaunchKernelGGL((attn_tile_qk_bf16_kernel<TQKB_DPROBS>), dim3(d->B * d->H * nt * nt), dim3(256), 0, st, g,
                       (const bf16_t*)d_o, (const bf16_t*)v, (const float*)nullptr, dprobs, (int64_t)0, (int64_t)0);
}
