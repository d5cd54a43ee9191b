// fp32 (reference precision) attention, third generation: RESIDENT SCORES.
//
// Why: on gfx950 the exact-fp32 MFMA runs at the fp32 vector rate (157 TF, 1/16 of bf16) while HBM3E moves 6+ TB/s, so the
// ridge of the fp32 roofline is ~20 FLOP/byte -- and recomputing one logit S[i][j] = q_i . k_j costs 128 FLOP for 4 bytes
// (32 FLOP/byte).  The recompute generation (attn_f32_dma.hip) executed 3 products forward (S, PV, S again for the head
// mean) and 8 backward (S for the row term, S/dP/dQ, S/dP/dV/dK) for 2 + 4 algorithmic ones.  With 288 GB of HBM per GPU the
// scaled logits of every layer fit many times over (983 MB per layer at B = 32 views, H = 12, T = 785; 11.8 GB per step),
// so here the forward sweep stores them ONCE, in the MFMA accumulator layout it produces them in, and everything after it
// streams them back instead of recomputing:
//     forward   2 products  + 1 write of S          head-mean: one read of S (no MFMA)
//     backward  5 products  (dP dQ | dP dV dK)       row term : one read of S and G (no MFMA), dQ / dK-dV: one read each
// Numerics are those of the recompute generation bit for bit where the same expression is evaluated (S is the forward's own
// k-ordered fp32 fmaf chain, P = exp2(S - lse2)); the head mean now averages exactly the P the forward used.
//
// Score layout ("scores", caller-owned, acr_attn_scores_floats(desc) floats): blocks of 32 keys x 32 queries,
//     block(b, h, qb, kb) at ((((b*H + h)*NB + qb)*NB + kb) * 1024 floats,   NB = ceil(T / 32)
// inside a block the 32x32 accumulator tile of v_mfma_f32_32x32x2_f32 as the forward holds it: lane (theta = query & 31, hh),
// register reg <-> key kappa = (reg & 3) + 8 (reg >> 2) + 4 hh;  float offset = (reg >> 2)*256 + lane*4 + (reg & 3), so that one
// global_store_dwordx4 per register quad writes 1 KB contiguously and the same-orientation readers (head mean, row term, dQ)
// get their 16 registers back with four 16-byte loads.  The dK/dV sweep needs the transposed orientation (key on the lane):
// it pulls its block by LDS-DMA with a chunk permutation on the source address, chunk ^ (2*quad + (chunk >> 5)), that makes
// the transposed ds_read_b32 walk bank-conflict free.  Keys >= T hold -inf (the forward masks before it stores), queries
// >= T hold finite junk and every reader gives those rows lse2 = +inf, i.e. P = 0.
#include <type_traits>

#include "acr_common.h"
#include "attn_f32.h"
#include "attn_f32_tiles.h"
#include "attn_f32_sres_tails.h"


// ---------------------------------------------------------------------------------------------
// forward: workgroup = (b, h, 128 queries), wave = 32 queries; K/V tiles of 32 keys stream through the two-slot LDS ring
// (attn_fwd_dma_kernel's loop) and every 32 x 32 logit tile is written to `sres` before the softmax consumes it
// ---------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_fwd_sres_kernel(AttnGeom g, const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, float* __restrict__ o,
                                                               float* __restrict__ lse2, float* __restrict__ sres, int ntail) {
    __shared__ __attribute__((aligned(1024))) float smem[4 * DT_FLOATS];       // [slot][K | V]
    __shared__ float mlsh[4 * 64];                                             // split tail: (m | l) of the four partial sweeps
    // the last `ntail` workgroups of the grid are split-tail workgroups, one per (b, h) (dispatched after every full one)
    const int nmain = (int)gridDim.x - ntail;
    if (NW == 4 && (int)blockIdx.x >= nmain) {
        attn_fwd_tail_body<4>(smem, mlsh, g, q, k, v, o, lse2, sres, acr_xcd_remap((int)blockIdx.x - nmain, ntail));
        return;
    }
    const int NB = (g.T + 31) >> 5, nqt = ntail ? NB / NW : (NB + NW - 1) / NW;
    int id = acr_xcd_remap(blockIdx.x, nmain);
    const int qt = id % nqt; id /= nqt;
    const int hd = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = (qt * NW + wave) * 32;
    const bool live = q0 < g.T;                            // wave-uniform: waves past the end only help with the DMA
    const int64_t base = (int64_t)b * g.sb + (int64_t)hd * g.sh;
    const float* kb = k + base;
    const float* vb = v + base;
    dma_tile32_nw<NW>(smem, kb, g.st, 0, g.T, wave, lane);
    dma_tile32_nw<NW>(smem + DT_FLOATS, vb, g.st, 0, g.T, wave, lane);
    float qreg[32];
    rows_from_global(qreg, q + base, g.st, q0, g.T, r, h, g.scale * ACR_LOG2E);
    float m = -INFINITY, l = 0.f;
    f32x16 o0 = {0}, o1 = {0};
    const LaneBasesA lb = lane_bases_a(r, h, smem);
    int doff[(8 + NW - 1) / NW];
    dma_offsets32_nw<NW>(doff, g.st, wave, lane);
    float* sblk = sres + sres_block(g, NB, b, hd, min(q0 >> 5, NB - 1), 0) + lane * 4;
    auto step = [&](int k0, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int KOFF = SLOT * 2 * DT_FLOATS * 4, VOFF = KOFF + DT_FLOATS * 4;
        acr_dma_barrier();                                 // slot SLOT has landed; the other slot is free
        __builtin_amdgcn_s_setprio(2);
        if (k0 + 64 <= g.T) {                              // next tile fully inside: precomputed lane offsets, uniform base
            dma_tile32_nw_i<NW>(smem + (SLOT ^ 1) * 2 * DT_FLOATS, kb + (int64_t)(k0 + 32) * g.st, doff, wave);
            dma_tile32_nw_i<NW>(smem + (SLOT ^ 1) * 2 * DT_FLOATS + DT_FLOATS, vb + (int64_t)(k0 + 32) * g.st, doff, wave);
        } else if (k0 + 32 < g.T) {                        // partial last tile: clamped rows
            dma_tile32_nw<NW>(smem + (SLOT ^ 1) * 2 * DT_FLOATS, kb, g.st, k0 + 32, g.T, wave, lane);
            dma_tile32_nw<NW>(smem + (SLOT ^ 1) * 2 * DT_FLOATS + DT_FLOATS, vb, g.st, k0 + 32, g.T, wave, lane);
        }
        __builtin_amdgcn_s_setprio(0);
        if (!live) return;
        f32x16 s = {0};
        rowop_x<KOFF>(s, lb, qreg);                        // s[reg] = S2[key = k0 + krow][query = q0 + r]
        __builtin_amdgcn_s_setprio(2);
        if (k0 + 32 > g.T) {                               // only the last key tile has keys beyond T (uniform branch)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (k0 + acr_krow(reg, h) >= g.T) s[reg] = -INFINITY;
        }
        float* sp = sblk + (int64_t)(k0 >> 5) * SB_FLOATS;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 t = {s[4 * gq], s[4 * gq + 1], s[4 * gq + 2], s[4 * gq + 3]};
            SRES_STORE(sp + gq * 256, t);
        }
        float mx = s[0];
#pragma unroll
        for (int reg = 1; reg < 16; ++reg) mx = fmaxf(mx, s[reg]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (__any(mx > m + 8.f)) {                         // deferred rescale (see attn_fwd_dma_kernel)
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
        __builtin_amdgcn_s_setprio(0);
        accop_x<VOFF, 0, false>(o0, p, lb);                // o[reg] = O^T[d = 32*blk + krow][query = r]
        accop_x<VOFF, 1, false>(o1, p, lb);
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

// ---------------------------------------------------------------------------------------------
// head mean of P (DPT/ACR.py:107-112) from the resident scores: HBM-bound stream, no MFMA.  One wave per (b, qb, kb) block:
// for every head four 16-byte loads, P = exp2(S - lse2), summed in head order in registers (deterministic), transposed
// through a private LDS tile so that the (T, T) map is written in 128-byte row segments.
// ---------------------------------------------------------------------------------------------
// The same sums with the workgroup = NW = 4 heads of one (sample, query block) (H % 4 == 0): the gradient block
// gm[b][32 queries][32 keys] of a step is the same for every head, and read the way the kernel above reads it -- a float4 per lane
// from 32 different rows per instruction -- it costs the texture path 32 cache lines per wave instruction against 8 for the score
// block (measured: 3.2 TB/s against the 5 TB/s of the head-mean pass over the same scores).  Here the 256 threads fetch the block
// once per workgroup as 32 row segments of 128 bytes (8 lines per instruction), park it in LDS (row stride 36 floats: the
// accumulator-order float4 reads of 16 lanes then touch every bank once) and all four waves read it from there; double-buffered,
// one fence-free barrier per key block, the next block's scores and gm piece in flight meanwhile.  Same operations in the same order
// as the kernel above: bit-identical delta.
#ifndef DELTA4_MINB
#define DELTA4_MINB 1
#endif
template <int NW> __global__ __launch_bounds__(64 * NW, DELTA4_MINB) void attn_delta_sres4_kernel(AttnGeom g, int NB, const float* __restrict__ sres,
                                                               const float* __restrict__ o, const float* __restrict__ d_o,
                                                               const float* __restrict__ lse2, const float* __restrict__ gm,
                                                               int64_t gm_sb, int64_t gm_st, float* __restrict__ delta) {
    __shared__ __attribute__((aligned(16))) float gt[3][32 * 36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int theta = lane & 31, hh = lane >> 5;
    const int HG = g.H / NW;
    const int hd = (blockIdx.x % HG) * NW + wave;
    const int t = blockIdx.x / HG;
    const int qb = t % NB, b = t / NB;
    const int qrow = qb * 32 + theta;
    const bool qok = qrow < g.T;
    const int qc = min(qrow, g.T - 1);
    float part = 0.f;
    {
        const int64_t off = (int64_t)b * g.osb + (int64_t)hd * g.osh + (int64_t)qc * g.ost + 32 * hh;
        const float* op = o + off;
        const float* dp = d_o + off;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(op + 4 * i), c = *reinterpret_cast<const f32x4*>(dp + 4 * i);
            part += a[0] * c[0] + a[1] * c[1] + a[2] * c[2] + a[3] * c[3];
        }
    }
    part += __shfl_xor(part, 32);
    const float l2q = qok ? lse2[((int64_t)b * g.H + hd) * g.T + qrow] : INFINITY;
    const float* sp = sres + sres_block(g, NB, b, hd, qb, 0) + lane * 4;
    const bool loader = tid < 256;                         // the first four waves fetch the gm blocks
    const int lrow = (tid & 255) >> 3, lc = (tid & 7) * 4;  // this thread's float4 of a gm block: row lrow, columns lc ..
    const float* gl = gm + (int64_t)b * gm_sb + (int64_t)min(qb * 32 + lrow, g.T - 1) * gm_st + lc;
    const int nfull = g.T >> 5;                            // key blocks entirely inside [0, T)
    float rho = 0.f;
    // Three register sets / three LDS tiles in rotation: the loads of key block kb + 2 are issued before block kb is summed, so two
    // blocks (8 KB per wave) are in flight across every barrier.  With one block ahead -- loads issued, the previous block summed
    // (300 cycles), loads waited for, barrier -- a step took the slowest of the four waves' memory round trips: 3.8 TB/s.
    f32x4 sv[3][4], gn[3];
    auto fetch = [&](int kb, int set) {
        if (loader) gn[set] = *reinterpret_cast<const f32x4*>(gl + kb * 32);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) sv[set][gq] = SRES_LOAD_NT(sp + (int64_t)kb * SB_FLOATS + gq * 256);
    };
    if (nfull > 0) {
        fetch(0, 0);
        if (nfull > 1) fetch(1, 1);
        if (loader) *reinterpret_cast<f32x4*>(&gt[0][lrow * 36 + lc]) = gn[0];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acr_barrier_nofence();
    }
    for (int kb0 = 0; kb0 < nfull; kb0 += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int kb = kb0 + u;
            if (kb < nfull) {                               // workgroup-uniform
                constexpr int n1[3] = {1, 2, 0}, n2[3] = {2, 0, 1};
                if (kb + 2 < nfull) fetch(kb + 2, n2[u]);
                const float* gb = &gt[u][theta * 36 + 4 * hh];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(gb + 8 * gq);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rho = fmaf(__builtin_amdgcn_exp2f(sv[u][gq][e] - l2q), gv[e], rho);
                }
                if (kb + 1 < nfull && loader) *reinterpret_cast<f32x4*>(&gt[n1[u]][lrow * 36 + lc]) = gn[n1[u]];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of tile u and its writes of the next one
                acr_barrier_nofence();
            }
        }
    }
    if (nfull < NB) {                                       // the partial last key block: as the kernel above (scores there are -inf past T)
        const int k0 = nfull * 32;
        const float* gr = gm + (int64_t)b * gm_sb + (int64_t)qc * gm_st;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(sp + (int64_t)nfull * SB_FLOATS + gq * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gv = gr[min(k0 + 8 * gq + 4 * hh + e, g.T - 1)];
                rho = fmaf(__builtin_amdgcn_exp2f(s4[e] - l2q), gv, rho);
            }
        }
    }
    rho += __shfl_xor(rho, 32);
    if (hh == 0 && qok) delta[((int64_t)b * g.H + hd) * g.T + qrow] = part + rho * (1.f / (float)g.H);
}

__global__ __launch_bounds__(256) void attn_pmean_sres_kernel(AttnGeom g, int NB, const float* __restrict__ sres,
                                                              const float* __restrict__ lse2, float* __restrict__ out,
                                                              int64_t out_sb, int64_t out_st) {
    __shared__ float tile[4][32 * 33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int theta = lane & 31, hh = lane >> 5;
    const int nblk = g.B * NB * NB;
    int id = blockIdx.x * 4 + wave;
    const bool act = id < nblk;
    id = min(id, nblk - 1);
    const int kb = id % NB;
    const int t = id / NB;
    const int qb = t % NB, b = t / NB;
    const int qrow = qb * 32 + theta;
    const float* sp = sres + sres_block(g, NB, b, 0, qb, kb) + lane * 4;
    const int64_t hstride = (int64_t)NB * NB * SB_FLOATS;
    const float* lp = lse2 + (int64_t)b * g.H * g.T + min(qrow, g.T - 1);
    f32x16 pm = {0};
    for (int h0 = 0; h0 < g.H; h0 += 4) {                  // four heads (16 KB per wave) of loads in flight
        f32x4 sv[4][4];
        float lv[4];
#pragma unroll
        for (int hi = 0; hi < 4; ++hi) {
            const int hd = min(h0 + hi, g.H - 1);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) sv[hi][gq] = SRES_LOAD_NT(sp + hd * hstride + gq * 256);
            lv[hi] = lp[(int64_t)hd * g.T];
        }
#pragma unroll
        for (int hi = 0; hi < 4; ++hi) {
            if (h0 + hi < g.H) {
                const float l2 = qrow < g.T ? lv[hi] : INFINITY;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) pm[reg] += __builtin_amdgcn_exp2f(sv[hi][reg >> 2][reg & 3] - l2);
            }
        }
    }
    const float mul = 1.f / (float)g.H;
    float* tl = tile[wave];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) tl[theta * 33 + acr_krow(reg, hh)] = pm[reg] * mul;
    __syncthreads();
    if (act) {
        const int key = kb * 32 + theta;
        float* ob = out + (int64_t)b * out_sb;
        if (key < g.T) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = 2 * i + hh;
                const int qq = qb * 32 + row;
                if (qq < g.T) ob[(int64_t)qq * out_st + key] = tl[row * 33 + theta];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// delta[b,h,i] = rowsum(dO*O) + (1/H) sum_j P_h[i,j] G[b,i,j] from the resident scores: HBM/L2-bound stream, no MFMA.  One
// wave per (b, qb, h) (h fastest: the twelve heads that read the same 32 rows of G run next to each other); lane (theta, hh)
// sums its 16 keys of every block in register order, the two lane halves are added at the end (deterministic).
// G rows are read in 16-byte groups (pitch % 4 == 0, checked by the launcher); the partial last key block takes clamped
// scalar loads (its masked keys have P = 0 exactly).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_sres_kernel(AttnGeom g, int NB, const float* __restrict__ sres,
                                                              const float* __restrict__ o, const float* __restrict__ d_o,
                                                              const float* __restrict__ lse2, const float* __restrict__ gm,
                                                              int64_t gm_sb, int64_t gm_st, float* __restrict__ delta) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int theta = lane & 31, hh = lane >> 5;
    int id = blockIdx.x * 4 + wave;
    if (id >= g.B * NB * g.H) return;                      // no barriers below
    const int hd = id % g.H;
    const int t = id / g.H;
    const int qb = t % NB, b = t / NB;
    const int qrow = qb * 32 + theta;
    const bool qok = qrow < g.T;
    const int qc = min(qrow, g.T - 1);
    float part = 0.f;
    {
        const int64_t off = (int64_t)b * g.osb + (int64_t)hd * g.osh + (int64_t)qc * g.ost + 32 * hh;
        const float* op = o + off;
        const float* dp = d_o + off;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(op + 4 * i), c = *reinterpret_cast<const f32x4*>(dp + 4 * i);
            part += a[0] * c[0] + a[1] * c[1] + a[2] * c[2] + a[3] * c[3];
        }
    }
    part += __shfl_xor(part, 32);
    float rho = 0.f;
    if (gm != nullptr) {                                    // uniform over the launch
        const float l2q = qok ? lse2[((int64_t)b * g.H + hd) * g.T + qrow] : INFINITY;
        const float* sp = sres + sres_block(g, NB, b, hd, qb, 0) + lane * 4;
        const float* gr = gm + (int64_t)b * gm_sb + (int64_t)qc * gm_st;
        const int nfull = g.T >> 5;                        // key blocks entirely inside [0, T)
#pragma unroll 4
        for (int kb = 0; kb < nfull; ++kb) {
            f32x4 sv[4], gv[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                sv[gq] = SRES_LOAD(sp + (int64_t)kb * SB_FLOATS + gq * 256);
                gv[gq] = *reinterpret_cast<const f32x4*>(gr + kb * 32 + 8 * gq + 4 * hh);
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                rho = fmaf(__builtin_amdgcn_exp2f(sv[reg >> 2][reg & 3] - l2q), gv[reg >> 2][reg & 3], rho);
        }
        if (nfull < NB) {
            const int k0 = nfull * 32;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 sv = *reinterpret_cast<const f32x4*>(sp + (int64_t)nfull * SB_FLOATS + gq * 256);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gv = gr[min(k0 + 8 * gq + 4 * hh + e, g.T - 1)];
                    rho = fmaf(__builtin_amdgcn_exp2f(sv[e] - l2q), gv, rho);
                }
            }
        }
        rho += __shfl_xor(rho, 32);
    }
    if (hh == 0 && qok) delta[((int64_t)b * g.H + hd) * g.T + qrow] = part + rho * (1.f / (float)g.H);
}

// ---------------------------------------------------------------------------------------------
// Counted waits (round 4).  A step's tile DMA must have landed at the step's barrier, but the private streams of a wave -- its
// score blocks (two steps ahead) and its G block (one step) -- are YOUNGER vector-memory operations and may stay in flight:
// vmcnt retires in issue order, so "at most n outstanding" with n = the number of younger operations is exactly "the tile has
// landed".  n is wave-uniform; every stream is LDS-DMA (register prefetch rings turn into loop-carried copies that hipcc
// waits for right behind the loads -- found in this file's round-3 ISA).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sres_wait_vm(int n) {
    if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#define SRES_FENCE() asm volatile("" ::: "memory")

// ---------------------------------------------------------------------------------------------
// dQ: workgroup = (b, h, 128 queries); dO rows of the wave's 32 queries in registers; K/V tiles stream through the LDS ring;
// the wave's score blocks (as stored: query on the lane) go through a private two-slot LDS ring two steps ahead, its 32 x 32
// block of G through a private single-slot tile one step ahead, both by LDS-DMA.
//   dP^T = V dO^T (32 MFMAs)   dS^T = exp2(S - lse2) (dP^T + G/H - delta)   dQ += dS K (32 MFMAs)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void attn_dq_sres_body(float* smem, float* ssm, float* gsm, int bid, int nblk, bool tails, const AttnGeom& g,
                                                  const float* __restrict__ k, const float* __restrict__ v,
                                                  const float* __restrict__ d_o, const float* __restrict__ lse2,
                                                  const float* __restrict__ delta, const float* __restrict__ sres,
                                                  const float* __restrict__ gm, int64_t gm_sb, int64_t gm_st, float* __restrict__ dq) {
    const int NB = (g.T + 31) >> 5, nqt = tails ? NB >> 2 : (NB + 3) >> 2;      // tails: the last block has its own workgroup
    int id = acr_xcd_remap(bid, nblk);
    const int qt = id % nqt; id /= nqt;
    const int hd = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = qt * 128 + wave * 32;
    const bool live = q0 < g.T;
    const int64_t base = (int64_t)b * g.sb + (int64_t)hd * g.sh;
    const int64_t obase = (int64_t)b * g.osb + (int64_t)hd * g.osh;
    const float* kb = k + base;
    const float* vb = v + base;
    // private streams of this wave: score blocks (qb = q0 / 32, kb = step) copied as stored, G block rows = its queries with
    // 16-byte chunk c of row q in slot c ^ ((q >> 1) & 7) (the lane's row reads are then bank-conflict free)
    const float* srow = sres + sres_block(g, NB, b, hd, min(q0 >> 5, NB - 1), 0) + lane * 4;
    float* sw = ssm + wave * 2 * SB_FLOATS;
    auto dma_scores = [&](int kblk, int slot) {
        const float* src = srow + (int64_t)kblk * SB_FLOATS;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            __builtin_amdgcn_global_load_lds((glb_vp)(src + gq * 256), (lds_vp)(sw + slot * SB_FLOATS + gq * 256), 16, 0, SRES_DMA_AUX);
    };
    const float* gb0 = gm ? gm + (int64_t)b * gm_sb : nullptr;          // uniform
    float* gw = gsm + wave * SB_FLOATS;
    const float* grow[4];
    int gchunk[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = 8 * p + (lane >> 3);
        grow[p] = gb0 ? gb0 + (int64_t)min(q0 + row, g.T - 1) * gm_st : nullptr;
        gchunk[p] = 4 * ((lane & 7) ^ ((row >> 1) & 7));
    }
    auto dma_g = [&](int k0) {
        if (gb0 == nullptr) return;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_global_load_lds((glb_vp)(grow[p] + min(k0 + gchunk[p], (int)gm_st - 4)), (lds_vp)(gw + p * 256), 16, 0, 0);
    };
    dma_tile32(smem, kb, g.st, 0, g.T, wave, lane);
    dma_tile32(smem + DT_FLOATS, vb, g.st, 0, g.T, wave, lane);
    if (live) {
        dma_g(0);
        dma_scores(0, 0);
        if (32 < g.T) dma_scores(1, 1);
    }
    float doreg[32];
    rows_from_global(doreg, d_o + obase, g.ost, q0, g.T, r, h, 1.f);
    const bool qok = q0 + r < g.T;
    const float l2q = qok ? lse2[((int64_t)b * g.H + hd) * g.T + q0 + r] : INFINITY;    // queries beyond T: p = exp2(-inf) = 0
    const float dl = qok ? delta[((int64_t)b * g.H + hd) * g.T + q0 + r] : 0.f;
    const float invH = gb0 ? 1.f / (float)g.H : 0.f;
    f32x16 dq0 = {0}, dq1 = {0};
    const LaneBasesA lb = lane_bases_a(r, h, smem);
    int doff[2];
    dma_offsets32(doff, g.st, wave, lane);
    const uint32_t saddr = lds_addr_of(sw) + lane * 16;                                 // + slot * 4096 + gq * 1024
    uint32_t gaddr[4];                                                                  // quad gq = keys 8 gq + 4 h .. + 3 of row r
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) gaddr[gq] = lds_addr_of(gw) + r * 128 + (((2 * gq + h) ^ ((r >> 1) & 7)) << 4);
    auto step = [&](int k0, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int KOFF = SLOT * 2 * DT_FLOATS * 4, VOFF = KOFF + DT_FLOATS * 4;
        // issue order of a live wave's step:  tile(t+1) [4] | G(t+1) [4]  scores(t+2) [4].  At this barrier tile(t) must have
        // landed; behind it the previous step issued G(t) and scores(t+1).
        sres_wait_vm((k0 > 0 && live) ? (gb0 ? 4 : 0) + (k0 + 32 < g.T ? 4 : 0) : 0);
        acr_barrier_nofence();                             // not __syncthreads(): its release fence drains every DMA (acr_common.h)
        if (k0 + 64 <= g.T) {
            dma_tile32_i(smem + (SLOT ^ 1) * 2 * DT_FLOATS, kb + (int64_t)(k0 + 32) * g.st, doff, wave);
            dma_tile32_i(smem + (SLOT ^ 1) * 2 * DT_FLOATS + DT_FLOATS, vb + (int64_t)(k0 + 32) * g.st, doff, wave);
        } else if (k0 + 32 < g.T) {
            dma_tile32(smem + (SLOT ^ 1) * 2 * DT_FLOATS, kb, g.st, k0 + 32, g.T, wave, lane);
            dma_tile32(smem + (SLOT ^ 1) * 2 * DT_FLOATS + DT_FLOATS, vb, g.st, k0 + 32, g.T, wave, lane);
        }
        SRES_FENCE();
        if (!live) return;
        f32x16 dp = {0};
        rowop_x<VOFF>(dp, lb, doreg);                      // dP^T[key = krow][query = r]
        // G(t) must have landed in this wave's tile: behind it are scores(t+1) [4] and this step's tile(t+1) [4]
        if (k0 > 0) sres_wait_vm(k0 + 32 < g.T ? 8 : 0);
        f32x4 s4[4], g4[4];
        ACR_LDS_RD128(s4[0], saddr, SLOT * 4096); ACR_LDS_RD128(s4[1], saddr, SLOT * 4096 + 1024);
        ACR_LDS_RD128(s4[2], saddr, SLOT * 4096 + 2048); ACR_LDS_RD128(s4[3], saddr, SLOT * 4096 + 3072);
        if (gb0 != nullptr) {
            ACR_LDS_RD128(g4[0], gaddr[0], 0); ACR_LDS_RD128(g4[1], gaddr[1], 0);
            ACR_LDS_RD128(g4[2], gaddr[2], 0); ACR_LDS_RD128(g4[3], gaddr[3], 0);
            ACR_LDS_WAIT4(0, g4[0], g4[1], g4[2], g4[3]);
            if (k0 + 32 > g.T) {                           // keys beyond T: their columns hold whatever the row pitch holds
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k0 + 8 * gq + 4 * h + e >= g.T) g4[gq][e] = 0.f;
            }
        } else {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) g4[gq] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        ACR_LDS_WAIT4(0, s4[0], s4[1], s4[2], s4[3]);
        f32x16 ds;
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            ds[reg] = __builtin_amdgcn_exp2f(s4[reg >> 2][reg & 3] - l2q) * (dp[reg] + g4[reg >> 2][reg & 3] * invH - dl);
        __builtin_amdgcn_s_setprio(0);
        // the private tiles have been read: refill G for the next step and the score slot for the step AFTER next
        if (k0 + 32 < g.T) dma_g(k0 + 32);
        if (k0 + 64 < g.T) dma_scores((k0 >> 5) + 2, SLOT);
        SRES_FENCE();
        accop_x<KOFF, 0, true>(dq0, ds, lb);               // dQ[query = krow][d = 32*blk + r]
        accop_x<KOFF, 1, true>(dq1, ds, lb);
    };
    for (int k0 = 0; k0 < g.T; k0 += 64) {
        step(k0, std::integral_constant<int, 0>{});
        if (k0 + 32 < g.T) step(k0 + 32, std::integral_constant<int, 1>{});
    }
    if (!live) return;
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

// ---------------------------------------------------------------------------------------------
// dK, dV: workgroup = (b, h, 128 keys); V rows of the wave's 32 keys in registers; Q / dO tiles stream through the shared ring;
// every wave streams ITS score blocks (q-block j x its key block) by LDS-DMA into a private two-slot ring two steps ahead and
// reads them transposed (key on the lane), and ITS 32 x 32 block of G into a private single-slot tile (natural [query][key]
// rows: the lane's reads are 32 consecutive floats); lse2 / delta of the step's 32 queries are loaded one per lane at the top
// of the step (consumed behind the 32 MFMAs of dP) and reach the accumulator rows by ds_bpermute.
//   dP = dO V^T (32 MFMAs)   P = exp2(S - lse2)   dS = P (dP + G/H - delta)   dV += P^T dO (32)   dK += dS^T Q (32)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void attn_dkdv_sres_body(float* smem, float* ssm, float* gsm, int bid, int nblk, bool tails, const AttnGeom& g,
                                                    const float* __restrict__ q, const float* __restrict__ v,
                                                    const float* __restrict__ d_o, const float* __restrict__ lse2,
                                                    const float* __restrict__ delta, const float* __restrict__ sres,
                                                    const float* __restrict__ gm, int64_t gm_sb, int64_t gm_st,
                                                    float* __restrict__ dk, float* __restrict__ dv) {
    const int NB = (g.T + 31) >> 5, nkt = tails ? NB >> 2 : (NB + 3) >> 2;
    int id = acr_xcd_remap(bid, nblk);
    const int ktile = id % nkt; id /= nkt;
    const int hd = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int key0 = ktile * 128 + wave * 32;
    const bool live = key0 < g.T;
    const int64_t base = (int64_t)b * g.sb + (int64_t)hd * g.sh;
    const int64_t obase = (int64_t)b * g.osb + (int64_t)hd * g.osh;
    const float* qb = q + base;
    const float* dob = d_o + obase;
    const float* lrow = lse2 + ((int64_t)b * g.H + hd) * g.T;
    const float* drow = delta + ((int64_t)b * g.H + hd) * g.T;
    // score blocks of this wave: (qb = step, kb = key0 / 32); lane c of DMA piece gq fetches global chunk c ^ (2 gq + (c >> 5))
    const float* scol = sres + sres_block(g, NB, b, hd, 0, min(key0 >> 5, NB - 1));
    const int64_t sstep = (int64_t)NB * SB_FLOATS;
    int soff[4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) soff[gq] = gq * 256 + 4 * (lane ^ (2 * gq + (lane >> 5)));
    float* sw = ssm + wave * 2 * SB_FLOATS;
    auto dma_scores = [&](int qblk, int slot) {
        const float* src = scol + (int64_t)qblk * sstep;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            __builtin_amdgcn_global_load_lds((glb_vp)(src + soff[gq]), (lds_vp)(sw + slot * SB_FLOATS + gq * 256), 16, 0, SRES_DMA_AUX);
    };
    // G block of the step: rows = the 32 queries, 128 bytes = this wave's 32 keys; columns clamped into the row (the last key
    // block reaches beyond T: those lanes' P is exactly 0 and whatever they compute never leaves their own accumulator row)
    const float* gb0 = gm ? gm + (int64_t)b * gm_sb : nullptr;          // uniform
    float* gw = gsm + wave * SB_FLOATS;
    const int gcol = min(key0 + 4 * (lane & 7), (int)gm_st - 4);
    auto dma_g = [&](int q0) {
        if (gb0 == nullptr) return;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float* src = gb0 + (int64_t)min(q0 + 8 * p + (lane >> 3), g.T - 1) * gm_st + gcol;
            __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(gw + p * 256), 16, 0, 0);
        }
    };
    dma_tile32(smem, qb, g.st, 0, g.T, wave, lane);
    dma_tile32(smem + DT_FLOATS, dob, g.ost, 0, g.T, wave, lane);
    if (live) {
        dma_g(0);
        dma_scores(0, 0);
        if (32 < g.T) dma_scores(1, 1);
    }
    float vreg[32];
    rows_from_global(vreg, v + base, g.st, key0, g.T, r, h, 1.f);
    const float invH = gb0 ? 1.f / (float)g.H : 0.f;
    f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
    const LaneBasesA lb = lane_bases_a(r, h, smem);
    int qoff[2], dooff[2];                                   // lane parts of the Q / dO tile DMA source addresses (interior tiles)
    dma_offsets32(qoff, g.st, wave, lane);
    dma_offsets32(dooff, g.ost, wave, lane);
    // transposed score reads: lane (kappa = r, h): LDS byte address = tb[reg & 3] + slot*4096 + 128*(reg >> 2)
    uint32_t tb[4];
    {
        const int gk = r >> 3, hk = (r >> 2) & 1, ek = r & 3, mm = 2 * gk + hk;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            tb[j] = lds_addr_of(ssm) + ((wave * 2 * SB_FLOATS) + gk * 256 + 128 * hk + ek + 4 * ((j + 4 * h) ^ mm)) * 4;
    }
    const uint32_t gaddr = lds_addr_of(gsm) + (wave * SB_FLOATS + 4 * h * 32 + r) * 4;      // + 128 * c_reg per register
    auto step = [&](int q0, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int QOFF = SLOT * 2 * DT_FLOATS * 4, DOOFF = QOFF + DT_FLOATS * 4;
        // issue order of a live wave's step:  tile(t+1) [4]  lse2 / delta(t) [2 plain loads, consumed in this step] | G(t+1) [4]
        // scores(t+2) [4].  At this barrier tile(t) must have landed; behind it the previous step issued G(t) and scores(t+1).
        sres_wait_vm((q0 > 0 && live) ? (gb0 ? 4 : 0) + (q0 + 32 < g.T ? 4 : 0) : 0);
        acr_barrier_nofence();
        if (q0 + 64 <= g.T) {
            dma_tile32_i(smem + (SLOT ^ 1) * 2 * DT_FLOATS, qb + (int64_t)(q0 + 32) * g.st, qoff, wave);
            dma_tile32_i(smem + (SLOT ^ 1) * 2 * DT_FLOATS + DT_FLOATS, dob + (int64_t)(q0 + 32) * g.ost, dooff, wave);
        } else if (q0 + 32 < g.T) {
            dma_tile32(smem + (SLOT ^ 1) * 2 * DT_FLOATS, qb, g.st, q0 + 32, g.T, wave, lane);
            dma_tile32(smem + (SLOT ^ 1) * 2 * DT_FLOATS + DT_FLOATS, dob, g.ost, q0 + 32, g.T, wave, lane);
        }
        SRES_FENCE();
        if (!live) return;
        const int qi = min(q0 + r, g.T - 1);
        const float lq_lane = lrow[qi], dq_lane = drow[qi];         // this step's lse2 / delta, one query per lane
        f32x16 dp = {0};
        rowop_x<DOOFF>(dp, lb, vreg);                      // dP[query = krow][key = r]
        // G(t) must have landed in this wave's tile: behind it are scores(t+1) [4], this step's tile(t+1) [4] (and the two plain
        // loads above, if they are still in flight: 8 is the stricter count)
        if (q0 > 0) sres_wait_vm(q0 + 32 < g.T ? 8 : 0);
        float s[16], gv[16];
#define SRES_RDS(REG) ACR_LDS_RD32(s[REG], tb[(REG) & 3], SLOT * SB_FLOATS * 4 + 128 * ((REG) >> 2))
#define SRES_RDG(REG) ACR_LDS_RD32(gv[REG], gaddr, 128 * (((REG) & 3) + 8 * ((REG) >> 2)))
        SRES_RDS(0); SRES_RDS(1); SRES_RDS(2); SRES_RDS(3); SRES_RDS(4); SRES_RDS(5); SRES_RDS(6); SRES_RDS(7);
        SRES_RDS(8); SRES_RDS(9); SRES_RDS(10); SRES_RDS(11); SRES_RDS(12); SRES_RDS(13); SRES_RDS(14); SRES_RDS(15);
        if (gb0 != nullptr) {
            SRES_RDG(0); SRES_RDG(1); SRES_RDG(2); SRES_RDG(3); SRES_RDG(4); SRES_RDG(5); SRES_RDG(6); SRES_RDG(7);
            SRES_RDG(8); SRES_RDG(9); SRES_RDG(10); SRES_RDG(11); SRES_RDG(12); SRES_RDG(13); SRES_RDG(14); SRES_RDG(15);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gv[0]), "+v"(gv[1]), "+v"(gv[2]), "+v"(gv[3]), "+v"(gv[4]), "+v"(gv[5]), "+v"(gv[6]),
                         "+v"(gv[7]), "+v"(gv[8]), "+v"(gv[9]), "+v"(gv[10]), "+v"(gv[11]), "+v"(gv[12]), "+v"(gv[13]), "+v"(gv[14]), "+v"(gv[15]));
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) gv[reg] = 0.f;
        }
#undef SRES_RDS
#undef SRES_RDG
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]),
                     "+v"(s[8]), "+v"(s[9]), "+v"(s[10]), "+v"(s[11]), "+v"(s[12]), "+v"(s[13]), "+v"(s[14]), "+v"(s[15]));
        if (q0 + 32 > g.T) {                               // last query block: rows beyond T are junk, P = 0 there
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (q0 + acr_krow(reg, h) >= g.T) s[reg] = -INFINITY;
        }
        f32x16 p, ds;
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int src = acr_krow(reg, h);              // lane that holds this accumulator row's query
            const float lq = __shfl(lq_lane, src), dq_ = __shfl(dq_lane, src);
            const float pv = __builtin_amdgcn_exp2f(s[reg] - lq);
            p[reg] = pv;
            ds[reg] = pv * (dp[reg] + gv[reg] * invH - dq_);
        }
        __builtin_amdgcn_s_setprio(0);
        // the private tiles have been read: refill G for the next step and the score slot for the step AFTER next
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (q0 + 32 < g.T) dma_g(q0 + 32);
        if (q0 + 64 < g.T) dma_scores((q0 >> 5) + 2, SLOT);
        SRES_FENCE();
        accop_x<DOOFF, 0, true>(dv0, p, lb);               // dV[key = krow][d = 32*blk + r]
        accop_x<DOOFF, 1, true>(dv1, p, lb);
        accop_x<QOFF, 0, true>(dk0, ds, lb);
        accop_x<QOFF, 1, true>(dk1, ds, lb);
    };
    for (int q0 = 0; q0 < g.T; q0 += 64) {
        step(q0, std::integral_constant<int, 0>{});
        if (q0 + 32 < g.T) step(q0 + 32, std::integral_constant<int, 1>{});
    }
    if (!live) return;
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
__global__ __launch_bounds__(256, 2) void attn_bwd_sres_kernel(AttnGeom g, const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, const float* __restrict__ d_o,
                                                               const float* __restrict__ lse2, const float* __restrict__ delta,
                                                               const float* __restrict__ sres, const float* __restrict__ gm,
                                                               int64_t gm_sb, int64_t gm_st, float* __restrict__ dq,
                                                               float* __restrict__ dk, float* __restrict__ dv, int ntail) {
    __shared__ __attribute__((aligned(1024))) float smem[4 * DT_FLOATS];       // [slot][Q | dO]  resp.  [slot][K | V]
    __shared__ __attribute__((aligned(1024))) float ssm[4 * 2 * SB_FLOATS];    // [wave][slot] score blocks (both bodies)
    __shared__ __attribute__((aligned(1024))) float gsm[4 * SB_FLOATS];        // [wave] G block (both bodies); 80 KB in all: two per CU
    // grid: [dK/dV full | dQ full | dK/dV split tails | dQ split tails]; tails (one per (b, h) and sweep) are dispatched last
    const int half = ((int)gridDim.x - 2 * ntail) >> 1;
    const int bid = (int)blockIdx.x;
    if (bid < half)
        attn_dkdv_sres_body(smem, ssm, gsm, bid, half, ntail != 0, g, q, v, d_o, lse2, delta, sres, gm, gm_sb, gm_st, dk, dv);
    else if (bid < 2 * half)
        attn_dq_sres_body(smem, ssm, gsm, bid - half, half, ntail != 0, g, k, v, d_o, lse2, delta, sres, gm, gm_sb, gm_st, dq);
    else if (bid < 2 * half + ntail)
        attn_dkdv_tail_body<4>(smem, ssm, g, q, v, d_o, lse2, delta, sres, gm, gm_sb, gm_st, dk, dv, acr_xcd_remap(bid - 2 * half, ntail));
    else
        attn_dq_tail_body<4>(smem, g, k, v, d_o, lse2, delta, sres, gm, gm_sb, gm_st, dq, acr_xcd_remap(bid - 2 * half - ntail, ntail));
}

// ---------------------------------------------------------------------------------------------
// launchers (called from attn_f32.hip)
// ---------------------------------------------------------------------------------------------
// one leftover 32-row block per (b, h) and enough full workgroups for the split to pay (see "Split tail" above)
static bool split_tail(int NB) { return acr_opt(ACR_OPT_ATTN_F32_NOSPLITTAIL) == 0 && (NB & 3) == 1 && NB >= 5; }

void acr_attn_fwd_f32_sres(const AttnGeom& g, const float* q, const float* k, const float* v, float* o, float* lse2, float* scores,
                           float* pmean, int64_t pmean_sb, int64_t pmean_st, hipStream_t st) {
    const int NB = (g.T + 31) / 32;
    // waves (32-query blocks) per workgroup: 4.  T = 785 is 25 blocks = 5 x 5 exactly, but five-wave workgroups measured 0.86 ms
    // against 0.64 ms at B = 32, H = 12 (160 VGPRs allow 12 wave slots per CU, i.e. only two 5-wave workgroups against three
    // 4-wave ones): the leftover block goes to a split-tail workgroup instead (attn_f32_sres_tails.h)
    const int ntail = split_tail(NB) ? g.B * g.H : 0;
    const int nmain = g.B * g.H * (ntail ? NB / 4 : (NB + 3) / 4);
    hipLaunchKernelGGL(attn_fwd_sres_kernel<4>, dim3(nmain + ntail), dim3(256), 0, st, g, q, k, v, o, lse2, scores, ntail);
    if (pmean) acr_attn_pmean_sres(g, scores, lse2, pmean, pmean_sb, pmean_st, st);
}

void acr_attn_pmean_sres(const AttnGeom& g, const float* scores, const float* lse2, float* pmean, int64_t pmean_sb, int64_t pmean_st,
                         hipStream_t st) {
    const int NB = (g.T + 31) / 32;
    hipLaunchKernelGGL(attn_pmean_sres_kernel, dim3((g.B * NB * NB + 3) / 4), dim3(256), 0, st, g, NB, scores, lse2, pmean, pmean_sb,
                       pmean_st);
}

void acr_attn_delta_sres(const AttnGeom& g, const float* scores, const float* o, const float* d_o, const float* lse2, const float* gm,
                         int64_t gm_sb, int64_t gm_st, float* delta, hipStream_t st) {
    const int NB = (g.T + 31) / 32;
    // groups of FOUR heads: with all twelve heads of ViT-B in one workgroup gm is fetched once instead of three times, but the
    // twelve-wave barrier per key block costs more than that saves (277 us against 254 us per launch at the bench shape)
#ifndef DELTA4_NW
#define DELTA4_NW 4
#endif
    if (gm != nullptr && (g.H % DELTA4_NW) == 0 && acr_opt(ACR_OPT_ATTN_DELTA_1HEAD) == 0)
        hipLaunchKernelGGL(attn_delta_sres4_kernel<DELTA4_NW>, dim3(g.B * NB * (g.H / DELTA4_NW)), dim3(64 * DELTA4_NW), 0, st, g, NB, scores, o, d_o,
                           lse2, gm, gm_sb, gm_st, delta);
    else
        hipLaunchKernelGGL(attn_delta_sres_kernel, dim3((g.B * NB * g.H + 3) / 4), dim3(256), 0, st, g, NB, scores, o, d_o, lse2, gm, gm_sb,
                           gm_st, delta);
}

void acr_attn_bwd_f32_sres(const AttnGeom& g, const float* q, const float* k, const float* v, const float* o, const float* d_o,
                           const float* lse2, const float* scores, const float* gm, int64_t gm_sb, int64_t gm_st, float* dq,
                           float* dk, float* dv, float* delta, hipStream_t st) {
    const int nt = (g.T + 127) / 128, NB = (g.T + 31) / 32;
    acr_attn_delta_sres(g, scores, o, d_o, lse2, gm, gm_sb, gm_st, delta, st);
    const int ntail = split_tail(NB) ? g.B * g.H : 0;
    const int nmain = g.B * g.H * (ntail ? NB / 4 : nt);
    hipLaunchKernelGGL(attn_bwd_sres_kernel, dim3(2 * nmain + 2 * ntail), dim3(256), 0, st, g, q, k, v, d_o, lse2, (const float*)delta,
                       scores, gm, gm_sb, gm_st, dq, dk, dv, ntail);
}
