// Split-tail workgroups of the resident-score attention (attn_f32_sres.hip -- read its header for the score-block layout), shared
// with the split-product generation (attn_f32_x3.hip): the ONE leftover 32-row block of every (b, h) -- T = 785 is 25 blocks = 6 x 4
// + 1 for four-wave workgroups, 3 x 8 + 1 for the eight-wave backward of attn_f32_x3.hip -- handled by a workgroup of a different
// shape whose NW waves split the OTHER dimension (forward / dQ: the key tiles, dK/dV: the query tiles) NW ways, each wave with a
// private, barrier-free tile stream (one 8 KB LDS image per wave), partial results merged through LDS in wave order
// (deterministic).  Exact-fp32 products (v_mfma_f32_32x32x2_f32) in both generations: under ACR_F32_BF16X3 that is 1/NB of the
// rows computed by the other fp32-accurate arithmetic, from the fp32 operands.
#pragma once
#include "acr_common.h"
#include "attn_f32.h"
#include "attn_f32_tiles.h"

#define SB_FLOATS 1024
// Cache policy of the score stream.  Every byte of `scores` is written once and read once per consumer, 983 MB per layer
// against 4 MB of L2 per XCD and 256 MB of Infinity Cache.  Measured per kernel (scripts/lab/attn_gen.py with lab builds):
// nontemporal loads take the head-mean stream from 235 to 183 us and the dQ body's loads the backward from 1549 to 1535 us;
// nontemporal stores the forward from 580 to 567 us; the row-term stream gets 5 % SLOWER with them (its G rows want to stay
// cached beside the scores) and the dK/dV body's LDS-DMA with the nt policy (aux = 2) is within noise: both keep the default.
#define SRES_LOAD_NT(p) __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p))
#define SRES_LOAD(p) (*reinterpret_cast<const f32x4*>(p))
#define SRES_STORE(p, v) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p))
#define SRES_LOAD_DQ(p) SRES_LOAD_NT(p)
#define SRES_DMA_AUX 0

__device__ __forceinline__ int64_t sres_block(const AttnGeom& g, int NB, int b, int hd, int qb, int kb) {
    return ((((int64_t)b * g.H + hd) * NB + qb) * NB + kb) * SB_FLOATS;
}

// ---------------------------------------------------------------------------------------------
// Split tail (all three MFMA sweeps).  T = 785 is 25 blocks of 32 rows = 6 workgroups of 4 blocks + ONE block: as a seventh
// workgroup with a single live wave it costs a full workgroup's time (a sweep is bound by its per-step barrier / DMA / softmax
// chain, not by the matrix pipe) and turns 3.0 rounds of the chip's workgroup slots into 3.5 -> 4.  When NB % 4 == 1 the
// last block of every (b, h) is therefore handled by a "tail" workgroup of a different shape, scheduled after all full ones:
// its four waves split the OTHER dimension (forward / dQ: the key tiles, dK/dV: the query tiles) four ways, each wave with a
// private, barrier-free tile stream (one 8 KB LDS image per wave: K then V, resp. dO then Q), and the four partial results
// are merged through LDS in wave order (deterministic).  A tail workgroup takes ~1/4 of the steps of a full one.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wave_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int NW>
__device__ __forceinline__ void attn_fwd_tail_body(float* smem, float* mlsh, const AttnGeom& g, const float* __restrict__ q,
                                                   const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ o,
                                                   float* __restrict__ lse2, float* __restrict__ sres, int id) {
    const int NB = (g.T + 31) >> 5;
    const int hd = id % g.H, b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = (NB - 1) * 32;                          // every wave: the same (last) query block
    const int64_t base = (int64_t)b * g.sb + (int64_t)hd * g.sh;
    const float* kb = k + base;
    const float* vb = v + base;
    float* tl = smem + wave * DT_FLOATS;                   // this wave's private tile image
    float qreg[32];
    rows_from_global(qreg, q + base, g.st, q0, g.T, r, h, g.scale * ACR_LOG2E);
    float m = -INFINITY, l = 0.f;
    f32x16 o0 = {0}, o1 = {0};
    const LaneBases lb = lane_bases_at(r, h, wave * DT_FLOATS * 4);
    const char* sm = reinterpret_cast<const char*>(smem);
    float* sblk = sres + sres_block(g, NB, b, hd, NB - 1, 0) + lane * 4;
    for (int kt = wave; kt < NB; kt += NW) {
        const int k0 = kt * 32;
        dma_tile32_one(tl, kb, g.st, k0, g.T, lane);
        wave_wait_dma();
        f32x16 s = {0};
        rowop_i<0>(s, sm, lb, qreg);
        if (k0 + 32 > g.T) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (k0 + acr_krow(reg, h) >= g.T) s[reg] = -INFINITY;
        }
        wave_wait_lds();                                   // the K image has been read: V may overwrite it
        dma_tile32_one(tl, vb, g.st, k0, g.T, lane);
        float* sp = sblk + (int64_t)kt * SB_FLOATS;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 t = {s[4 * gq], s[4 * gq + 1], s[4 * gq + 2], s[4 * gq + 3]};
            SRES_STORE(sp + gq * 256, t);
        }
        float mx = s[0];
#pragma unroll
        for (int reg = 1; reg < 16; ++reg) mx = fmaxf(mx, s[reg]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (__any(mx > m + 8.f)) {
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
        wave_wait_dma();
        accop_b_i<0, 0>(o0, p, sm, lb);
        accop_b_i<0, 1>(o1, p, sm, lb);
        wave_wait_lds();                                   // before the next K tile lands on this image
    }
    // ---- merge the four partial (m, l, O) in wave order.  Lane (r, h) of every wave holds the same query r.
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) { tl[e * 64 + lane] = o0[e]; tl[(16 + e) * 64 + lane] = o1[e]; }
    if (h == 0) { mlsh[wave * 64 + r] = m; mlsh[wave * 64 + 32 + r] = l; }
    __syncthreads();
    if (wave != 0) return;
    float M = m;
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, mlsh[w * 64 + r]);
    float L = 0.f;
    f32x16 a0 = {0}, a1 = {0};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const float f = __builtin_amdgcn_exp2f(mlsh[w * 64 + r] - M);      // a wave without tiles has m = -inf: weight 0
        L += mlsh[w * 64 + 32 + r] * f;
        const float* pw = smem + w * DT_FLOATS;
#pragma unroll
        for (int e = 0; e < 16; ++e) { a0[e] += pw[e * 64 + lane] * f; a1[e] += pw[(16 + e) * 64 + lane] * f; }
    }
    if (q0 + r < g.T) {
        const float inv = 1.f / L;
        float* ob = o + (int64_t)b * g.osb + (int64_t)(q0 + r) * g.ost + (int64_t)hd * g.osh;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            f32x4 a = {a0[4 * grp] * inv, a0[4 * grp + 1] * inv, a0[4 * grp + 2] * inv, a0[4 * grp + 3] * inv};
            f32x4 c = {a1[4 * grp] * inv, a1[4 * grp + 1] * inv, a1[4 * grp + 2] * inv, a1[4 * grp + 3] * inv};
            *reinterpret_cast<f32x4*>(ob + 8 * grp + 4 * h) = a;
            *reinterpret_cast<f32x4*>(ob + 32 + 8 * grp + 4 * h) = c;
        }
        if (h == 0) lse2[((int64_t)b * g.H + hd) * g.T + q0 + r] = M + log2f(L);
    }
}

// ---------------------------------------------------------------------------------------------
// split-tail workgroups of the backward sweeps (see "Split tail" at the top): the last query block's dQ with the key tiles
// dealt over the four waves, the last key block's dK / dV with the query tiles dealt over the four waves; private tile
// images, no barriers inside the loops, partial sums merged through LDS in wave order.
// ---------------------------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ void attn_dq_tail_body(float* smem, const AttnGeom& g, const float* __restrict__ k,
                                                  const float* __restrict__ v, const float* __restrict__ d_o,
                                                  const float* __restrict__ lse2, const float* __restrict__ delta,
                                                  const float* __restrict__ sres, const float* __restrict__ gm, int64_t gm_sb,
                                                  int64_t gm_st, float* __restrict__ dq, int id) {
    const int NB = (g.T + 31) >> 5;
    const int hd = id % g.H, b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = (NB - 1) * 32;
    const int64_t base = (int64_t)b * g.sb + (int64_t)hd * g.sh;
    const int64_t obase = (int64_t)b * g.osb + (int64_t)hd * g.osh;
    const float* kb = k + base;
    const float* vb = v + base;
    float* tl = smem + wave * DT_FLOATS;
    float doreg[32];
    rows_from_global(doreg, d_o + obase, g.ost, q0, g.T, r, h, 1.f);
    const bool qok = q0 + r < g.T;
    const float l2q = qok ? lse2[((int64_t)b * g.H + hd) * g.T + q0 + r] : INFINITY;
    const float dl = qok ? delta[((int64_t)b * g.H + hd) * g.T + q0 + r] : 0.f;
    const float invH = 1.f / (float)g.H;
    const float* grow = gm ? gm + (int64_t)b * gm_sb + (int64_t)min(q0 + r, g.T - 1) * gm_st : nullptr;
    const float* sblk = sres + sres_block(g, NB, b, hd, NB - 1, 0) + lane * 4;
    f32x16 dq0 = {0}, dq1 = {0};
    const LaneBases lb = lane_bases_at(r, h, wave * DT_FLOATS * 4);
    const char* sm = reinterpret_cast<const char*>(smem);
    for (int kt = wave; kt < NB; kt += NW) {
        const int k0 = kt * 32;
        dma_tile32_one(tl, vb, g.st, k0, g.T, lane);
        f32x4 s4[4], g4[4];
        const float* sp = sblk + (int64_t)kt * SB_FLOATS;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) s4[gq] = SRES_LOAD_DQ(sp + gq * 256);
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
        wave_wait_dma();
        f32x16 dp = {0};
        rowop_i<0>(dp, sm, lb, doreg);                     // dP^T[key = krow][query = r]
        wave_wait_lds();
        dma_tile32_one(tl, kb, g.st, k0, g.T, lane);
        f32x16 ds;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const float gv = g4[reg >> 2][reg & 3] * invH;
            ds[reg] = __builtin_amdgcn_exp2f(s4[reg >> 2][reg & 3] - l2q) * (dp[reg] + gv - dl);
        }
        wave_wait_dma();
        accop_a_i<0, 0>(dq0, ds, sm, lb);                  // dQ[query = krow][d = 32*blk + r]
        accop_a_i<0, 1>(dq1, ds, sm, lb);
        wave_wait_lds();
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) { tl[e * 64 + lane] = dq0[e]; tl[(16 + e) * 64 + lane] = dq1[e]; }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll 1                                        // one partial at a time: unrolled over eight waves the loads of all of them are hoisted and spill
    for (int w = 1; w < NW; ++w) {
        const float* pw = smem + w * DT_FLOATS;
#pragma unroll
        for (int e = 0; e < 16; ++e) { dq0[e] += pw[e * 64 + lane]; dq1[e] += pw[(16 + e) * 64 + lane]; }
    }
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

template <int NW>
__device__ __forceinline__ void attn_dkdv_tail_body(float* smem, float* ssm, const AttnGeom& g, const float* __restrict__ q,
                                                    const float* __restrict__ v, const float* __restrict__ d_o,
                                                    const float* __restrict__ lse2, const float* __restrict__ delta,
                                                    const float* __restrict__ sres, const float* __restrict__ gm, int64_t gm_sb,
                                                    int64_t gm_st, float* __restrict__ dk, float* __restrict__ dv, int id) {
    const int NB = (g.T + 31) >> 5;
    const int hd = id % g.H, b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int key0 = (NB - 1) * 32;                        // every wave: the same (last) key block
    const int64_t base = (int64_t)b * g.sb + (int64_t)hd * g.sh;
    const int64_t obase = (int64_t)b * g.osb + (int64_t)hd * g.osh;
    const float* qb = q + base;
    const float* dob = d_o + obase;
    const float* lrow = lse2 + ((int64_t)b * g.H + hd) * g.T;
    const float* drow = delta + ((int64_t)b * g.H + hd) * g.T;
    const float* scol = sres + sres_block(g, NB, b, hd, 0, NB - 1);
    const int64_t sstep = (int64_t)NB * SB_FLOATS;
    float* tl = smem + wave * DT_FLOATS;                   // dO tile, then Q tile
    float* sw = ssm + wave * 2 * SB_FLOATS;                // [score block | lse2 x 32, delta x 32]
    float* rcw = sw + SB_FLOATS;
    float vreg[32];
    rows_from_global(vreg, v + base, g.st, key0, g.T, r, h, 1.f);
    const int key = key0 + r;
    const int gcl = min(key, g.T - 1);
    const float invH = 1.f / (float)g.H;
    const float* gb0 = gm ? gm + (int64_t)b * gm_sb : nullptr;
    f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
    const LaneBases lb = lane_bases_at(r, h, wave * DT_FLOATS * 4);
    const char* sm = reinterpret_cast<const char*>(smem);
    const char* ssb = reinterpret_cast<const char*>(ssm);
    int tb[4];
    {
        const int gk = r >> 3, hk = (r >> 2) & 1, ek = r & 3, mm = 2 * gk + hk;
#pragma unroll
        for (int j = 0; j < 4; ++j) tb[j] = ((wave * 2 * SB_FLOATS) + gk * 256 + 128 * hk + ek + 4 * ((j + 4 * h) ^ mm)) * 4;
    }
    for (int qt = wave; qt < NB; qt += NW) {
        const int q0 = qt * 32;
        dma_tile32_one(tl, dob, g.ost, q0, g.T, lane);
        {
            const float* src = scol + (int64_t)qt * sstep;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                __builtin_amdgcn_global_load_lds((glb_vp)(src + gq * 256 + 4 * (lane ^ (2 * gq + (lane >> 5)))),
                                                 (lds_vp)(sw + gq * 256), 16, 0, 0);
        }
        dma_rowconst(rcw, lrow, drow, q0, g.T, lane);
        float gv[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            gv[reg] = gb0 ? gb0[(int64_t)min(q0 + acr_krow(reg, h), g.T - 1) * gm_st + gcl] * invH : 0.f;
        wave_wait_dma();
        f32x16 dp = {0};
        rowop_i<0>(dp, sm, lb, vreg);                      // dP[query = krow][key = r]
        f32x16 s;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) s[reg] = *reinterpret_cast<const float*>(ssb + tb[reg & 3] + 128 * (reg >> 2));
        if (q0 + 32 > g.T) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (q0 + acr_krow(reg, h) >= g.T) s[reg] = -INFINITY;
        }
        f32x16 p, ds;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(rcw + 8 * gq + 4 * h);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(rcw + 32 + 8 * gq + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int reg = 4 * gq + e;
                const float pv = __builtin_amdgcn_exp2f(s[reg] - l4[e]);
                p[reg] = pv;
                ds[reg] = pv * (dp[reg] + gv[reg] - d4[e]);
            }
        }
        accop_a_i<0, 0>(dv0, p, sm, lb);                   // dV[key = krow][d = 32*blk + r]
        accop_a_i<0, 1>(dv1, p, sm, lb);
        wave_wait_lds();                                   // dO image, score block and row constants have been read
        dma_tile32_one(tl, qb, g.st, q0, g.T, lane);
        wave_wait_dma();
        accop_a_i<0, 0>(dk0, ds, sm, lb);
        accop_a_i<0, 1>(dk1, ds, sm, lb);
        wave_wait_lds();
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        tl[e * 64 + lane] = dk0[e]; tl[(16 + e) * 64 + lane] = dk1[e];
        sw[e * 64 + lane] = dv0[e]; sw[(16 + e) * 64 + lane] = dv1[e];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll 1                                        // one partial at a time: unrolled over eight waves the loads of all of them are hoisted and spill
    for (int w = 1; w < NW; ++w) {
        const float* pk = smem + w * DT_FLOATS;
        const float* pv = ssm + w * 2 * SB_FLOATS;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            dk0[e] += pk[e * 64 + lane]; dk1[e] += pk[(16 + e) * 64 + lane];
            dv0[e] += pv[e * 64 + lane]; dv1[e] += pv[(16 + e) * 64 + lane];
        }
    }
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

