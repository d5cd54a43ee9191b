// fp32 (reference precision) attention, second generation: the same five sweeps as attn_f32.hip, rebuilt around what the
// fp32 GEMM work measured on gfx950 (scripts/lab):
//   * tiles stream global -> LDS by LDS-DMA (global_load_lds_dwordx4) into a two-slot ring, ONE barrier per step: no staging
//     registers, no ds_write path (VGPR traffic of loads / LDS stores costs the 64-cycle fp32 MFMA ~1.5 % per instruction),
//     and the next tile is in flight during the whole current step instead of being waited for between two barriers
//     (the first generation spent 34 % of its wave cycles parked on s_waitcnt / s_barrier);
//   * four waves per workgroup share every streamed tile (half the DMA instructions per MFMA of the two-wave version);
//   * unpadded 256-byte rows (what the DMA writes) with a 16-byte-chunk XOR swizzle, chunk ^ (row & 15), applied on the
//     DMA's per-lane SOURCE address and mirrored on every read: ds_read_b128 row reads (16-lane groups see 16 different
//     row residues -> 16 different slots) and ds_read_b32 column-block reads (a permutation inside one row) are
//     bank-conflict free.
// Numerics are unchanged: every product is a k-ordered fp32 fmaf chain on v_mfma_f32_32x32x2_f32, fp32 softmax; nothing is
// summed with atomics.  Operand conventions (rowop / accop) as in attn_f32.hip.
//
// Partial last tile (measured, scripts/lab/attn_tail.py): a workgroup costs about the same whether 4 or 1 of its waves have
// rows -- a step is bound by its barrier / DMA / softmax latency chain, not by the matrix pipe -- so with T = 785 = 6 x 128 + 17
// the seventh tile of every (b, h) costs 1/7 of a sweep for 2 % of the rows, and 7 x 384 workgroups are 3.5 / 5.25 rounds of
// the 768 / 512 resident slots (T = 785 takes 29 % longer per image than T = 768; B = 96 is 14 % cheaper per image than
// B = 32).  Tried and dropped, all correct but none faster over the four sweeps (3.36 ms per layer as is):
//   * partial tiles last in the grid with their row-less waves exiting at once: the freed wave slots cannot be refilled (a
//     workgroup is dispatched only when EVERY SIMD has a free slot) and the lone wave has to issue all 16 DMA pieces of a
//     step itself: forward -2 %, dK/dV +7 %;
//   * partial tiles as separate 64-thread workgroups after the full tiles: 3.35 ms (a lone wave is as slow per step);
//   * the same concurrently on a low-priority helper stream: the narrow workgroups still keep a full one from being
//     dispatched beside them: backward -3 %, forward +2 %;
//   * dQ at 3 waves per SIMD (168 VGPRs): spills 82 registers, 2x slower.
#include <type_traits>

#include "acr_common.h"
#include "attn_f32.h"
#include "attn_f32_tiles.h"


// ---------------------------------------------------------------------------------------------
// forward: one workgroup = (b, h, 128 queries); wave w owns queries q0 + 32w ..; K/V tiles of 32 keys stream through LDS
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_fwd_dma_kernel(AttnGeom g, const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, float* __restrict__ o,
                                                              float* __restrict__ lse2) {
    __shared__ __attribute__((aligned(1024))) float smem[4 * DT_FLOATS];       // [slot][K | V]
    const int nqt = (g.T + 127) >> 7;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int qt = id % nqt; id /= nqt;
    const int hd = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = qt * 128 + wave * 32;
    const bool live = q0 < g.T;                            // wave-uniform: waves past the end only help with the DMA
    const int64_t base = (int64_t)b * g.sb + (int64_t)hd * g.sh;
    const float* kb = k + base;
    const float* vb = v + base;
    dma_tile32(smem, kb, g.st, 0, g.T, wave, lane);
    dma_tile32(smem + DT_FLOATS, vb, g.st, 0, g.T, wave, lane);
    float qreg[32];
    rows_from_global(qreg, q + base, g.st, q0, g.T, r, h, g.scale * ACR_LOG2E);
    float m = -INFINITY, l = 0.f;
    f32x16 o0 = {0}, o1 = {0};
    const LaneBases lb = lane_bases(r, h);
    int doff[2];
    dma_offsets32(doff, g.st, wave, lane);
    const char* sm = reinterpret_cast<const char*>(smem);
    // one 32-key step on ring slot SLOT (compile time: every LDS address below is lane base + immediate)
    auto step = [&](int k0, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int KOFF = SLOT * 2 * DT_FLOATS * 4, VOFF = KOFF + DT_FLOATS * 4;
        acr_dma_barrier();                                 // slot SLOT has landed; the other slot is free
        // The co-resident wave of the other workgroup streams 64-cycle fp32 MFMAs through this SIMD; at equal priority this
        // wave's DMA issue and softmax VALU get one issue slot per MFMA gap (measured with s_memtime stamps: 4 DMA
        // instructions 1 280 cycles, ~100 VALU instructions 2 750 cycles).  VALU cannot overlap the fp32 MFMA anyway, so the
        // short non-matrix phases run at raised priority and the partner's chain resumes right after.
        __builtin_amdgcn_s_setprio(2);
        if (k0 + 64 <= g.T) {                              // next tile fully inside: precomputed lane offsets, uniform base
            dma_tile32_i(smem + (SLOT ^ 1) * 2 * DT_FLOATS, kb + (int64_t)(k0 + 32) * g.st, doff, wave);
            dma_tile32_i(smem + (SLOT ^ 1) * 2 * DT_FLOATS + DT_FLOATS, vb + (int64_t)(k0 + 32) * g.st, doff, wave);
        } else if (k0 + 32 < g.T) {                        // partial last tile: clamped rows
            dma_tile32(smem + (SLOT ^ 1) * 2 * DT_FLOATS, kb, g.st, k0 + 32, g.T, wave, lane);
            dma_tile32(smem + (SLOT ^ 1) * 2 * DT_FLOATS + DT_FLOATS, vb, g.st, k0 + 32, g.T, wave, lane);
        }
        __builtin_amdgcn_s_setprio(0);
        if (!live) return;
        f32x16 s = {0};
        rowop_i<KOFF>(s, sm, lb, qreg);                    // s[reg] = S2[key = k0 + krow][query = q0 + r]
        __builtin_amdgcn_s_setprio(2);
        if (k0 + 32 > g.T) {                               // only the last key tile has keys beyond T (uniform branch)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (k0 + acr_krow(reg, h) >= g.T) s[reg] = -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int reg = 1; reg < 16; ++reg) mx = fmaxf(mx, s[reg]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // Deferred rescale: the running reference m moves only when some row's maximum has grown by more than 2^8 since
        // it was set (p then stays below 2^8: no overflow, full fp32 precision); most steps skip the 32 multiplies of O and
        // the exp2 of alpha.  exp2 = one v_exp_f32.
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
        __builtin_amdgcn_s_setprio(0);
        accop_b_i<VOFF, 0>(o0, p, sm, lb);                 // o[reg] = O^T[d = 32*blk + krow][query = r]
        accop_b_i<VOFF, 1>(o1, p, sm, lb);
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
// head-mean of P (DPT/ACR.py:107-112): one workgroup = (b, 128 queries, 128 keys), 4 waves as 2 x 2 of 64 x 64; the heads
// are looped INSIDE (deterministic mean, no atomics).  Per head S = q k^T is a K = 64 GEMM streamed as two 32-deep chunks
// through the two-slot ring (rows of 128 bytes: the GEMM kernels' swizzle, (row >> 1) & 7 on 16-byte chunks).
// ---------------------------------------------------------------------------------------------
#define PM_TILE 4096               // 128 rows x 32 floats
__device__ __forceinline__ void dma_rows128(float* lds, const float* __restrict__ g, int64_t st, int row0, int Tn, int c0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;
        const int row = piece * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        const float* src = g + (int64_t)min(row0 + row, Tn - 1) * st + c0 + lc * 4;
        __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(lds + piece * 256), 16, 0, 0);
    }
}
__device__ __forceinline__ f32x4 frag128(const float* s, int row, int qq, int h) {
    return *reinterpret_cast<const f32x4*>(s + row * 32 + (((2 * qq + h) ^ ((row >> 1) & 7)) << 2));
}

__global__ __launch_bounds__(256, 2) void attn_pmean_dma_kernel(AttnGeom g, const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ lse2, float* __restrict__ out,
                                                                int64_t out_sb, int64_t out_st) {
    __shared__ __attribute__((aligned(1024))) float smem[4 * PM_TILE];         // [slot][Q chunk | K chunk]
    __shared__ float lsh[16 * 128];                                            // lse2 of the 128 queries, up to 16 heads at a time
    const int nt = (g.T + 127) >> 7;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int kti = id % nt; id /= nt;
    const int qti = id % nt;
    const int b = id / nt;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5, wq = wave >> 1, wk = wave & 1;
    const int q0 = qti * 128, k0 = kti * 128;
    const float sc = g.scale * ACR_LOG2E;
    f32x16 pm[2][2], acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { pm[i][j][e] = 0.f; acc[i][j][e] = 0.f; }
    const int nstep = 2 * g.H;                              // (head, 32-wide chunk of the head dimension)
    auto issue = [&](int step, int slot) {
        const int hd = step >> 1, c0 = (step & 1) * 32;
        const int64_t base = (int64_t)b * g.sb + (int64_t)hd * g.sh;
        dma_rows128(smem + slot * 2 * PM_TILE, q + base, g.st, q0, g.T, c0, wave, lane);
        dma_rows128(smem + slot * 2 * PM_TILE + PM_TILE, k + base, g.st, k0, g.T, c0, wave, lane);
    };
    issue(0, 0);
    int cur = 0;
    for (int step = 0; step < nstep; ++step, cur ^= 1) {
        const int hd = step >> 1;
        if ((step & 31) == 0) {                             // (re)fill the lse2 table for heads hd .. hd+15 (before the barrier)
            for (int i = tid; i < 16 * 128; i += 256) {
                const int hh = hd + (i >> 7), qq = q0 + (i & 127);
                lsh[i] = (hh < g.H && qq < g.T) ? lse2[((int64_t)b * g.H + hh) * g.T + qq] : INFINITY;   // rows beyond T: p = 0
            }
        }
        acr_dma_barrier();
        if (step + 1 < nstep) issue(step + 1, cur ^ 1);
        const float* sq = smem + cur * 2 * PM_TILE;
        const float* sk = sq + PM_TILE;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            f32x4 av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = frag128(sq, wq * 64 + i * 32 + r, qq, h);
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = frag128(sk, wk * 64 + j * 32 + r, qq, h);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bv[j][s], acc[i][j], 0, 0, 0);
        }
        if (step & 1) {                                     // head complete: acc[i][j][e] = S[query = krow(e,h)][key = r]
            const float* lrow = lsh + (hd & 15) * 128 + wq * 64;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float l2 = lrow[i * 32 + acr_krow(e, h)];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        pm[i][j][e] += __builtin_amdgcn_exp2f(fmaf(acc[i][j][e], sc, -l2));
                        acc[i][j][e] = 0.f;
                    }
                }
        }
    }
    const float mul = 1.f / (float)g.H;
    float* ob = out + (int64_t)b * out_sb;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int key = k0 + wk * 64 + j * 32 + r;
        if (key >= g.T) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int qq = q0 + wq * 64 + i * 32 + acr_krow(e, h);
                if (qq < g.T) ob[(int64_t)qq * out_st + key] = pm[i][j][e] * mul;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// delta[b,h,i] = rowsum(dO*O) + (1/H) sum_j P_h[i,j] G[b,i,j]: workgroup = (b, h, 128 queries), wave = 32 queries whose Q
// tile sits in LDS (A operand, queries on the accumulator rows so that G[i][k0 + lane] loads coalesce); K tiles stream.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_delta_dma_kernel(AttnGeom g, const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ o, const float* __restrict__ d_o,
                                                                const float* __restrict__ lse2, const float* __restrict__ gm,
                                                                int64_t gm_sb, int64_t gm_st, float* __restrict__ delta) {
    __shared__ __attribute__((aligned(1024))) float smem[6 * DT_FLOATS];       // [4 Q tiles | 2 K slots]
    __shared__ float dsh[128];
    const int nqt = (g.T + 127) >> 7;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
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
    float part = 0.f;
    if (q0 + r < g.T) {
        const float* op = o + obase + (int64_t)(q0 + r) * g.ost + 32 * h;
        const float* dp = d_o + obase + (int64_t)(q0 + r) * g.ost + 32 * h;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(op + 4 * i), c = *reinterpret_cast<const f32x4*>(dp + 4 * i);
            part += a[0] * c[0] + a[1] * c[1] + a[2] * c[2] + a[3] * c[3];
        }
    }
    part += __shfl_xor(part, 32);
    if (h == 0) dsh[wave * 32 + r] = part;
    float rho[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) rho[reg] = 0.f;
    if (gm != nullptr) {                                    // uniform over the launch
        float* qtiles = smem;
        float* kslots = smem + 4 * DT_FLOATS;
#pragma unroll
        for (int w = 0; w < 4; ++w) dma_tile32(qtiles + w * DT_FLOATS, q + base, g.st, qt * 128 + w * 32, g.T, wave, lane);
        dma_tile32(kslots, k + base, g.st, 0, g.T, wave, lane);
        float l2r[16];
        int goff[16];
        const float* lrow = lse2 + ((int64_t)b * g.H + hd) * g.T;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int qq = q0 + acr_krow(reg, h);
            const float lv = lrow[min(qq, g.T - 1)];
            l2r[reg] = (qq < g.T) ? lv : INFINITY;          // rows beyond T: p = exp2(-inf) = 0
            goff[reg] = min(qq, g.T - 1) * (int)gm_st;
        }
        const float* grow = gm + (int64_t)b * gm_sb;
        const float sc = g.scale * ACR_LOG2E;
        int cur = 0;
        for (int k0 = 0; k0 < g.T; k0 += 32, cur ^= 1) {
            acr_dma_barrier();
            if (k0 + 32 < g.T) dma_tile32(kslots + (cur ^ 1) * DT_FLOATS, k + base, g.st, k0 + 32, g.T, wave, lane);
            if (!live) continue;
            const int key = k0 + r;
            const int kc = min(key, g.T - 1);
            float gv[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) gv[reg] = grow[goff[reg] + kc];      // unconditional (clamped), coalesced over lanes
            float kreg[32];
            rows_from_lds(kreg, kslots + cur * DT_FLOATS, r, h);
            f32x16 s = {0};
            rowop(s, qtiles + wave * DT_FLOATS, kreg, r, h);  // s[reg] = S[query = krow][key = k0 + r] (unscaled)
            const float kmask = (key < g.T) ? 1.f : 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) rho[reg] = fmaf(__builtin_amdgcn_exp2f(fmaf(s[reg], sc, -l2r[reg])) * kmask, gv[reg], rho[reg]);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) rho[reg] += __shfl_xor(rho[reg], off);
        }
    }
    __syncthreads();
    if (r == 0) {
        const float invH = 1.f / (float)g.H;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int kr = acr_krow(reg, h);
            if (q0 + kr < g.T) delta[((int64_t)b * g.H + hd) * g.T + q0 + kr] = dsh[wave * 32 + kr] + rho[reg] * invH;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dQ: workgroup = (b, h, 128 queries); Q and dO rows of the wave's 32 queries in registers; K/V tiles stream
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void attn_dq_body(float* smem, int bid, int nblk, const AttnGeom& g, const float* __restrict__ q,
                                             const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ d_o,
                                             const float* __restrict__ lse2, const float* __restrict__ delta,
                                             const float* __restrict__ gm, int64_t gm_sb, int64_t gm_st, float* __restrict__ dq) {
    const int nqt = (g.T + 127) >> 7;
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
    dma_tile32(smem, kb, g.st, 0, g.T, wave, lane);
    dma_tile32(smem + DT_FLOATS, vb, g.st, 0, g.T, wave, lane);
    float qreg[32], doreg[32];
    rows_from_global(qreg, q + base, g.st, q0, g.T, r, h, g.scale * ACR_LOG2E);
    rows_from_global(doreg, d_o + obase, g.ost, q0, g.T, r, h, 1.f);
    const bool qok = q0 + r < g.T;
    const float l2q = qok ? lse2[((int64_t)b * g.H + hd) * g.T + q0 + r] : INFINITY;    // queries beyond T: p = exp2(-inf) = 0
    const float dl = qok ? delta[((int64_t)b * g.H + hd) * g.T + q0 + r] : 0.f;
    const float invH = 1.f / (float)g.H;
    const float* grow = gm ? gm + (int64_t)b * gm_sb + (int64_t)min(q0 + r, g.T - 1) * gm_st : nullptr;
    f32x16 dq0 = {0}, dq1 = {0};
    int cur = 0;
    for (int k0 = 0; k0 < g.T; k0 += 32, cur ^= 1) {
        acr_dma_barrier();
        if (k0 + 32 < g.T) {
            dma_tile32(smem + (cur ^ 1) * 2 * DT_FLOATS, kb, g.st, k0 + 32, g.T, wave, lane);
            dma_tile32(smem + (cur ^ 1) * 2 * DT_FLOATS + DT_FLOATS, vb, g.st, k0 + 32, g.T, wave, lane);
        }
        if (!live) continue;
        const float* kt = smem + cur * 2 * DT_FLOATS;
        const float* vt = kt + DT_FLOATS;
        float gv[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) gv[reg] = 0.f;
        if (grow) {                                         // uniform branch; loads inside are unconditional (clamped)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) gv[reg] = grow[min(k0 + acr_krow(reg, h), g.T - 1)] * invH;
        }
        f32x16 s = {0}, dp = {0};
        rowop(s, kt, qreg, r, h);                           // S2^T[key = krow][query = r]
        rowop(dp, vt, doreg, r, h);                         // dP^T[key][query]
        f32x16 ds;
        if (k0 + 32 > g.T) {                               // last key tile: keys beyond T contribute nothing (uniform branch)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (k0 + acr_krow(reg, h) >= g.T) s[reg] = -INFINITY;
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) ds[reg] = __builtin_amdgcn_exp2f(s[reg] - l2q) * (dp[reg] + gv[reg] - dl);
        accop_a(dq0, ds, kt, 0, r, h);                      // dQ[query = krow][d = 32*blk + r]
        accop_a(dq1, ds, kt, 1, r, h);
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
// dK, dV: workgroup = (b, h, 128 keys); K and V rows of the wave's 32 keys in registers; Q / dO tiles (and the 32 queries'
// lse2 / delta, one 256-byte DMA) stream
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void attn_dkdv_body(float* smem, float* rc, int bid, int nblk, const AttnGeom& g, const float* __restrict__ q,
                                               const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ d_o,
                                               const float* __restrict__ lse2, const float* __restrict__ delta,
                                               const float* __restrict__ gm, int64_t gm_sb, int64_t gm_st, float* __restrict__ dk,
                                               float* __restrict__ dv) {
    const int nkt = (g.T + 127) >> 7;
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
    dma_tile32(smem, qb, g.st, 0, g.T, wave, lane);
    dma_tile32(smem + DT_FLOATS, dob, g.ost, 0, g.T, wave, lane);
    if (wave == 0) dma_rowconst(rc, lrow, drow, 0, g.T, lane);
    float kreg[32], vreg[32];
    rows_from_global(kreg, k + base, g.st, key0, g.T, r, h, g.scale * ACR_LOG2E);
    rows_from_global(vreg, v + base, g.st, key0, g.T, r, h, 1.f);
    const int key = key0 + r;
    const bool kok = key < g.T;
    const float invH = 1.f / (float)g.H;
    const float* gbase = gm ? gm + (int64_t)b * gm_sb + min(key, g.T - 1) : nullptr;
    f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
    int cur = 0;
    for (int q0 = 0; q0 < g.T; q0 += 32, cur ^= 1) {
        acr_dma_barrier();
        if (q0 + 32 < g.T) {
            dma_tile32(smem + (cur ^ 1) * 2 * DT_FLOATS, qb, g.st, q0 + 32, g.T, wave, lane);
            dma_tile32(smem + (cur ^ 1) * 2 * DT_FLOATS + DT_FLOATS, dob, g.ost, q0 + 32, g.T, wave, lane);
            if (wave == 0) dma_rowconst(rc + (cur ^ 1) * 64, lrow, drow, q0 + 32, g.T, lane);
        }
        if (!live) continue;
        const float* qtile = smem + cur * 2 * DT_FLOATS;
        const float* dotile = qtile + DT_FLOATS;
        const float* rcs = rc + cur * 64;
        float gv[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) gv[reg] = 0.f;
        if (gbase) {                                        // uniform branch; loads inside are unconditional (clamped)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) gv[reg] = gbase[(int64_t)min(q0 + acr_krow(reg, h), g.T - 1) * gm_st] * invH;
        }
        f32x16 s = {0}, dp = {0};
        rowop(s, qtile, kreg, r, h);                        // S2[query = krow][key = r]
        rowop(dp, dotile, vreg, r, h);                      // dP[query][key]
        f32x16 p, ds;
        if (q0 + 32 > g.T) {                               // last query tile holds clamped copies of row T-1 beyond T: p = 0 there
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (q0 + acr_krow(reg, h) >= g.T) s[reg] = -INFINITY;
        }
        if (!kok) {                                         // keys beyond T (lanes of the last wave only)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) s[reg] = -INFINITY;
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int kr = acr_krow(reg, h);
            const float pv = __builtin_amdgcn_exp2f(s[reg] - rcs[kr]);
            p[reg] = pv;
            ds[reg] = pv * (dp[reg] + gv[reg] - rcs[32 + kr]);
        }
        accop_a(dv0, p, dotile, 0, r, h);                   // dV[key = krow][d = 32*blk + r]
        accop_a(dv1, p, dotile, 1, r, h);
        accop_a(dk0, ds, qtile, 0, r, h);
        accop_a(dk1, ds, qtile, 1, r, h);
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

// dK/dV and dQ in ONE launch: the two sweeps are independent and each is a whole number of equally long workgroups on 512
// resident slots (5.25 rounds at B = 32, T = 785) -- launched separately each pays its own partly filled last round.  The first
// half of the grid does dK/dV, the second dQ; they share the LDS ring (a workgroup is one or the other).
__global__ __launch_bounds__(256, 2) void attn_bwd_dma_kernel(AttnGeom g, const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, const float* __restrict__ d_o,
                                                              const float* __restrict__ lse2, const float* __restrict__ delta,
                                                              const float* __restrict__ gm, int64_t gm_sb, int64_t gm_st,
                                                              float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv) {
    __shared__ __attribute__((aligned(1024))) float smem[4 * DT_FLOATS];
    __shared__ __attribute__((aligned(256))) float rc[2 * 64];                 // dK/dV: [slot][lse2 x 32 | delta x 32]
    const int half = gridDim.x >> 1;
    if ((int)blockIdx.x < half)
        attn_dkdv_body(smem, rc, blockIdx.x, half, g, q, k, v, d_o, lse2, delta, gm, gm_sb, gm_st, dk, dv);
    else
        attn_dq_body(smem, blockIdx.x - half, half, g, q, k, v, d_o, lse2, delta, gm, gm_sb, gm_st, dq);
}

// ---------------------------------------------------------------------------------------------
// launchers (called from attn_f32.hip for fp32 tensors)
// ---------------------------------------------------------------------------------------------
void acr_attn_fwd_f32_dma(const AttnGeom& g, const float* q, const float* k, const float* v, float* o, float* lse2, float* pmean,
                          int64_t pmean_sb, int64_t pmean_st, hipStream_t st) {
    const int nt = (g.T + 127) / 128;
    hipLaunchKernelGGL(attn_fwd_dma_kernel, dim3(g.B * g.H * nt), dim3(256), 0, st, g, q, k, v, o, lse2);
    if (pmean)
        hipLaunchKernelGGL(attn_pmean_dma_kernel, dim3(g.B * nt * nt), dim3(256), 0, st, g, q, k, (const float*)lse2, pmean, pmean_sb,
                           pmean_st);
}

void acr_attn_bwd_f32_dma(const AttnGeom& g, const float* q, const float* k, const float* v, const float* o, const float* d_o,
                          const float* lse2, const float* gm, int64_t gm_sb, int64_t gm_st, float* dq, float* dk, float* dv,
                          float* delta, hipStream_t st) {
    const int nt = (g.T + 127) / 128;
    const dim3 grid(g.B * g.H * nt);
    hipLaunchKernelGGL(attn_delta_dma_kernel, grid, dim3(256), 0, st, g, q, k, o, d_o, lse2, gm, gm_sb, gm_st, delta);
    hipLaunchKernelGGL(attn_bwd_dma_kernel, dim3(2 * grid.x), dim3(256), 0, st, g, q, k, v, d_o, lse2, (const float*)delta, gm, gm_sb, gm_st,
                       dq, dk, dv);
}
