// Attention kernels computing in exact fp32 on the gfx950 f32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// This is the parity / inference precision of the library: every product is a k-ordered fp32 fmaf
// chain (same numerics class as the reference's fp32 cuBLAS path, models/vision_transformer.py:203-211),
// but P = softmax(q k^T) is never written to HBM:
//
//   attn_fwd      flash forward per (b, h, 64-query tile): O, base-2 row log-sum-exp
//   attn_tile_qk  64x64 (query x key) tiles of P recomputed from q, k, lse2:
//                   PMEAN  -> mean over heads, straight into the (B,L,T,T) stack slice (DPT/ACR.py:107-112)
//                   PROBS  -> per-head P (API compat, Attention.get_attn)
//                   DPROBS -> per-head dO V^T (API compat, Attention.get_attn_gradients)
//   attn_delta    delta[b,h,i] = rowsum(dO*O) + (1/H) sum_j P_h[i,j] G[b,i,j]
//   attn_dq       dQ per (b, h, 64-query tile)
//   attn_dkdv     dK, dV per (b, h, 64-key tile)          (no atomics anywhere: deterministic)
//
// MFMA operand convention used throughout (32x32x2: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D col = l&31,
// D row = krow(reg, l>>5)).  Two product shapes cover every matmul here:
//   rowop : acc[reg] = sum_d X[krow(reg,h)][d] * Y[l&31][d]      X rows from an LDS tile (ds_read_b128),
//           Y rows held in 32 VGPRs (lane (r,h) owns Y[r][32h .. 32h+31]); k-slot h pairs d = 32h+s.
//   accop : acc[reg'] += sum_{reg} A/B-from-accumulator: an accumulator tile Z[krow(reg,h)][l&31] is
//           already the A operand (lane index = output row) or B operand (lane index = output column) of
//           the next product, whose other operand rows Wt[krow(reg,h)][32*blk + (l&31)] come from LDS
//           with conflict-free ds_read_b32 -- no transposes, no lane shuffles.
// On this fp32 path the MFMA pipe is the bound (64 cycles per 32x32x2); LDS and VALU hide under it.
#include "acr_common.h"
#include "attn_f32.h"

#define LDP 68                     // LDS row pitch (floats): 64 + 4 -> conflict-free b128 row reads
#define TILE_FLOATS (32 * LDP)

// ---------------------------------------------------------------------------------------------
// building blocks
// ---------------------------------------------------------------------------------------------
// Stage a 32-row x 64-col tile (rows row0.. of a (T, 64) strided matrix) into LDS, zero-filling rows >= T.
template <typename T, int NTHREADS>
__device__ __forceinline__ void stage_tile(float* lds, const T* g, int64_t st, int row0, int Tn, int tid, float mul) {
#pragma unroll
    for (int i = 0; i < 512 / NTHREADS; ++i) {
        const int slot = tid + i * NTHREADS;
        const int row = slot >> 4, c4 = slot & 15;
        // unconditional load from a clamped row + select (a guarded load makes hipcc branch and drain vmcnt per chunk)
        f32x4 v = acr_load4<T>(g + (int64_t)min(row0 + row, Tn - 1) * st + c4 * 4) * mul;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(lds + row * LDP + c4 * 4) = (row0 + row < Tn) ? v : z;
    }
}

// Lane (r, h) loads row (row0 + r), columns [32h, 32h+32) into 32 registers (zeros beyond T).
template <typename T>
__device__ __forceinline__ void load_rows(float (&reg)[32], const T* g, int64_t st, int row0, int Tn, int lane, float mul) {
    const int r = lane & 31, h = lane >> 5;
    const float okm = (row0 + r < Tn) ? mul : 0.f;
    const T* p = g + (int64_t)min(row0 + r, Tn - 1) * st + 32 * h;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f32x4 v = acr_load4<T>(p + 4 * i) * okm;
        reg[4 * i + 0] = v[0]; reg[4 * i + 1] = v[1]; reg[4 * i + 2] = v[2]; reg[4 * i + 3] = v[3];
    }
}
// Same, from an LDS tile.
__device__ __forceinline__ void load_rows_lds(float (&reg)[32], const float* tile, int lane) {
    const float* p = tile + (lane & 31) * LDP + 32 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * i);
        reg[4 * i + 0] = v[0]; reg[4 * i + 1] = v[1]; reg[4 * i + 2] = v[2]; reg[4 * i + 3] = v[3];
    }
}

// acc[reg] += sum_d tile[krow(reg,h)][d] * Y[l&31][d]
__device__ __forceinline__ void mma_rowop(f32x16& acc, const float* tile, const float (&y)[32], int lane) {
    const float* ap = tile + (lane & 31) * LDP + 32 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f32x4 a = *reinterpret_cast<const f32x4*>(ap + 4 * i);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], y[4 * i + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], y[4 * i + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], y[4 * i + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], y[4 * i + 3], acc, 0, 0, 0);
    }
}
// z is the A operand (its lane index becomes the output row): acc[i = z-lane][j = tile column 32*blk + l&31]
__device__ __forceinline__ void mma_accop_a(f32x16& acc, const f32x16& z, const float* tile, int blk, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const float b = tile[acr_krow(reg, h) * LDP + 32 * blk + r];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z[reg], b, acc, 0, 0, 0);
    }
}
// z is the B operand (its lane index becomes the output column): acc[i = tile column][j = z-lane]
__device__ __forceinline__ void mma_accop_b(f32x16& acc, const f32x16& z, const float* tile, int blk, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const float a = tile[acr_krow(reg, h) * LDP + 32 * blk + r];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, z[reg], acc, 0, 0, 0);
    }
}


// ---------------------------------------------------------------------------------------------
// forward: grid = B*H*ceil(T/64) blocks (XCD-remapped), 2 waves, each wave owns 32 query rows
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(128) void attn_fwd_kernel(AttnGeom g, const T* __restrict__ q, const T* __restrict__ k,
                                                       const T* __restrict__ v, T* __restrict__ o,
                                                       float* __restrict__ lse2) {
    __shared__ __attribute__((aligned(16))) float kt[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float vt[TILE_FLOATS];
    const int nqt = (g.T + 63) >> 6;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int qt = id % nqt; id /= nqt;
    const int h = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int q0 = qt * 64 + wave * 32;
    const int64_t base = (int64_t)b * g.sb + (int64_t)h * g.sh;
    const T* qb = q + base; const T* kb = k + base; const T* vb = v + base;

    float qreg[32];
    load_rows<T>(qreg, qb, g.st, q0, g.T, lane, g.scale * ACR_LOG2E);
    float m = -INFINITY, l = 0.f;
    f32x16 o0 = {0}, o1 = {0};
    for (int k0 = 0; k0 < g.T; k0 += 32) {
        __syncthreads();
        stage_tile<T, 128>(kt, kb, g.st, k0, g.T, tid, 1.f);
        stage_tile<T, 128>(vt, vb, g.st, k0, g.T, tid, 1.f);
        __syncthreads();
        f32x16 s = {0};
        mma_rowop(s, kt, qreg, lane);                     // s[reg] = S2[key = k0 + krow][query = q0 + r]
        float mx = -INFINITY;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            if (k0 + acr_krow(reg, hh) >= g.T) s[reg] = -INFINITY;
            mx = fmaxf(mx, s[reg]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(m, mx);
        const float alpha = exp2f(m - mn);
        float rs = 0.f;
        f32x16 p;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) { p[reg] = exp2f(s[reg] - mn); rs += p[reg]; }
        rs += __shfl_xor(rs, 32);
        l = l * alpha + rs;
        m = mn;
        o0 *= alpha; o1 *= alpha;
        mma_accop_b(o0, p, vt, 0, lane);                  // o[reg] = O^T[d = 32*blk + krow][query = r]
        mma_accop_b(o1, p, vt, 1, lane);
    }
    if (q0 + r < g.T) {
        const float inv = 1.f / l;
        T* ob = o + (int64_t)b * g.osb + (int64_t)(q0 + r) * g.ost + (int64_t)h * g.osh;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            f32x4 a = {o0[4 * grp] * inv, o0[4 * grp + 1] * inv, o0[4 * grp + 2] * inv, o0[4 * grp + 3] * inv};
            f32x4 c = {o1[4 * grp] * inv, o1[4 * grp + 1] * inv, o1[4 * grp + 2] * inv, o1[4 * grp + 3] * inv};
            acr_store4<T>(ob + 8 * grp + 4 * hh, a);
            acr_store4<T>(ob + 32 + 8 * grp + 4 * hh, c);
        }
        if (hh == 0) lse2[((int64_t)b * g.H + h) * g.T + q0 + r] = m + log2f(l);
    }
}

// ---------------------------------------------------------------------------------------------
// 64x64 tiles of P (or dO V^T): grid = B*[H]*nt*nt blocks, 4 waves as 2(query) x 2(key)
// ---------------------------------------------------------------------------------------------
enum { TQK_PMEAN = 0, TQK_PROBS = 1, TQK_DPROBS = 2 };

template <typename T, int MODE>
__global__ __launch_bounds__(256) void attn_tile_qk_kernel(AttnGeom g, const T* __restrict__ xq, const T* __restrict__ xk,
                                                           const float* __restrict__ lse2, float* __restrict__ out,
                                                           int64_t out_sb, int64_t out_st) {
    __shared__ __attribute__((aligned(16))) float qs[2 * TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float ks[2 * TILE_FLOATS];
    const int nt = (g.T + 63) >> 6;
    int id = acr_xcd_remap(blockIdx.x, gridDim.x);
    const int kti = id % nt; id /= nt;
    const int qti = id % nt; id /= nt;
    int b, hsel;
    if (MODE == TQK_PMEAN) { b = id; hsel = 0; } else { hsel = id % g.H; b = id / g.H; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wq = wave >> 1, wk = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int q0 = qti * 64, k0 = kti * 64;
    const bool is_q = (MODE != TQK_DPROBS);              // row operand strides: q/k vs dO/v
    f32x16 acc = {0};
    const int h_lo = (MODE == TQK_PMEAN) ? 0 : hsel, h_hi = (MODE == TQK_PMEAN) ? g.H : hsel + 1;
    for (int h = h_lo; h < h_hi; ++h) {
        const T* qb = is_q ? xq + (int64_t)b * g.sb + (int64_t)h * g.sh : xq + (int64_t)b * g.osb + (int64_t)h * g.osh;
        const int64_t qst = is_q ? g.st : g.ost;
        const T* kb = xk + (int64_t)b * g.sb + (int64_t)h * g.sh;
        __syncthreads();
        stage_tile<T, 256>(qs, qb, qst, q0, g.T, tid, 1.f);
        stage_tile<T, 256>(qs + TILE_FLOATS, qb, qst, q0 + 32, g.T, tid, 1.f);
        const float kmul = (MODE == TQK_DPROBS) ? 1.f : g.scale * ACR_LOG2E;
        stage_tile<T, 256>(ks, kb, g.st, k0, g.T, tid, kmul);
        stage_tile<T, 256>(ks + TILE_FLOATS, kb, g.st, k0 + 32, g.T, tid, kmul);
        __syncthreads();
        float kreg[32];
        load_rows_lds(kreg, ks + wk * TILE_FLOATS, lane);
        f32x16 s = {0};
        mma_rowop(s, qs + wq * TILE_FLOATS, kreg, lane);  // s[reg] = X[query = krow][key = r]
        if (MODE == TQK_DPROBS) {
            acc = s;
        } else {
            const float* lrow = lse2 + ((int64_t)b * g.H + h) * g.T;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int qq = min(q0 + wq * 32 + acr_krow(reg, hh), g.T - 1);
                acc[reg] += exp2f(s[reg] - lrow[qq]);
            }
        }
    }
    const int key = k0 + wk * 32 + r;
    if (key < g.T) {
        float* ob = (MODE == TQK_PMEAN) ? out + (int64_t)b * out_sb
                                        : out + ((int64_t)b * g.H + hsel) * (int64_t)g.T * g.T;
        const float mul = (MODE == TQK_PMEAN) ? 1.f / (float)g.H : 1.f;
        const int64_t ost = (MODE == TQK_PMEAN) ? out_st : (int64_t)g.T;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int qq = q0 + wq * 32 + acr_krow(reg, hh);
            if (qq < g.T) ob[(int64_t)qq * ost + key] = acc[reg] * mul;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// delta[b,h,i] = rowsum(dO*O) + (1/H) sum_j P_h[i,j] G[b,i,j]; 2 waves x 32 query rows
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(128) void attn_delta_kernel(AttnGeom g, const T* __restrict__ q, const T* __restrict__ k,
                                                         const T* __restrict__ o, const T* __restrict__ d_o,
                                                         const float* __restrict__ lse2, const float* __restrict__ gm,
                                                         int64_t gm_sb, int64_t gm_st, float* __restrict__ delta) {
    __shared__ __attribute__((aligned(16))) float qs[2 * TILE_FLOATS];
    __shared__ float dsh[64];
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
    // rowsum(dO * O) for query q0 + r
    float part = 0.f;
    if (q0 + r < g.T) {
        const T* op = o + obase + (int64_t)(q0 + r) * g.ost + 32 * hh;
        const T* dp = d_o + obase + (int64_t)(q0 + r) * g.ost + 32 * hh;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4 a = acr_load4<T>(op + 4 * i), c = acr_load4<T>(dp + 4 * i);
            part += a[0] * c[0] + a[1] * c[1] + a[2] * c[2] + a[3] * c[3];
        }
    }
    part += __shfl_xor(part, 32);
    if (hh == 0) dsh[wave * 32 + r] = part;
    float rho[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) rho[reg] = 0.f;
    if (gm != nullptr) {
        stage_tile<T, 128>(qs, q + base, g.st, qt * 64, g.T, tid, 1.f);
        stage_tile<T, 128>(qs + TILE_FLOATS, q + base, g.st, qt * 64 + 32, g.T, tid, 1.f);
        float l2r[16];
        int goff[16];
        const float* lrow = lse2 + ((int64_t)b * g.H + h) * g.T;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int qq = q0 + acr_krow(reg, hh);
            const float lv = lrow[min(qq, g.T - 1)];
            l2r[reg] = (qq < g.T) ? lv : INFINITY;          // rows beyond T: p = exp2(-inf) = 0
            goff[reg] = min(qq, g.T - 1) * (int)gm_st;
        }
        __syncthreads();
        const float* grow = gm + (int64_t)b * gm_sb;
        for (int k0 = 0; k0 < g.T; k0 += 32) {
            float kreg[32];
            load_rows<T>(kreg, k + base, g.st, k0, g.T, lane, g.scale * ACR_LOG2E);
            const int key = k0 + r;
            const int kc = min(key, g.T - 1);
            float gv[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) gv[reg] = grow[goff[reg] + kc];     // unconditional (clamped) loads
            f32x16 s = {0};
            mma_rowop(s, qs + wave * TILE_FLOATS, kreg, lane);   // s[reg] = S2[query = krow][key = k0 + r]
            const float kmask = (key < g.T) ? 1.f : 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) rho[reg] += exp2f(s[reg] - l2r[reg]) * kmask * gv[reg];
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
            const int kr = acr_krow(reg, hh);
            if (q0 + kr < g.T) delta[((int64_t)b * g.H + h) * g.T + q0 + kr] = dsh[wave * 32 + kr] + rho[reg] * invH;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dQ: grid = B*H*ceil(T/64), 2 waves x 32 query rows, sweep over 32-key tiles
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(128) void attn_dq_kernel(AttnGeom g, const T* __restrict__ q, const T* __restrict__ k,
                                                      const T* __restrict__ v, const T* __restrict__ d_o,
                                                      const float* __restrict__ lse2, const float* __restrict__ delta,
                                                      const float* __restrict__ gm, int64_t gm_sb, int64_t gm_st,
                                                      T* __restrict__ dq) {
    __shared__ __attribute__((aligned(16))) float kt[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float vt[TILE_FLOATS];
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
    float qreg[32], doreg[32];
    load_rows<T>(qreg, q + base, g.st, q0, g.T, lane, g.scale * ACR_LOG2E);
    load_rows<T>(doreg, d_o + obase, g.ost, q0, g.T, lane, 1.f);
    const bool qok = q0 + r < g.T;
    const float l2 = qok ? lse2[((int64_t)b * g.H + h) * g.T + q0 + r] : 0.f;
    const float dl = qok ? delta[((int64_t)b * g.H + h) * g.T + q0 + r] : 0.f;
    const float invH = 1.f / (float)g.H;
    const float* grow = gm ? gm + (int64_t)b * gm_sb + (int64_t)min(q0 + r, g.T - 1) * gm_st : nullptr;
    f32x16 dq0 = {0}, dq1 = {0};
    for (int k0 = 0; k0 < g.T; k0 += 32) {
        __syncthreads();
        stage_tile<T, 128>(kt, k + base, g.st, k0, g.T, tid, 1.f);
        stage_tile<T, 128>(vt, v + base, g.st, k0, g.T, tid, 1.f);
        __syncthreads();
        f32x16 s = {0}, dp = {0};
        mma_rowop(s, kt, qreg, lane);                     // S2^T[key = krow][query = r]
        mma_rowop(dp, vt, doreg, lane);                   // dP^T[key][query]
        float gv[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) gv[reg] = 0.f;
        if (grow) {                                        // uniform branch; loads inside are unconditional (clamped)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) gv[reg] = grow[min(k0 + acr_krow(reg, hh), g.T - 1)] * invH;
        }
        f32x16 ds;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int key = k0 + acr_krow(reg, hh);
            const bool ok = qok && key < g.T;
            const float p = ok ? exp2f(s[reg] - l2) : 0.f;
            ds[reg] = p * (dp[reg] + gv[reg] - dl);
        }
        mma_accop_a(dq0, ds, kt, 0, lane);                // dQ[query = krow][d = 32*blk + r]
        mma_accop_a(dq1, ds, kt, 1, lane);
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int qq = q0 + acr_krow(reg, hh);
        if (qq < g.T) {
            T* p = dq + base + (int64_t)qq * g.st;
            acr_store1<T>(p + r, dq0[reg] * g.scale);
            acr_store1<T>(p + 32 + r, dq1[reg] * g.scale);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dK, dV: grid = B*H*ceil(T/64), 2 waves x 32 keys (K, V rows in registers), sweep over 32-query tiles
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(128) void attn_dkdv_kernel(AttnGeom g, const T* __restrict__ q, const T* __restrict__ k,
                                                        const T* __restrict__ v, const T* __restrict__ d_o,
                                                        const float* __restrict__ lse2, const float* __restrict__ delta,
                                                        const float* __restrict__ gm, int64_t gm_sb, int64_t gm_st,
                                                        T* __restrict__ dk, T* __restrict__ dv) {
    __shared__ __attribute__((aligned(16))) float qtile[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float dotile[TILE_FLOATS];
    __shared__ float l2s[32], dls[32];
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
    float kreg[32], vreg[32];
    load_rows<T>(kreg, k + base, g.st, key0, g.T, lane, g.scale * ACR_LOG2E);
    load_rows<T>(vreg, v + base, g.st, key0, g.T, lane, 1.f);
    const int key = key0 + r;
    const bool kok = key < g.T;
    const float invH = 1.f / (float)g.H;
    const float* lrow = lse2 + ((int64_t)b * g.H + h) * g.T;
    const float* drow = delta + ((int64_t)b * g.H + h) * g.T;
    const float* gbase = gm ? gm + (int64_t)b * gm_sb + min(key, g.T - 1) : nullptr;
    f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
    for (int q0 = 0; q0 < g.T; q0 += 32) {
        __syncthreads();
        stage_tile<T, 128>(qtile, q + base, g.st, q0, g.T, tid, 1.f);
        stage_tile<T, 128>(dotile, d_o + obase, g.ost, q0, g.T, tid, 1.f);
        if (tid < 32) {
            const bool ok = q0 + tid < g.T;
            const int qc = min(q0 + tid, g.T - 1);
            const float lv = lrow[qc], dvv = drow[qc];
            l2s[tid] = ok ? lv : INFINITY;                  // queries beyond T: p = exp2(-inf) = 0
            dls[tid] = ok ? dvv : 0.f;
        }
        __syncthreads();
        f32x16 s = {0}, dp = {0};
        mma_rowop(s, qtile, kreg, lane);                  // S2[query = krow][key = r]
        mma_rowop(dp, dotile, vreg, lane);                // dP[query][key]
        float gv[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) gv[reg] = 0.f;
        if (gbase) {                                       // uniform branch; loads inside are unconditional (clamped)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) gv[reg] = gbase[min(q0 + acr_krow(reg, hh), g.T - 1) * (int)gm_st] * invH;
        }
        f32x16 p, ds;
        const float kmask = kok ? 1.f : 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int kr = acr_krow(reg, hh);
            const float pv = exp2f(s[reg] - l2s[kr]) * kmask;
            p[reg] = pv;
            ds[reg] = pv * (dp[reg] + gv[reg] - dls[kr]);
        }
        mma_accop_a(dv0, p, dotile, 0, lane);             // dV[key = krow][d = 32*blk + r]
        mma_accop_a(dv1, p, dotile, 1, lane);
        mma_accop_a(dk0, ds, qtile, 0, lane);
        mma_accop_a(dk1, ds, qtile, 1, lane);
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int kk = key0 + acr_krow(reg, hh);
        if (kk < g.T) {
            T* pk = dk + base + (int64_t)kk * g.st;
            T* pv = dv + base + (int64_t)kk * g.st;
            acr_store1<T>(pk + r, dk0[reg] * g.scale);
            acr_store1<T>(pk + 32 + r, dk1[reg] * g.scale);
            acr_store1<T>(pv + r, dv0[reg]);
            acr_store1<T>(pv + 32 + r, dv1[reg]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// GETAM row 0 (DPT/ACR.py:177-215): one thread per key, heads looped inside
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void getam_row_kernel(AttnGeom g, const T* __restrict__ q, const T* __restrict__ k,
                                                       const T* __restrict__ v, const T* __restrict__ d_o,
                                                       const float* __restrict__ lse2, int batch, int func,
                                                       float* __restrict__ cam_row, int64_t row_stride) {
    __shared__ float q0s[64], do0s[64];
    const int j = blockIdx.x * 64 + threadIdx.x;
    batch += blockIdx.y;                                   // acr_getam_rows_accum: one grid row per sample
    cam_row += (int64_t)blockIdx.y * row_stride;
    float sum_g = 0.f, sum_cg = 0.f;
    for (int h = 0; h < g.H; ++h) {
        const int64_t base = (int64_t)batch * g.sb + (int64_t)h * g.sh;
        __syncthreads();
        q0s[threadIdx.x] = acr_load1<T>(q + base + threadIdx.x) * (g.scale * ACR_LOG2E);
        do0s[threadIdx.x] = acr_load1<T>(d_o + (int64_t)batch * g.osb + (int64_t)h * g.osh + threadIdx.x);
        __syncthreads();
        if (j < g.T) {
            const T* kr = k + base + (int64_t)j * g.st;
            const T* vr = v + base + (int64_t)j * g.st;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 16; ++d4) {
                f32x4 kv = acr_load4<T>(kr + 4 * d4), vv = acr_load4<T>(vr + 4 * d4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s = fmaf(kv[e], q0s[4 * d4 + e], s);
                    dp = fmaf(vv[e], do0s[4 * d4 + e], dp);
                }
            }
            const float p = exp2f(s - lse2[((int64_t)batch * g.H + h) * g.T]);
            sum_g += fmaxf(dp, 0.f);
            sum_cg += fmaxf(dp * p, 0.f);
        }
    }
    if (j < g.T) {
        const float invH = 1.f / (float)g.H;
        const float mg = sum_g * invH, mcg = sum_cg * invH;
        float val;
        switch (func) {
            case ACR_GETAM_GRAD: val = mg; break;
            case ACR_GETAM_CAM_GRAD: val = mcg; break;
            case ACR_GETAM_GRAD_S: val = mg * mg; break;
            default: val = mcg * mg; break;
        }
        cam_row[j] += val;
    }
}

// ---------------------------------------------------------------------------------------------
// host entry points
// ---------------------------------------------------------------------------------------------
// bf16-MFMA launchers (attn_bf16.hip)
bool acr_bf16_mfma_ok(const acr_attn_desc* d, const void* const* ptrs, int n);
void acr_attn_fwd_bf16(const acr_attn_desc* d, const void* q, const void* k, const void* v, void* o, float* lse2,
                       float* pmean, int64_t pmean_sb, int64_t pmean_st, hipStream_t st);
void acr_attn_bwd_bf16(const acr_attn_desc* d, const void* q, const void* k, const void* v, const void* o,
                       const void* d_o, const float* lse2, const float* gm, int64_t gm_sb, int64_t gm_st, void* dq, void* dk,
                       void* dv, float* delta, hipStream_t st);
void acr_attn_probs_bf16(const acr_attn_desc* d, const void* q, const void* k, const float* lse2, float* probs,
                         hipStream_t st);
void acr_attn_dprobs_bf16(const acr_attn_desc* d, const void* d_o, const void* v, float* dprobs, hipStream_t st);

static int check_desc(const acr_attn_desc* d, const char* who) {
    ACR_CHECK_ARG(d != nullptr, "%s: null desc", who);
    ACR_CHECK_ARG(d->B > 0 && d->H > 0 && d->T > 0, "%s: bad geometry B=%d H=%d T=%d", who, d->B, d->H, d->T);
    ACR_CHECK_ARG(d->head_dim == 64, "%s: head_dim %d unsupported (library is built for 64)", who, d->head_dim);
    ACR_CHECK_ARG(d->dtype == ACR_F32 || d->dtype == ACR_BF16 || d->dtype == ACR_BF16_F32MATH || d->dtype == ACR_F32_BF16X3,
                  "%s: unknown dtype %d", who, d->dtype);
    ACR_CHECK_ARG((d->qkv_sb % 4) == 0 && (d->qkv_st % 4) == 0 && (d->qkv_sh % 4) == 0 && (d->o_sb % 4) == 0 &&
                      (d->o_st % 4) == 0 && (d->o_sh % 4) == 0,
                  "%s: strides must be multiples of 4 elements (vector loads)", who);
    ACR_CHECK_ARG((int64_t)d->B * d->H * ((d->T + 63) / 64) < (1ll << 31), "%s: grid too large", who);
    return ACR_OK;
}
static AttnGeom geom(const acr_attn_desc* d) {
    AttnGeom g;
    g.B = d->B; g.H = d->H; g.T = d->T; g.scale = d->scale;
    g.sb = d->qkv_sb; g.st = d->qkv_st; g.sh = d->qkv_sh;
    g.osb = d->o_sb; g.ost = d->o_st; g.osh = d->o_sh;
    return g;
}
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

template <typename T>
static int attn_fwd_t(const acr_attn_desc* d, const void* q, const void* k, const void* v, void* o, float* lse2,
                      float* pmean, int64_t pmean_sb, int64_t pmean_st, hipStream_t st) {
    AttnGeom g = geom(d);
    const int nqt = (d->T + 63) / 64;
    hipLaunchKernelGGL((attn_fwd_kernel<T>), dim3(d->B * d->H * nqt), dim3(128), 0, st, g, (const T*)q, (const T*)k,
                       (const T*)v, (T*)o, lse2);
    if (pmean)
        hipLaunchKernelGGL((attn_tile_qk_kernel<T, TQK_PMEAN>), dim3(d->B * nqt * nqt), dim3(256), 0, st, g,
                           (const T*)q, (const T*)k, (const float*)lse2, pmean, pmean_sb, pmean_st);
    return acr_check_launch("acr_attn_fwd");
}

extern "C" int acr_attn_fwd(const acr_attn_desc* d, const void* q, const void* k, const void* v, void* o,
                            float* lse2, float* pmean, int64_t pmean_sb, int64_t pmean_st, void* stream) {
    int rc = check_desc(d, "acr_attn_fwd");
    if (rc) return rc;
    ACR_CHECK_ARG(q && k && v && o && lse2, "acr_attn_fwd: null pointer");
    const bool f32 = d->dtype == ACR_F32;
    ACR_CHECK_ARG(f32 ? (aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o))
                      : (aligned8(q) && aligned8(k) && aligned8(v) && aligned8(o)),
                  "acr_attn_fwd: q/k/v/o must be 16-byte (fp32) / 8-byte (bf16) aligned");
    ACR_CHECK_ARG(!pmean || (pmean_st >= d->T && pmean_sb >= (int64_t)d->T * pmean_st),
                  "acr_attn_fwd: pmean row pitch < T or batch stride < T*pitch");
#ifdef ACR_LAB_ATTN_GEN1          /* lab build (scripts/lab/build_variant.sh): the first-generation register-staged fp32 sweeps */
    if (f32) return attn_fwd_t<float>(d, q, k, v, o, lse2, pmean, pmean_sb, pmean_st, (hipStream_t)stream);
#endif
    if (f32) {
        acr_attn_fwd_f32_dma(geom(d), (const float*)q, (const float*)k, (const float*)v, (float*)o, lse2, pmean, pmean_sb, pmean_st,
                             (hipStream_t)stream);
        return acr_check_launch("acr_attn_fwd");
    }
    const void* ptrs[4] = {q, k, v, o};
    if (d->dtype == ACR_BF16 && acr_bf16_mfma_ok(d, ptrs, 4)) {
        acr_attn_fwd_bf16(d, q, k, v, o, lse2, pmean, pmean_sb, pmean_st, (hipStream_t)stream);
        return acr_check_launch("acr_attn_fwd");
    }
    return attn_fwd_t<__bf16>(d, q, k, v, o, lse2, pmean, pmean_sb, pmean_st, (hipStream_t)stream);
}

template <typename T>
static int attn_bwd_t(const acr_attn_desc* d, const void* q, const void* k, const void* v, const void* o,
                      const void* d_o, const float* lse2, const float* gm, int64_t gm_sb, int64_t gm_st, void* dq,
                      void* dk, void* dv, float* delta, hipStream_t st) {
    AttnGeom g = geom(d);
    const int nqt = (d->T + 63) / 64;
    const dim3 grid(d->B * d->H * nqt);
    hipLaunchKernelGGL((attn_delta_kernel<T>), grid, dim3(128), 0, st, g, (const T*)q, (const T*)k, (const T*)o,
                       (const T*)d_o, lse2, gm, gm_sb, gm_st, delta);
    hipLaunchKernelGGL((attn_dkdv_kernel<T>), grid, dim3(128), 0, st, g, (const T*)q, (const T*)k, (const T*)v,
                       (const T*)d_o, lse2, (const float*)delta, gm, gm_sb, gm_st, (T*)dk, (T*)dv);
    hipLaunchKernelGGL((attn_dq_kernel<T>), grid, dim3(128), 0, st, g, (const T*)q, (const T*)k, (const T*)v,
                       (const T*)d_o, lse2, (const float*)delta, gm, gm_sb, gm_st, (T*)dq);
    return acr_check_launch("acr_attn_bwd");
}

extern "C" int acr_attn_bwd(const acr_attn_desc* d, const void* q, const void* k, const void* v, const void* o,
                            const void* d_o, const float* lse2, const float* gmean, int64_t gmean_sb, int64_t gmean_st,
                            void* dq, void* dk, void* dv, float* delta_ws, void* stream) {
    int rc = check_desc(d, "acr_attn_bwd");
    if (rc) return rc;
    ACR_CHECK_ARG(q && k && v && o && d_o && lse2 && dq && dk && dv && delta_ws, "acr_attn_bwd: null pointer");
    ACR_CHECK_ARG(!gmean || (gmean_st >= d->T && gmean_sb >= (int64_t)d->T * gmean_st),
                  "acr_attn_bwd: gmean row pitch < T or batch stride < T*pitch");
    const bool f32 = d->dtype == ACR_F32;
    ACR_CHECK_ARG(f32 ? (aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o) && aligned16(d_o))
                      : (aligned8(q) && aligned8(k) && aligned8(v) && aligned8(o) && aligned8(d_o)),
                  "acr_attn_bwd: inputs must be 16-byte (fp32) / 8-byte (bf16) aligned");
#ifdef ACR_LAB_ATTN_GEN1
    if (f32)
        return attn_bwd_t<float>(d, q, k, v, o, d_o, lse2, gmean, gmean_sb, gmean_st, dq, dk, dv, delta_ws, (hipStream_t)stream);
#endif
    if (f32) {
        acr_attn_bwd_f32_dma(geom(d), (const float*)q, (const float*)k, (const float*)v, (const float*)o, (const float*)d_o, lse2,
                             gmean, gmean_sb, gmean_st, (float*)dq, (float*)dk, (float*)dv, delta_ws, (hipStream_t)stream);
        return acr_check_launch("acr_attn_bwd");
    }
    const void* ptrs[8] = {q, k, v, o, d_o, dq, dk, dv};
    // the bf16-MFMA dQ kernel pulls G in 16-byte groups: it needs a row pitch that is a multiple of 4 floats and
    // covers roundup4(T); other layouts take the exact-fp32 kernels (still HIP, any pitch)
    const bool g_ok = !gmean || ((gmean_st & 3) == 0 && gmean_st >= ((d->T + 3) & ~3) && (gmean_sb & 3) == 0 &&
                                 (reinterpret_cast<uintptr_t>(gmean) & 15) == 0);
    if (d->dtype == ACR_BF16 && g_ok && acr_bf16_mfma_ok(d, ptrs, 8)) {
        acr_attn_bwd_bf16(d, q, k, v, o, d_o, lse2, gmean, gmean_sb, gmean_st, dq, dk, dv, delta_ws, (hipStream_t)stream);
        return acr_check_launch("acr_attn_bwd");
    }
    return attn_bwd_t<__bf16>(d, q, k, v, o, d_o, lse2, gmean, gmean_sb, gmean_st, dq, dk, dv, delta_ws, (hipStream_t)stream);
}

// ---- resident-score generation (fp32 only; attn_f32_sres.hip) ------------------------------------------------------------
extern "C" int64_t acr_attn_scores_floats(const acr_attn_desc* d) {
    if (d == nullptr || d->B <= 0 || d->H <= 0 || d->T <= 0) return 0;
    if (d->dtype == ACR_F32_BF16X3) return acr_attn_x3_scores_floats(geom(d));
    const int64_t nb = (d->T + 31) / 32;
    return (int64_t)d->B * d->H * nb * nb * 1024;
}

extern "C" int64_t acr_attn_bwd_ws_floats(const acr_attn_desc* d) {
    if (d == nullptr || d->B <= 0 || d->H <= 0 || d->T <= 0) return 0;
    if (d->dtype == ACR_F32_BF16X3) return acr_attn_x3_bwd_ws_floats(geom(d));
    return (int64_t)d->B * d->H * d->T;
}

// the split-product kernels address the bf16 planes with 32-bit lane offsets inside one 32-row tile and need the plain packed
// layouts of the reference's activations (head h of a token = 64 contiguous elements)
static int check_x3(const acr_attn_desc* d, const char* who) {
    ACR_CHECK_ARG(d->qkv_sh >= 64 && d->o_sh >= 64, "%s: ACR_F32_BF16X3 needs head strides >= 64", who);
    ACR_CHECK_ARG((int64_t)32 * d->H * 64 < (1ll << 30), "%s: too many heads", who);
    return ACR_OK;
}

extern "C" int acr_attn_fwd_scores(const acr_attn_desc* d, const void* q, const void* k, const void* v, void* o, float* lse2,
                                   float* scores, float* pmean, int64_t pmean_sb, int64_t pmean_st, void* stream) {
    int rc = check_desc(d, "acr_attn_fwd_scores");
    if (rc) return rc;
    if (d->dtype != ACR_F32 && d->dtype != ACR_F32_BF16X3) {
        acr_set_error("acr_attn_fwd_scores: fp32 tensors only (the bf16 kernels recompute the logits)");
        return ACR_ERR_UNSUPPORTED;
    }
    ACR_CHECK_ARG(q && k && v && o && lse2 && scores, "acr_attn_fwd_scores: null pointer");
    ACR_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o) && aligned16(scores),
                  "acr_attn_fwd_scores: q/k/v/o/scores must be 16-byte aligned");
    ACR_CHECK_ARG(!pmean || (pmean_st >= d->T && pmean_sb >= (int64_t)d->T * pmean_st),
                  "acr_attn_fwd_scores: pmean row pitch < T or batch stride < T*pitch");
    if (d->dtype == ACR_F32_BF16X3) {
        rc = check_x3(d, "acr_attn_fwd_scores");
        if (rc) return rc;
        acr_attn_fwd_f32_x3(geom(d), (const float*)q, (const float*)k, (const float*)v, (float*)o, lse2, scores, pmean, pmean_sb, pmean_st,
                            (hipStream_t)stream);
        return acr_check_launch("acr_attn_fwd_scores");
    }
    acr_attn_fwd_f32_sres(geom(d), (const float*)q, (const float*)k, (const float*)v, (float*)o, lse2, scores, pmean, pmean_sb,
                          pmean_st, (hipStream_t)stream);
    return acr_check_launch("acr_attn_fwd_scores");
}

// acr_attn_fwd_scores for ACR_F32_BF16X3 whose output ALSO leaves as the operand image of the product that reads it (proj):
// the image pass over o (one read + the image's write, 42 us per layer at the bench shape) folds into the forward's epilogue.
extern "C" int acr_attn_fwd_scores_oimg(const acr_attn_desc* d, const void* q, const void* k, const void* v, void* o, float* lse2,
                                        float* scores, float* pmean, int64_t pmean_sb, int64_t pmean_st, float* o_image, void* stream) {
    int rc = check_desc(d, "acr_attn_fwd_scores_oimg");
    if (rc) return rc;
    if (d->dtype != ACR_F32_BF16X3) {
        acr_set_error("acr_attn_fwd_scores_oimg: ACR_F32_BF16X3 only (the image is the split-product operand form)");
        return ACR_ERR_UNSUPPORTED;
    }
    ACR_CHECK_ARG(q && k && v && o && lse2 && scores && o_image, "acr_attn_fwd_scores_oimg: null pointer");
    ACR_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o) && aligned16(scores) && aligned16(o_image),
                  "acr_attn_fwd_scores_oimg: q/k/v/o/scores/o_image must be 16-byte aligned");
    ACR_CHECK_ARG(!pmean || (pmean_st >= d->T && pmean_sb >= (int64_t)d->T * pmean_st),
                  "acr_attn_fwd_scores_oimg: pmean row pitch < T or batch stride < T*pitch");
    ACR_CHECK_ARG(d->o_sh == 64 && d->o_st == (int64_t)d->H * 64 && d->o_sb == (int64_t)d->T * d->o_st,
                  "acr_attn_fwd_scores_oimg: o must be the dense (B, T, H*64) activation (its image is that of the (B*T) x (H*64) matrix)");
    rc = check_x3(d, "acr_attn_fwd_scores_oimg");
    if (rc) return rc;
    acr_attn_fwd_f32_x3(geom(d), (const float*)q, (const float*)k, (const float*)v, (float*)o, lse2, scores, pmean, pmean_sb, pmean_st,
                        (hipStream_t)stream, reinterpret_cast<char*>(o_image));
    return acr_check_launch("acr_attn_fwd_scores_oimg");
}

// 1 when acr_attn_fwd_scores_oimg is the faster forward for this problem (split products, and a T whose leftover block the plain
// forward would not hand to split-tail workgroups), 0 when the caller should run acr_attn_fwd_scores + an image pass over o.
extern "C" int acr_attn_fwd_oimg_offered(const acr_attn_desc* d) {
    if (!d || d->dtype != ACR_F32_BF16X3 || d->T <= 0) return 0;
    return acr_attn_x3_fwd_uses_split_tail(d->T) ? 0 : 1;
}

extern "C" int acr_attn_bwd_scores(const acr_attn_desc* d, const void* q, const void* k, const void* v, const void* o,
                                   const void* d_o, const float* lse2, const float* scores, const float* gmean, int64_t gmean_sb,
                                   int64_t gmean_st, void* dq, void* dk, void* dv, float* delta_ws, void* stream) {
    int rc = check_desc(d, "acr_attn_bwd_scores");
    if (rc) return rc;
    if (d->dtype != ACR_F32 && d->dtype != ACR_F32_BF16X3) {
        acr_set_error("acr_attn_bwd_scores: fp32 tensors only (the bf16 kernels recompute the logits)");
        return ACR_ERR_UNSUPPORTED;
    }
    ACR_CHECK_ARG(q && k && v && o && d_o && lse2 && scores && dq && dk && dv && delta_ws, "acr_attn_bwd_scores: null pointer");
    ACR_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(o) && aligned16(d_o) && aligned16(scores),
                  "acr_attn_bwd_scores: inputs must be 16-byte aligned");
    ACR_CHECK_ARG(!gmean || (gmean_st >= d->T && gmean_sb >= (int64_t)d->T * gmean_st),
                  "acr_attn_bwd_scores: gmean row pitch < T or batch stride < T*pitch");
    ACR_CHECK_ARG(!gmean || ((gmean_st & 3) == 0 && (gmean_sb & 3) == 0 && aligned16(gmean)),
                  "acr_attn_bwd_scores: gmean must be 16-byte aligned with pitch and batch stride multiples of 4 floats");
    if (d->dtype == ACR_F32_BF16X3) {                        // q, k, v are read from the planes the forward left behind the scores
        rc = check_x3(d, "acr_attn_bwd_scores");
        if (rc) return rc;
        ACR_CHECK_ARG(aligned16(delta_ws) && aligned16(dq) && aligned16(dk) && aligned16(dv), "acr_attn_bwd_scores: delta_ws / dq / dk / dv must be 16-byte aligned");
        acr_attn_bwd_f32_x3(geom(d), (const float*)q, (const float*)k, (const float*)v, (const float*)o, (const float*)d_o, lse2, scores,
                            gmean, gmean_sb, gmean_st, (float*)dq, (float*)dk, (float*)dv, delta_ws, (hipStream_t)stream);
        return acr_check_launch("acr_attn_bwd_scores");
    }
    acr_attn_bwd_f32_sres(geom(d), (const float*)q, (const float*)k, (const float*)v, (const float*)o, (const float*)d_o, lse2,
                          scores, gmean, gmean_sb, gmean_st, (float*)dq, (float*)dk, (float*)dv, delta_ws, (hipStream_t)stream);
    return acr_check_launch("acr_attn_bwd_scores");
}

extern "C" int acr_attn_probs(const acr_attn_desc* d, const void* q, const void* k, const float* lse2, float* probs,
                              void* stream) {
    int rc = check_desc(d, "acr_attn_probs");
    if (rc) return rc;
    ACR_CHECK_ARG(q && k && lse2 && probs, "acr_attn_probs: null pointer");
    AttnGeom g = geom(d);
    const int nt = (d->T + 63) / 64;
    ACR_CHECK_ARG((int64_t)d->B * d->H * nt * nt < (1ll << 31), "acr_attn_probs: grid too large");
    const dim3 grid(d->B * d->H * nt * nt);
    const void* pp[2] = {q, k};
    if (d->dtype == ACR_F32)
        hipLaunchKernelGGL((attn_tile_qk_kernel<float, TQK_PROBS>), grid, dim3(256), 0, (hipStream_t)stream, g,
                           (const float*)q, (const float*)k, lse2, probs, (int64_t)0, (int64_t)0);
    else if (d->dtype == ACR_BF16 && acr_bf16_mfma_ok(d, pp, 2))
        acr_attn_probs_bf16(d, q, k, lse2, probs, (hipStream_t)stream);
    else
        hipLaunchKernelGGL((attn_tile_qk_kernel<__bf16, TQK_PROBS>), grid, dim3(256), 0, (hipStream_t)stream, g,
                           (const __bf16*)q, (const __bf16*)k, lse2, probs, (int64_t)0, (int64_t)0);
    return acr_check_launch("acr_attn_probs");
}

extern "C" int acr_attn_dprobs(const acr_attn_desc* d, const void* d_o, const void* v, float* dprobs, void* stream) {
    int rc = check_desc(d, "acr_attn_dprobs");
    if (rc) return rc;
    ACR_CHECK_ARG(d_o && v && dprobs, "acr_attn_dprobs: null pointer");
    AttnGeom g = geom(d);
    const int nt = (d->T + 63) / 64;
    ACR_CHECK_ARG((int64_t)d->B * d->H * nt * nt < (1ll << 31), "acr_attn_dprobs: grid too large");
    const dim3 grid(d->B * d->H * nt * nt);
    const void* pp[2] = {d_o, v};
    if (d->dtype == ACR_F32)
        hipLaunchKernelGGL((attn_tile_qk_kernel<float, TQK_DPROBS>), grid, dim3(256), 0, (hipStream_t)stream, g,
                           (const float*)d_o, (const float*)v, (const float*)nullptr, dprobs, (int64_t)0, (int64_t)0);
    else if (d->dtype == ACR_BF16 && acr_bf16_mfma_ok(d, pp, 2))
        acr_attn_dprobs_bf16(d, d_o, v, dprobs, (hipStream_t)stream);
    else
        hipLaunchKernelGGL((attn_tile_qk_kernel<__bf16, TQK_DPROBS>), grid, dim3(256), 0, (hipStream_t)stream, g,
                           (const __bf16*)d_o, (const __bf16*)v, (const float*)nullptr, dprobs, (int64_t)0, (int64_t)0);
    return acr_check_launch("acr_attn_dprobs");
}

extern "C" int acr_getam_row_accum(const acr_attn_desc* d, const void* q, const void* k, const void* v,
                                   const void* d_o, const float* lse2, int32_t batch, int32_t func, float* cam_row,
                                   void* stream) {
    int rc = check_desc(d, "acr_getam_row_accum");
    if (rc) return rc;
    ACR_CHECK_ARG(q && k && v && d_o && lse2 && cam_row, "acr_getam_row_accum: null pointer");
    ACR_CHECK_ARG(batch >= 0 && batch < d->B, "acr_getam_row_accum: batch %d out of range", batch);
    ACR_CHECK_ARG(func >= 0 && func <= 3, "acr_getam_row_accum: unknown func %d", func);
    AttnGeom g = geom(d);
    const dim3 grid((d->T + 63) / 64);
    if (d->dtype == ACR_F32)
        hipLaunchKernelGGL((getam_row_kernel<float>), grid, dim3(64), 0, (hipStream_t)stream, g, (const float*)q,
                           (const float*)k, (const float*)v, (const float*)d_o, lse2, batch, func, cam_row, (int64_t)0);
    else
        hipLaunchKernelGGL((getam_row_kernel<__bf16>), grid, dim3(64), 0, (hipStream_t)stream, g, (const __bf16*)q,
                           (const __bf16*)k, (const __bf16*)v, (const __bf16*)d_o, lse2, batch, func, cam_row, (int64_t)0);
    return acr_check_launch("acr_getam_row_accum");
}

// the same for EVERY sample of the batch in one launch: cam_rows[b] (row pitch row_stride >= T) += f(...) of sample b
extern "C" int acr_getam_rows_accum(const acr_attn_desc* d, const void* q, const void* k, const void* v, const void* d_o,
                                    const float* lse2, int32_t func, float* cam_rows, int64_t row_stride, void* stream) {
    int rc = check_desc(d, "acr_getam_rows_accum");
    if (rc) return rc;
    ACR_CHECK_ARG(q && k && v && d_o && lse2 && cam_rows, "acr_getam_rows_accum: null pointer");
    ACR_CHECK_ARG(func >= 0 && func <= 3, "acr_getam_rows_accum: unknown func %d", func);
    ACR_CHECK_ARG(row_stride >= d->T && d->B < 65536, "acr_getam_rows_accum: row pitch < T or batch too large");
    ACR_CHECK_ARG(d->dtype != ACR_F32_BF16X3, "acr_getam_rows_accum: exact kernels only (pass ACR_F32)");
    AttnGeom g = geom(d);
    const dim3 grid((d->T + 63) / 64, d->B);
    if (d->dtype == ACR_F32)
        hipLaunchKernelGGL((getam_row_kernel<float>), grid, dim3(64), 0, (hipStream_t)stream, g, (const float*)q,
                           (const float*)k, (const float*)v, (const float*)d_o, lse2, 0, func, cam_rows, row_stride);
    else
        hipLaunchKernelGGL((getam_row_kernel<__bf16>), grid, dim3(64), 0, (hipStream_t)stream, g, (const __bf16*)q,
                           (const __bf16*)k, (const __bf16*)v, (const __bf16*)d_o, lse2, 0, func, cam_rows, row_stride);
    return acr_check_launch("acr_getam_rows_accum");
}
