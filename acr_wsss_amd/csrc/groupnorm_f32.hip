// Fused GroupNorm(32) [+ residual add] [+ ReLU] of the ResNetV2 stem at the REFERENCE precision (fp32 NCHW), forward and
// backward (models/layers/norm_act.py:69-85, models/resnetv2.py:205-215).
//
// A (sample, group) of the fp32 stem is up to 401 KB (8 channels x 112^2 x 4 B): it does not fit the register file the way
// the bf16 kernel (groupnorm.hip) keeps it, so these kernels STREAM it twice -- the second pass re-reads what the first
// just pulled through L2 / Infinity Cache:
//   forward : pass 1 x -> mean, rstd (shifted sums: sum(x - x0), sum((x - x0)^2) with x0 = the group's first element, fp32,
//             no catastrophic cancellation); pass 2 x (+ resid) -> y.        2 reads (+1) + 1 write
//             (stock: row-moments + normalise + clamp kernels = 3 reads + 2 writes, plus the residual add's 2 reads + 1 write)
//   backward: pass 1 dy, x (+ resid for the ReLU mask) -> per-channel sum(dy), sum(dy * xhat) and the two group sums;
//             pass 2 dy, x (+ resid) -> dx (+ dresid).                        4 (6) reads + 1 (2) writes
//             (stock: threshold-backward, internal-gradients, elementwise backward, gamma/beta kernels, the add's backward)
// Deterministic: every sum is a fixed-order block reduction (no atomics); d(gamma), d(beta) per (sample, channel) are summed
// over samples in order by a second kernel.  HW must be a multiple of 4 (16-byte vectors inside a channel).
#include "acr_common.h"

#define GNF_GROUPS 32
// Cache policy of the second (last) pass: nontemporal loads and stores.  The activations are 100-400 MB per tensor -- nothing
// the next kernel could still find in the 4 MB L2 / 256 MB Infinity Cache -- and without the hint the pass evicts the lines
// its own first pass just pulled in for the neighbouring workgroups.  Measured in the fp32 bench (rocprofv3 sums per step):
// forward 3.37 -> 2.97 ms, backward 6.67 -> 6.02 ms.
#define GNF_LD2(p) __builtin_nontemporal_load(&(p))
#define GNF_ST(p, v) __builtin_nontemporal_store(v, &(p))
enum { GNF_NONE = 0, GNF_RELU = 1, GNF_ADD_RELU = 2 };

template <int NT>
__device__ __forceinline__ float gnf_block_sum(float v, float* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                                        // sh may still be read from a previous reduction
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += sh[w];           // same fixed order in every thread
    return t;
}

// Work decomposition inside a workgroup (round 5).  The first version walked a (sample, group) with all NT threads in lockstep: ONE
// 16-byte load in flight per thread (`s_waitcnt vmcnt(0)` in every iteration, found in the ISA), the channel of every vector by an
// integer division, and -- in the backward -- two block reductions (four barriers) per CHANNEL: 3.0 TB/s, half of what the HBM
// gives.  Now the group is cut into UNITS = (channel, segment of the channel's pixels); a wave owns whole units (no barrier while
// it streams them), keeps FOUR independent 16-byte loads per lane in flight, takes gamma / beta of its unit's channel as
// wave-uniform scalars, and the per-unit sums meet in LDS once per pass, where they are added in unit order (deterministic).
#define GNF_MAXU 64                      // units per (sample, group): max(channels per group, waves) <= 64
struct GnfUnits { int S, segv, units; };                    // segments per channel, vectors per segment, cg * S
__device__ __forceinline__ float gnf_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
// body(v, a) for every vector v of [v0, v1) of this lane (stride 64), four loads in flight; LD = how the vectors are loaded
#define GNF_FOR4(xv, v0, v1, lane, LDM, BODY)                                                     \
    {                                                                                              \
        int v_ = (v0) + (lane);                                                                    \
        for (; v_ + 192 < (v1); v_ += 256) {                                                       \
            const f32x4 a0_ = LDM((xv)[v_]), a1_ = LDM((xv)[v_ + 64]), a2_ = LDM((xv)[v_ + 128]), a3_ = LDM((xv)[v_ + 192]); \
            BODY(v_, a0_) BODY(v_ + 64, a1_) BODY(v_ + 128, a2_) BODY(v_ + 192, a3_)              \
        }                                                                                          \
        for (; v_ < (v1); v_ += 64) { const f32x4 a0_ = LDM((xv)[v_]); BODY(v_, a0_) }             \
    }
#define GNF_LD1(p) (p)

// The ReLU mask of relu(gn(x) + resid) as ONE BYTE per 16-byte vector (bit e = element e was positive), written by the forward and read
// by the backward INSTEAD of the residual (round 6): the backward needs the residual only to re-derive that mask -- 1/16 of its bytes.
// Same expression, same bits: the mask is the one the forward applied.
__device__ __forceinline__ uint8_t gnf_relu_bits(const f32x4& pre) {
    return (uint8_t)((pre[0] > 0.f ? 1 : 0) | (pre[1] > 0.f ? 2 : 0) | (pre[2] > 0.f ? 4 : 0) | (pre[3] > 0.f ? 8 : 0));
}
__device__ __forceinline__ f32x4 gnf_apply_bits(const f32x4& gy, uint32_t bits) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = ((bits >> e) & 1u) ? gy[e] : 0.f;
    return o;
}

template <int NT, int ACT>
__global__ __launch_bounds__(NT) void gnf_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, float* __restrict__ stats, int C, int HW, int cg, float eps,
                                                     GnfUnits un, uint8_t* __restrict__ mk) {
    constexpr int NW = NT / 64;
    __shared__ float sh[2 * NW];
    const int g = blockIdx.x % GNF_GROUPS, n = blockIdx.x / GNF_GROUPS;
    const int64_t base = ((int64_t)n * C + (int64_t)g * cg) * HW;
    const int vpc = HW >> 2;
    const float inv_n = 1.f / (float)(cg * HW);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float x0 = x[base];
    float s1 = 0.f, s2 = 0.f;
    for (int u = wave; u < un.units; u += NW) {
        const int cl = u / un.S, seg = u - cl * un.S;
        const int v0 = seg * un.segv, v1 = min(vpc, v0 + un.segv);
        const f32x4* xv = reinterpret_cast<const f32x4*>(x + base + (int64_t)cl * HW);
#define GNF_B1(v, a) { _Pragma("unroll") for (int e = 0; e < 4; ++e) { const float d = a[e] - x0; s1 += d; s2 = fmaf(d, d, s2); } }
        GNF_FOR4(xv, v0, v1, lane, GNF_LD1, GNF_B1)
#undef GNF_B1
    }
    s1 = gnf_wave_sum(s1);
    s2 = gnf_wave_sum(s2);
    if (lane == 0) { sh[wave] = s1; sh[NW + wave] = s2; }
    __syncthreads();
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { t1 += sh[w]; t2 += sh[NW + w]; }      // wave order, every thread alike
    const float m1 = t1 * inv_n, m2 = t2 * inv_n;
    const float mean = x0 + m1;
    const float rstd = rsqrtf(fmaxf(m2 - m1 * m1, 0.f) + eps);
    for (int u = wave; u < un.units; u += NW) {
        const int cl = u / un.S, seg = u - cl * un.S;
        const int v0 = seg * un.segv, v1 = min(vpc, v0 + un.segv);
        const int c = g * cg + cl;                          // wave-uniform: gamma / beta are scalar loads
        const float ga = gamma[c] * rstd;
        const float be = beta[c] - mean * ga;
        const f32x4* xv = reinterpret_cast<const f32x4*>(x + base + (int64_t)cl * HW);
        const f32x4* rv = reinterpret_cast<const f32x4*>(res + (ACT == GNF_ADD_RELU ? base + (int64_t)cl * HW : 0));
        f32x4* yv = reinterpret_cast<f32x4*>(y + base + (int64_t)cl * HW);
        uint8_t* mv = (ACT == GNF_ADD_RELU && mk) ? mk + ((base + (int64_t)cl * HW) >> 2) : nullptr;
        auto fin = [&](int v, const f32x4& a, const f32x4& r) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(a[e], ga, be);
            if (ACT == GNF_ADD_RELU) o += r;
            if (ACT == GNF_ADD_RELU && mv) mv[v] = gnf_relu_bits(o);
            if (ACT != GNF_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
            }
            GNF_ST(yv[v], o);
        };
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        int v = v0 + lane;
        for (; v + 192 < v1; v += 256) {
            const f32x4 a0 = GNF_LD2(xv[v]), a1 = GNF_LD2(xv[v + 64]), a2 = GNF_LD2(xv[v + 128]), a3 = GNF_LD2(xv[v + 192]);
            f32x4 r0 = z4, r1 = z4, r2 = z4, r3 = z4;
            if (ACT == GNF_ADD_RELU) { r0 = rv[v]; r1 = rv[v + 64]; r2 = rv[v + 128]; r3 = rv[v + 192]; }
            fin(v, a0, r0); fin(v + 64, a1, r1); fin(v + 128, a2, r2); fin(v + 192, a3, r3);
        }
        for (; v < v1; v += 64) {
            const f32x4 a0 = GNF_LD2(xv[v]);
            f32x4 r0 = z4;
            if (ACT == GNF_ADD_RELU) r0 = rv[v];
            fin(v, a0, r0);
        }
    }
    if (tid == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
}

template <int NT, int ACT>
__global__ __launch_bounds__(NT) void gnf_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ res, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ stats,
                                                     float* __restrict__ dx, float* __restrict__ dres,
                                                     float* __restrict__ dgamma_part, float* __restrict__ dbeta_part, int C, int HW, int cg,
                                                     GnfUnits un, const uint8_t* __restrict__ mk) {
    constexpr int NW = NT / 64;
    __shared__ float shu[2 * GNF_MAXU];                     // per unit: sum(dy'), sum(dy' * xhat)
    __shared__ float shq[3 * (GNF_MAXU / 2)];               // per channel: sum(dy'), sum(dy' * xhat), gamma
    __shared__ float shc[2];
    const int g = blockIdx.x % GNF_GROUPS, n = blockIdx.x / GNF_GROUPS;
    const int64_t base = ((int64_t)n * C + (int64_t)g * cg) * HW;
    const int vpc = HW >> 2;
    const float inv_n = 1.f / (float)(cg * HW);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
    // the forward's exact expression decides the ReLU mask, so it is the one forward applied
    const bool bymask = ACT == GNF_ADD_RELU && mk != nullptr;      // uniform: the forward's mask bytes instead of the residual
    auto masked = [&](const f32x4& gy, const f32x4& a, const f32x4& r, float ga, float be) {
        f32x4 o = gy;
        if (bymask) return gnf_apply_bits(gy, __float_as_uint(r[0]));
        if (ACT != GNF_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pre = fmaf(a[e], ga, be);
                if (ACT == GNF_ADD_RELU) pre += r[e];
                if (!(pre > 0.f)) o[e] = 0.f;
            }
        }
        return o;
    };
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int u = wave; u < un.units; u += NW) {
        const int cl = u / un.S, seg = u - cl * un.S;
        const int v0 = seg * un.segv, v1 = min(vpc, v0 + un.segv);
        const int c = g * cg + cl;
        const float gam = gamma[c], ga = gam * rstd, be = beta[c] - mean * ga;
        const f32x4* xv = reinterpret_cast<const f32x4*>(x + base + (int64_t)cl * HW);
        const f32x4* gv = reinterpret_cast<const f32x4*>(dy + base + (int64_t)cl * HW);
        const f32x4* rv = reinterpret_cast<const f32x4*>(res + ((ACT == GNF_ADD_RELU && !bymask) ? base + (int64_t)cl * HW : 0));
        const uint8_t* mv = bymask ? mk + ((base + (int64_t)cl * HW) >> 2) : nullptr;
        // r of a vector: the residual, or (bymask) the vector's mask byte carried in r[0]'s bits
        auto ldr = [&](int vv) { f32x4 r = {0.f, 0.f, 0.f, 0.f}; if (bymask) r[0] = __uint_as_float((uint32_t)mv[vv]); else r = rv[vv]; return r; };
        float db = 0.f, dg = 0.f;
        auto acc = [&](const f32x4& a, const f32x4& gy0, const f32x4& r) {
            const f32x4 gy = masked(gy0, a, r, ga, be);
#pragma unroll
            for (int e = 0; e < 4; ++e) { db += gy[e]; dg = fmaf(gy[e], (a[e] - mean) * rstd, dg); }
        };
        int v = v0 + lane;
        for (; v + 64 < v1; v += 128) {                     // two vectors of each of the (up to) three streams in flight
            const f32x4 a0 = xv[v], a1 = xv[v + 64], y0 = gv[v], y1 = gv[v + 64];
            f32x4 r0 = z4, r1 = z4;
            if (ACT == GNF_ADD_RELU) { r0 = ldr(v); r1 = ldr(v + 64); }
            acc(a0, y0, r0); acc(a1, y1, r1);
        }
        for (; v < v1; v += 64) {
            const f32x4 a0 = xv[v], y0 = gv[v];
            f32x4 r0 = z4;
            if (ACT == GNF_ADD_RELU) r0 = ldr(v);
            acc(a0, y0, r0);
        }
        db = gnf_wave_sum(db);
        dg = gnf_wave_sum(dg);
        if (lane == 0) { shu[2 * u] = db; shu[2 * u + 1] = dg; }
    }
    __syncthreads();
    // channel sums in segment order -- one THREAD per channel (round 6: thread 0 used to walk all cg <= 32 channels alone, a chain
    // of dependent LDS reads, scalar gamma loads and global stores that the other 1023 threads waited for) -- then the group
    // sums in channel order by one thread: the same additions in the same order as before
    if (tid < cg) {
        const int cl = tid;
        float db = 0.f, dg = 0.f;
        for (int sg = 0; sg < un.S; ++sg) { db += shu[2 * (cl * un.S + sg)]; dg += shu[2 * (cl * un.S + sg) + 1]; }
        const int c = g * cg + cl;
        dgamma_part[(int64_t)n * C + c] = dg;
        dbeta_part[(int64_t)n * C + c] = db;
        shq[3 * cl] = db; shq[3 * cl + 1] = dg; shq[3 * cl + 2] = gamma[c];
    }
    __syncthreads();
    if (tid == 0) {
        float s1 = 0.f, s2 = 0.f;
        for (int cl = 0; cl < cg; ++cl) {
            s1 = fmaf(shq[3 * cl], shq[3 * cl + 2], s1);
            s2 = fmaf(shq[3 * cl + 1], shq[3 * cl + 2], s2);
        }
        shc[0] = s1 * inv_n;
        shc[1] = s2 * inv_n;
    }
    __syncthreads();
    const float c1 = shc[0], c2 = shc[1];
    for (int u = wave; u < un.units; u += NW) {
        const int cl = u / un.S, seg = u - cl * un.S;
        const int v0 = seg * un.segv, v1 = min(vpc, v0 + un.segv);
        const int c = g * cg + cl;
        const float gam = gamma[c], ga = gam * rstd, be = beta[c] - mean * ga;
        const f32x4* xv = reinterpret_cast<const f32x4*>(x + base + (int64_t)cl * HW);
        const f32x4* gv = reinterpret_cast<const f32x4*>(dy + base + (int64_t)cl * HW);
        const f32x4* rv = reinterpret_cast<const f32x4*>(res + ((ACT == GNF_ADD_RELU && !bymask) ? base + (int64_t)cl * HW : 0));
        const uint8_t* mv = bymask ? mk + ((base + (int64_t)cl * HW) >> 2) : nullptr;
        auto ldr = [&](int vv) { f32x4 r = {0.f, 0.f, 0.f, 0.f}; if (bymask) r[0] = __uint_as_float((uint32_t)mv[vv]); else r = GNF_LD2(rv[vv]); return r; };
        f32x4* ov = reinterpret_cast<f32x4*>(dx + base + (int64_t)cl * HW);
        f32x4* dv = reinterpret_cast<f32x4*>(dres + (ACT == GNF_ADD_RELU ? base + (int64_t)cl * HW : 0));
        auto fin = [&](int vv, const f32x4& a, const f32x4& gy0, const f32x4& r) {
            const f32x4 gy = masked(gy0, a, r, ga, be);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rstd * (fmaf(gy[e], gam, -c1) - (a[e] - mean) * rstd * c2);
            GNF_ST(ov[vv], o);
            if (ACT == GNF_ADD_RELU) GNF_ST(dv[vv], gy);
        };
        int v = v0 + lane;
        for (; v + 64 < v1; v += 128) {
            const f32x4 a0 = GNF_LD2(xv[v]), a1 = GNF_LD2(xv[v + 64]), y0 = GNF_LD2(gv[v]), y1 = GNF_LD2(gv[v + 64]);
            f32x4 r0 = z4, r1 = z4;
            if (ACT == GNF_ADD_RELU) { r0 = ldr(v); r1 = ldr(v + 64); }
            fin(v, a0, y0, r0); fin(v + 64, a1, y1, r1);
        }
        for (; v < v1; v += 64) {
            const f32x4 a0 = GNF_LD2(xv[v]), y0 = GNF_LD2(gv[v]);
            f32x4 r0 = z4;
            if (ACT == GNF_ADD_RELU) r0 = ldr(v);
            fin(v, a0, y0, r0);
        }
    }
}

// ---- register-resident variant: groups of at most 8 vectors per lane (stage 2 and the small maps: 40 of the step's 52 norms) --------
// A (sample, group) of up to NT * 32 floats is read ONCE: every lane keeps its (up to) eight 16-byte vectors of x -- in the
// backward also of dy and of the residual -- in registers across the reduction, all loads of the group in flight at once (the
// streamed kernels above re-read the group for their second pass and, on these small groups, spend their time in load latency:
// 1.1-1.6 TB/s on the 28 x 28 maps).  Same units, same shifted sums; slot i of a lane = (unit wave + NW (i / ch), 64-vector chunk
// i % ch of that unit), wave-uniform; per-slot wave sums meet in LDS and are added in (channel, segment, chunk) order.
#define GNF_RV 8
// Slots of a wave, walked in order without divisions (all wave-uniform, scalar unit): slot = (unit wave + NW k, chunk j of it)
struct GnfWalk {
    int k, j, cl, seg, dq, dr;
    __device__ __forceinline__ GnfWalk(int wave, int NW, const GnfUnits& un) {
        k = 0; j = 0;
        cl = wave / un.S; seg = wave - cl * un.S;
        dq = NW / un.S; dr = NW - dq * un.S;
    }
    __device__ __forceinline__ bool valid(int wave, int NW, const GnfUnits& un, int upw) const { return k < upw && wave + k * NW < un.units; }
    __device__ __forceinline__ void next(const GnfUnits& un, int ch) {
        if (++j == ch) {
            j = 0; ++k;
            cl += dq; seg += dr;
            if (seg >= un.S) { seg -= un.S; ++cl; }
        }
    }
    // the lane's vector index inside the channel, -1 past the segment's end
    __device__ __forceinline__ int vector(int lane, const GnfUnits& un, int vpc) const {
        const int v0 = seg * un.segv, v1 = min(vpc, v0 + un.segv);
        const int v = v0 + j * 64 + lane;
        return v < v1 ? v : -1;
    }
};
#define GNF_AT(T, p, boff) (*reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<float*>(p)) + (boff)))

// the byte offset of slot (cl, vector v) of the walk inside the (sample, group); a lane past its segment re-reads a valid vector
// and ignores it.  Recomputed in every loop from the walk (scalar unit + two VALU operations) instead of being kept in a register
// per slot: with RV = 25 slots the offsets alone were 25 of the 128 registers a 1024-thread workgroup's waves may hold.
__device__ __forceinline__ uint32_t gnf_off(const GnfWalk& wk, int lane, const GnfUnits& un, int vpc, int HW, bool& ok) {
    const int v = wk.vector(lane, un, vpc);
    ok = v >= 0;
    return (uint32_t)(wk.cl * HW + 4 * max(v, 0)) * 4u;
}

template <int NT, int ACT, int RV>
__global__ __launch_bounds__(NT, 4) void gnf_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ y, float* __restrict__ stats, int C, int HW, int cg, float eps,
                                                         GnfUnits un, int upw, int ch, uint8_t* __restrict__ mk) {
    constexpr int NW = NT / 64;
    __shared__ float sh[2 * NW];
    const int g = blockIdx.x % GNF_GROUPS, n = blockIdx.x / GNF_GROUPS;
    const int64_t base = ((int64_t)n * C + (int64_t)g * cg) * HW;
    const int vpc = HW >> 2;
    const float inv_n = 1.f / (float)(cg * HW);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* xb = x + base;                             // wave-uniform bases + one 32-bit byte offset per slot
    const float* rb = res + (ACT == GNF_ADD_RELU ? base : 0);
    float* yb = y + base;
    const float x0 = xb[0];
    f32x4 xr[RV];
    uint32_t okm = 0;
    GnfWalk wk(wave, NW, un);
#pragma unroll
    for (int i = 0; i < RV; ++i, wk.next(un, ch)) {
        if (!wk.valid(wave, NW, un, upw)) break;
        bool ok;
        const uint32_t off = gnf_off(wk, lane, un, vpc, HW, ok);
        okm |= (ok ? 1u : 0u) << i;
        xr[i] = GNF_LD2(GNF_AT(const f32x4, xb, off));
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < RV; ++i) {
        if ((okm >> i) & 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = xr[i][e] - x0; s1 += d; s2 = fmaf(d, d, s2); }
        }
    }
    s1 = gnf_wave_sum(s1);
    s2 = gnf_wave_sum(s2);
    if (lane == 0) { sh[wave] = s1; sh[NW + wave] = s2; }
    __syncthreads();
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { t1 += sh[w]; t2 += sh[NW + w]; }
    const float m1 = t1 * inv_n, m2 = t2 * inv_n;
    const float mean = x0 + m1;
    const float rstd = rsqrtf(fmaxf(m2 - m1 * m1, 0.f) + eps);
    wk = GnfWalk(wave, NW, un);
#pragma unroll
    for (int i = 0; i < RV; ++i, wk.next(un, ch)) {
        if (!wk.valid(wave, NW, un, upw)) break;
        const int cl = wk.cl;
        const int c = g * cg + cl;
        const float ga = gamma[c] * rstd;
        const float be = beta[c] - mean * ga;
        bool ok;
        const uint32_t off = gnf_off(wk, lane, un, vpc, HW, ok);
        if ((okm >> i) & 1) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(xr[i][e], ga, be);
            if (ACT == GNF_ADD_RELU) o += GNF_LD2(GNF_AT(const f32x4, rb, off));
            if (ACT == GNF_ADD_RELU && mk) mk[(base >> 2) + (off >> 4)] = gnf_relu_bits(o);
            if (ACT != GNF_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
            }
            GNF_ST(GNF_AT(f32x4, yb, off), o);
        }
    }
    if (tid == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
}

template <int NT, int ACT, int RV>
__global__ __launch_bounds__(NT, 4) void gnf_bwd_reg_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ res, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ stats,
                                                         float* __restrict__ dx, float* __restrict__ dres,
                                                         float* __restrict__ dgamma_part, float* __restrict__ dbeta_part, int C, int HW,
                                                         int cg, GnfUnits un, int upw, int ch, const uint8_t* __restrict__ mk) {
    constexpr int NW = NT / 64;
    __shared__ float shp[NW * RV * 2];                  // [wave][slot]: sum(dy'), sum(dy' * xhat)
    __shared__ float shq[3 * (GNF_MAXU / 2)];           // per channel: sum(dy'), sum(dy' * xhat), gamma
    __shared__ float shc[2];
    const int g = blockIdx.x % GNF_GROUPS, n = blockIdx.x / GNF_GROUPS;
    const int64_t base = ((int64_t)n * C + (int64_t)g * cg) * HW;
    const int vpc = HW >> 2;
    const float inv_n = 1.f / (float)(cg * HW);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
    const float* xb = x + base;
    const float* gb = dy + base;
    const float* rb = res + (ACT == GNF_ADD_RELU ? base : 0);
    float* ob = dx + base;
    float* db_ = dres + (ACT == GNF_ADD_RELU ? base : 0);
    f32x4 xr[RV], gr[RV];
    uint32_t okm = 0;
    GnfWalk wk(wave, NW, un);
#pragma unroll
    for (int i = 0; i < RV; ++i, wk.next(un, ch)) {
        if (!wk.valid(wave, NW, un, upw)) break;
        bool ok;
        const uint32_t off = gnf_off(wk, lane, un, vpc, HW, ok);
        okm |= (ok ? 1u : 0u) << i;
        xr[i] = GNF_LD2(GNF_AT(const f32x4, xb, off));
        gr[i] = GNF_LD2(GNF_AT(const f32x4, gb, off));
    }
    // the forward's exact expression decides the ReLU mask; the masked gradient replaces dy in the registers
    wk = GnfWalk(wave, NW, un);
#pragma unroll
    for (int i = 0; i < RV; ++i, wk.next(un, ch)) {
        if (!wk.valid(wave, NW, un, upw)) break;
        const int cl = wk.cl;
        const int c = g * cg + cl;
        const float gam = gamma[c], ga = gam * rstd, be = beta[c] - mean * ga;
        float db = 0.f, dg = 0.f;
        if (ACT == GNF_ADD_RELU && mk != nullptr) {          // the forward's mask bytes (uniform branch): no residual read
            bool ok;
            const uint32_t off = gnf_off(wk, lane, un, vpc, HW, ok);
            gr[i] = gnf_apply_bits(gr[i], mk[(base >> 2) + (off >> 4)]);
        } else if (ACT != GNF_NONE) {
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (ACT == GNF_ADD_RELU) {
                bool ok;
                r = GNF_LD2(GNF_AT(const f32x4, rb, gnf_off(wk, lane, un, vpc, HW, ok)));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pre = fmaf(xr[i][e], ga, be);
                if (ACT == GNF_ADD_RELU) pre += r[e];
                if (!(pre > 0.f)) gr[i][e] = 0.f;
            }
        }
        if ((okm >> i) & 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { db += gr[i][e]; dg = fmaf(gr[i][e], (xr[i][e] - mean) * rstd, dg); }
        }
        db = gnf_wave_sum(db);
        dg = gnf_wave_sum(dg);
        if (lane == 0) { shp[(wave * RV + i) * 2] = db; shp[(wave * RV + i) * 2 + 1] = dg; }
    }
    __syncthreads();
    // channel sums in (segment, chunk) order, one thread per channel; group sums in channel order by one thread (see gnf_bwd_kernel)
    if (tid < cg) {
        const int cl = tid;
        float db = 0.f, dg = 0.f;
        for (int sg = 0; sg < un.S; ++sg) {
            const int u = cl * un.S + sg, w = u % NW, k = u / NW;
            for (int j = 0; j < ch; ++j) { db += shp[(w * RV + k * ch + j) * 2]; dg += shp[(w * RV + k * ch + j) * 2 + 1]; }
        }
        const int c = g * cg + cl;
        dgamma_part[(int64_t)n * C + c] = dg;
        dbeta_part[(int64_t)n * C + c] = db;
        shq[3 * cl] = db; shq[3 * cl + 1] = dg; shq[3 * cl + 2] = gamma[c];
    }
    __syncthreads();
    if (tid == 0) {
        float s1 = 0.f, s2 = 0.f;
        for (int cl = 0; cl < cg; ++cl) {
            s1 = fmaf(shq[3 * cl], shq[3 * cl + 2], s1);
            s2 = fmaf(shq[3 * cl + 1], shq[3 * cl + 2], s2);
        }
        shc[0] = s1 * inv_n;
        shc[1] = s2 * inv_n;
    }
    __syncthreads();
    const float c1 = shc[0], c2 = shc[1];
    wk = GnfWalk(wave, NW, un);
#pragma unroll
    for (int i = 0; i < RV; ++i, wk.next(un, ch)) {
        if (!wk.valid(wave, NW, un, upw)) break;
        const int cl = wk.cl;
        const float gam = gamma[g * cg + cl];
        bool ok;
        const uint32_t off = gnf_off(wk, lane, un, vpc, HW, ok);
        if ((okm >> i) & 1) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rstd * (fmaf(gr[i][e], gam, -c1) - (xr[i][e] - mean) * rstd * c2);
            GNF_ST(GNF_AT(f32x4, ob, off), o);
            if (ACT == GNF_ADD_RELU) GNF_ST(GNF_AT(f32x4, db_, off), gr[i]);
        }
    }
}

// units of a (sample, group) for NW waves: every channel cut into S segments so that there are at least NW units, segment lengths
// multiples of 64 vectors (a wave's loads stay whole KiB)
static GnfUnits gnf_units(int cg, int HW, int nwaves) {
    GnfUnits u;
    const int vpc = HW >> 2;
    u.S = cg >= nwaves ? 1 : (nwaves + cg - 1) / cg;
    u.segv = ((vpc + u.S - 1) / u.S + 63) / 64 * 64;
    u.S = (vpc + u.segv - 1) / u.segv;
    u.units = cg * u.S;
    return u;
}

// ---- small launches (CAM generation: two views of one image = 64 (sample, group) pairs on 256 CUs) ---------------------------------
// A (sample, group) is cut into P parts: gnf_part_kernel reduces each part to shifted sums (same shift x0 = the group's first
// element), gnf_apply_kernel combines the P partials in part order (deterministic) and normalises its own part.  Two launches
// of N * 32 * P workgroups instead of one of N * 32: 27 -> ~10 us for the 1024-channel maps of a 384^2 image.
__global__ __launch_bounds__(256) void gnf_part_kernel(const float* __restrict__ x, float* __restrict__ parts, int C, int HW, int cg, int P, int vper) {
    __shared__ float sh[4];
    const int part = blockIdx.x % P, ng = blockIdx.x / P;
    const int g = ng % GNF_GROUPS, n = ng / GNF_GROUPS;
    const int64_t base = ((int64_t)n * C + (int64_t)g * cg) * HW;
    const int nvec = (cg * HW) >> 2;
    const int v0 = part * vper, v1 = min(nvec, v0 + vper);
    const f32x4* xv = reinterpret_cast<const f32x4*>(x + base);
    const float x0 = x[base];
    float s1 = 0.f, s2 = 0.f;
    for (int v = v0 + threadIdx.x; v < v1; v += 256) {
        const f32x4 a = xv[v];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = a[e] - x0; s1 += d; s2 = fmaf(d, d, s2); }
    }
    s1 = gnf_block_sum<256>(s1, sh);
    s2 = gnf_block_sum<256>(s2, sh);
    if (threadIdx.x == 0) { parts[2 * blockIdx.x] = s1; parts[2 * blockIdx.x + 1] = s2; }
}
template <int ACT>
__global__ __launch_bounds__(256) void gnf_apply_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ parts, float* __restrict__ y,
                                                        float* __restrict__ stats, int C, int HW, int cg, int P, int vper, float eps) {
    const int part = blockIdx.x % P, ng = blockIdx.x / P;
    const int g = ng % GNF_GROUPS, n = ng / GNF_GROUPS;
    const int64_t base = ((int64_t)n * C + (int64_t)g * cg) * HW;
    const int nvec = (cg * HW) >> 2, vpc = HW >> 2;
    const float inv_n = 1.f / (float)(cg * HW);
    float s1 = 0.f, s2 = 0.f;
    for (int p = 0; p < P; ++p) { s1 += parts[2 * (ng * P + p)]; s2 += parts[2 * (ng * P + p) + 1]; }      // part order, every thread alike
    const float x0 = x[base];
    const float m1 = s1 * inv_n, m2 = s2 * inv_n;
    const float mean = x0 + m1;
    const float rstd = rsqrtf(fmaxf(m2 - m1 * m1, 0.f) + eps);
    const f32x4* xv = reinterpret_cast<const f32x4*>(x + base);
    const f32x4* rv = reinterpret_cast<const f32x4*>(res + (ACT == GNF_ADD_RELU ? base : 0));
    f32x4* yv = reinterpret_cast<f32x4*>(y + base);
    const int v0 = part * vper, v1 = min(nvec, v0 + vper);
    for (int v = v0 + threadIdx.x; v < v1; v += 256) {
        const int c = g * cg + v / vpc;
        const float ga = gamma[c] * rstd;
        const float be = beta[c] - mean * ga;
        const f32x4 a = xv[v];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(a[e], ga, be);
        if (ACT == GNF_ADD_RELU) o += rv[v];
        if (ACT != GNF_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        yv[v] = o;
    }
    if (part == 0 && threadIdx.x == 0) { stats[2 * ng] = mean; stats[2 * ng + 1] = rstd; }
}
static int gnf_fwd_parts(int N, int C, int HW) {
    const int groups = N * GNF_GROUPS, nvec = (C / GNF_GROUPS) * HW / 4;
    if (groups > 128 || nvec < 4096) return 1;
    int P = 512 / groups;
    if (P > 8) P = 8;
    while (P > 1 && nvec / P < 1024) --P;                    // at least 4 vectors per thread and part
    return P;
}
extern "C" size_t acr_groupnorm_fwd_ws_floats(int32_t N, int32_t C, int32_t HW) {
    if (N <= 0 || C <= 0 || HW <= 0 || (C % GNF_GROUPS) != 0) return 0;
    const int P = gnf_fwd_parts(N, C, HW);
    return P > 1 ? (size_t)N * GNF_GROUPS * P * 2 : 0;
}

// dgamma[c] = sum_n part[n][c] in sample order (8 interleaved chains combined in fixed order), fp32 out
__global__ __launch_bounds__(256) void gnf_param_reduce_kernel(const float* __restrict__ gpart, const float* __restrict__ bpart, int N,
                                                               int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float sh[8][33];
    const int cl = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;                      // over 2*C columns
    float s = 0.f;
    if (i < 2 * C) {
        const float* src = (i < C) ? gpart + i : bpart + (i - C);
        for (int n = sg; n < N; n += 8) s += src[(int64_t)n * C];
    }
    sh[sg][cl] = s;
    __syncthreads();
    if (sg == 0 && i < 2 * C) {
        float t = sh[0][cl];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sh[k][cl];
        if (i < C) dgamma[i] = t;
        else dbeta[i - C] = t;
    }
}

static int gnf_check(const char* who, int N, int C, int HW, int act) {
    ACR_CHECK_ARG(N > 0 && C > 0 && (C % GNF_GROUPS) == 0, "%s: C=%d must be a multiple of 32", who, C);
    ACR_CHECK_ARG(HW > 0 && (HW % 4) == 0, "%s: H*W=%d must be a multiple of 4 (16-byte vectors per channel)", who, HW);
    ACR_CHECK_ARG(act >= 0 && act <= 2, "%s: unknown act %d", who, act);
    ACR_CHECK_ARG(C / GNF_GROUPS <= GNF_MAXU / 2, "%s: C=%d: at most %d channels per group", who, C, GNF_MAXU / 2);
    return ACR_OK;
}

// register-resident plan: threads per workgroup, units, units per wave, chunks per unit; slots = vectors every lane holds.
// Round 6: the kernels are templated on the slot count.  A 1024-thread workgroup's waves may hold 128 registers each (four waves
// per SIMD), i.e. up to 25 vectors of ONE operand: the forward keeps ALL of the step's groups resident (the 401 KB groups of the
// 112^2 maps and of the stem norm: 25 slots -- one read + one write instead of two reads + one write), the backward, which holds
// x and dy, groups of up to 13 slots (128 channels at 112^2, 512 at 56^2: 2 (3) reads instead of 4 (6)).
#define GNF_RV_FWD_MAX 25
#define GNF_RV_BWD_MAX 13
struct GnfRegPlan { bool ok; int nt; GnfUnits un; int upw, ch, slots; };
static GnfRegPlan gnf_reg_plan_nt(int cg, int HW, int nt) {
    GnfRegPlan p;
    p.nt = nt;
    const int nw = p.nt / 64;
    p.un = gnf_units(cg, HW, nw);
    p.ch = (p.un.segv + 63) / 64;
    p.upw = (p.un.units + nw - 1) / nw;
    p.slots = p.upw * p.ch;
    p.ok = p.un.units <= GNF_MAXU;
    return p;
}
// Workgroup size of the register-resident kernels (ACR_OPT_GN_PLAN).  The register footprint of a group on a CU is the same
// whatever the workgroup size; what changes is how many INDEPENDENT workgroups share the CU -- a lone 1024-thread workgroup
// cannot overlap its load phase with another one's store phase.
static GnfRegPlan gnf_reg_plan(int cg, int HW, int max_slots, bool backward) {
    const int nvec = cg * (HW >> 2);
    int pref = acr_opt(ACR_OPT_GN_PLAN);
    if (pref == 3) pref = backward ? 2 : 0;
    static const int order[3] = {256, 512, 1024};
    GnfRegPlan p;
    if (pref == 0) {
        p = gnf_reg_plan_nt(cg, HW, nvec > 2048 ? 1024 : 256);
        p.ok = p.ok && p.slots <= max_slots && (p.slots <= GNF_RV || p.nt == 1024);
        return p;
    }
    for (int i = 0; i < 3; ++i) {
        p = gnf_reg_plan_nt(cg, HW, order[i]);
        const int cap = (pref == 1 && order[i] == 256) ? GNF_RV : max_slots;      // 1: small workgroups only for small groups
        if (p.ok && p.slots <= cap) return p;
    }
    p.ok = false;
    return p;
}
#define GNF_LAUNCH_REG(KERNEL, NT_, RV_, P, ...)                                                                                     \
    {                                                                                                                                \
        if (act == 0) hipLaunchKernelGGL((KERNEL<NT_, 0, RV_>), grid, dim3(NT_), 0, st, __VA_ARGS__, (P).un, (P).upw, (P).ch, relu_mask);       \
        else if (act == 1) hipLaunchKernelGGL((KERNEL<NT_, 1, RV_>), grid, dim3(NT_), 0, st, __VA_ARGS__, (P).un, (P).upw, (P).ch, relu_mask);  \
        else hipLaunchKernelGGL((KERNEL<NT_, 2, RV_>), grid, dim3(NT_), 0, st, __VA_ARGS__, (P).un, (P).upw, (P).ch, relu_mask);                \
    }
#define GNF_LAUNCH_REG_NT(KERNEL, RV_, P, ...)                                                     \
    if ((P).nt == 256) GNF_LAUNCH_REG(KERNEL, 256, RV_, P, __VA_ARGS__)                            \
    else if ((P).nt == 512) GNF_LAUNCH_REG(KERNEL, 512, RV_, P, __VA_ARGS__)                       \
    else GNF_LAUNCH_REG(KERNEL, 1024, RV_, P, __VA_ARGS__)
#define GNF_DISPATCH_REG_FWD(KERNEL, P, ...)                                                       \
    if ((P).slots <= GNF_RV) { GNF_LAUNCH_REG_NT(KERNEL, GNF_RV, P, __VA_ARGS__) }                 \
    else if ((P).slots <= GNF_RV_BWD_MAX) { GNF_LAUNCH_REG_NT(KERNEL, GNF_RV_BWD_MAX, P, __VA_ARGS__) } \
    else { GNF_LAUNCH_REG_NT(KERNEL, GNF_RV_FWD_MAX, P, __VA_ARGS__) }
#define GNF_DISPATCH_REG_BWD(KERNEL, P, ...)                                                       \
    if ((P).slots <= GNF_RV) { GNF_LAUNCH_REG_NT(KERNEL, GNF_RV, P, __VA_ARGS__) }                 \
    else { GNF_LAUNCH_REG_NT(KERNEL, GNF_RV_BWD_MAX, P, __VA_ARGS__) }

#define GNF_DISPATCH(KERNEL, ...)                                                                                \
    if (big) {                                                                                                    \
        const GnfUnits un = gnf_units(cg, HW, 16);                                                                \
        if (act == 0) hipLaunchKernelGGL((KERNEL<1024, 0>), grid, dim3(1024), 0, st, __VA_ARGS__, un, relu_mask);           \
        else if (act == 1) hipLaunchKernelGGL((KERNEL<1024, 1>), grid, dim3(1024), 0, st, __VA_ARGS__, un, relu_mask);      \
        else hipLaunchKernelGGL((KERNEL<1024, 2>), grid, dim3(1024), 0, st, __VA_ARGS__, un, relu_mask);                    \
    } else {                                                                                                      \
        const GnfUnits un = gnf_units(cg, HW, 4);                                                                 \
        if (act == 0) hipLaunchKernelGGL((KERNEL<256, 0>), grid, dim3(256), 0, st, __VA_ARGS__, un, relu_mask);             \
        else if (act == 1) hipLaunchKernelGGL((KERNEL<256, 1>), grid, dim3(256), 0, st, __VA_ARGS__, un, relu_mask);        \
        else hipLaunchKernelGGL((KERNEL<256, 2>), grid, dim3(256), 0, st, __VA_ARGS__, un, relu_mask);                      \
    }

static int gnf_fwd_impl(const float* x, const float* resid, const float* gamma, const float* beta, float* y, float* stats,
                        int32_t N, int32_t C, int32_t HW, float eps, int32_t act, float* ws, uint8_t* relu_mask, void* stream) {
    ACR_CHECK_ARG(x && gamma && beta && y && stats && (act != GNF_ADD_RELU || resid), "acr_groupnorm_fwd_f32: null pointer");
    ACR_CHECK_ARG(!relu_mask || (act == GNF_ADD_RELU && !ws), "acr_groupnorm_fwd_mask_f32: the mask exists for act = 2 (residual + ReLU) without the small-launch workspace");
    int rc = gnf_check("acr_groupnorm_fwd_f32", N, C, HW, act);
    if (rc) return rc;
    ACR_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)resid & 15) == 0, "acr_groupnorm_fwd_f32: 16-byte alignment");
    const int cg = C / GNF_GROUPS;
    const bool big = (int64_t)cg * HW >= 32768;              // >= 8 vectors per thread at 1024 threads
    const dim3 grid(N * GNF_GROUPS);
    hipStream_t st = (hipStream_t)stream;
    const int P = ws ? gnf_fwd_parts(N, C, HW) : 1;
    if (P > 1) {                                            // small launch: every (sample, group) cut into P parts, two launches
        const int nvec = cg * HW / 4, vper = (nvec + P - 1) / P;
        const dim3 pgrid(N * GNF_GROUPS * P);
        hipLaunchKernelGGL(gnf_part_kernel, pgrid, dim3(256), 0, st, x, ws, C, HW, cg, P, vper);
        if (act == 0) hipLaunchKernelGGL((gnf_apply_kernel<0>), pgrid, dim3(256), 0, st, x, resid, gamma, beta, (const float*)ws, y, stats, C, HW, cg, P, vper, eps);
        else if (act == 1) hipLaunchKernelGGL((gnf_apply_kernel<1>), pgrid, dim3(256), 0, st, x, resid, gamma, beta, (const float*)ws, y, stats, C, HW, cg, P, vper, eps);
        else hipLaunchKernelGGL((gnf_apply_kernel<2>), pgrid, dim3(256), 0, st, x, resid, gamma, beta, (const float*)ws, y, stats, C, HW, cg, P, vper, eps);
        return acr_check_launch("acr_groupnorm_fwd_f32(parts)");
    }
    const GnfRegPlan rp = gnf_reg_plan(cg, HW, GNF_RV_FWD_MAX, false);
    if (rp.ok) {                                            // the group fits the workgroup's registers: one read
        GNF_DISPATCH_REG_FWD(gnf_fwd_reg_kernel, rp, x, resid, gamma, beta, y, stats, C, HW, cg, eps)
        return acr_check_launch("acr_groupnorm_fwd_f32(reg)");
    }
    GNF_DISPATCH(gnf_fwd_kernel, x, resid, gamma, beta, y, stats, C, HW, cg, eps)
    return acr_check_launch("acr_groupnorm_fwd_f32");
}

extern "C" int acr_groupnorm_fwd_f32(const float* x, const float* resid, const float* gamma, const float* beta, float* y, float* stats,
                                     int32_t N, int32_t C, int32_t HW, float eps, int32_t act, float* ws, void* stream) {
    return gnf_fwd_impl(x, resid, gamma, beta, y, stats, N, C, HW, eps, act, ws, nullptr, stream);
}
// act = 2 (relu(gn(x) + resid)) that also writes the ReLU mask, one byte per 16-byte vector of y (N*C*HW/4 bytes): the backward
// (acr_groupnorm_bwd_mask_f32) reads it instead of the residual.
extern "C" int acr_groupnorm_fwd_mask_f32(const float* x, const float* resid, const float* gamma, const float* beta, float* y, float* stats,
                                          int32_t N, int32_t C, int32_t HW, float eps, uint8_t* relu_mask, void* stream) {
    ACR_CHECK_ARG(relu_mask, "acr_groupnorm_fwd_mask_f32: null mask");
    return gnf_fwd_impl(x, resid, gamma, beta, y, stats, N, C, HW, eps, GNF_ADD_RELU, nullptr, relu_mask, stream);
}

static int gnf_bwd_impl(const float* dy, const float* x, const float* resid, const float* gamma, const float* beta,
                        const float* stats, float* dx, float* dresid, float* dgamma_part, float* dbeta_part,
                        float* dgamma, float* dbeta, int32_t N, int32_t C, int32_t HW, int32_t act, const uint8_t* relu_mask, void* stream) {
    ACR_CHECK_ARG(dy && x && gamma && beta && stats && dx && dgamma_part && dbeta_part && (act != GNF_ADD_RELU || ((resid || relu_mask) && dresid)),
                  "acr_groupnorm_bwd_f32: null pointer");
    int rc = gnf_check("acr_groupnorm_bwd_f32", N, C, HW, act);
    if (rc) return rc;
    ACR_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)resid & 15) == 0 &&
                      ((uintptr_t)dresid & 15) == 0, "acr_groupnorm_bwd_f32: 16-byte alignment");
    const int cg = C / GNF_GROUPS;
    const bool big = (int64_t)HW >= 4096;                    // per-channel loops: 1024 threads only when a channel feeds them
    const dim3 grid(N * GNF_GROUPS);
    hipStream_t st = (hipStream_t)stream;
    const GnfRegPlan rp = gnf_reg_plan(cg, HW, GNF_RV_BWD_MAX, true);
    if (rp.ok) {
        GNF_DISPATCH_REG_BWD(gnf_bwd_reg_kernel, rp, dy, x, resid, gamma, beta, stats, dx, dresid, dgamma_part, dbeta_part, C, HW, cg)
    } else {
        GNF_DISPATCH(gnf_bwd_kernel, dy, x, resid, gamma, beta, stats, dx, dresid, dgamma_part, dbeta_part, C, HW, cg)
    }
    if (dgamma && dbeta)
        hipLaunchKernelGGL(gnf_param_reduce_kernel, dim3((2 * C + 31) / 32), dim3(256), 0, st, (const float*)dgamma_part,
                           (const float*)dbeta_part, N, C, dgamma, dbeta);
    return acr_check_launch("acr_groupnorm_bwd_f32");
}
extern "C" int acr_groupnorm_bwd_f32(const float* dy, const float* x, const float* resid, const float* gamma, const float* beta,
                                     const float* stats, float* dx, float* dresid, float* dgamma_part, float* dbeta_part,
                                     float* dgamma, float* dbeta, int32_t N, int32_t C, int32_t HW, int32_t act, void* stream) {
    return gnf_bwd_impl(dy, x, resid, gamma, beta, stats, dx, dresid, dgamma_part, dbeta_part, dgamma, dbeta, N, C, HW, act, nullptr, stream);
}
// backward of act = 2 from the forward's mask bytes: the residual is not read (6 -> 5 resp. 3 -> 2 operand reads + 1/16)
extern "C" int acr_groupnorm_bwd_mask_f32(const float* dy, const float* x, const uint8_t* relu_mask, const float* gamma, const float* beta,
                                          const float* stats, float* dx, float* dresid, float* dgamma_part, float* dbeta_part,
                                          float* dgamma, float* dbeta, int32_t N, int32_t C, int32_t HW, void* stream) {
    ACR_CHECK_ARG(relu_mask, "acr_groupnorm_bwd_mask_f32: null mask");
    return gnf_bwd_impl(dy, x, nullptr, gamma, beta, stats, dx, dresid, dgamma_part, dbeta_part, dgamma, dbeta, N, C, HW, GNF_ADD_RELU, relu_mask,
                        stream);
}
