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
#include "attn_f32_sres_tails.h"

typedef __bf16 bf16_t;
typedef __attribute__((address_space(3))) void* x3_lds_vp;
typedef const __attribute__((address_space(1))) void* x3_glb_vp;

#define X3_PLANE_B 4096                    // bytes of one plane image of a 32-row tile
#define X3_TILE_B (3 * X3_PLANE_B)         // one operand tile: three planes
#define X3_SLOT_B (2 * X3_TILE_B)          // one ring slot: two operand tiles
#define X3_SB_FLOATS 1024                  // one 32 x 32 score block (layout: attn_f32_sres.hip)

#define X3_STORE_NT(p, v) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p))
#define X3_LOAD_NT(p) __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p))


// ---- counted waits ------------------------------------------------------------------------------------------------------------
// A step's tile DMA must have landed at the step's barrier, but the streams that run further ahead (score blocks two steps
// ahead, G one step, the forward's score stores) are YOUNGER vector-memory operations and may stay in flight: vmcnt retires in
// issue order, so "at most n outstanding" with n = the number of younger operations is exactly "the tile has landed".  n is
// wave-uniform; a count that is not listed waits for the next smaller one (stricter, never wrong).
template <int N>
__device__ __forceinline__ void x3_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void x3_wait_vm(int n) {
    if (n >= 9) x3_vmcnt<9>();
    else if (n >= 8) x3_vmcnt<8>();
    else if (n >= 7) x3_vmcnt<7>();
    else if (n >= 5) x3_vmcnt<5>();
    else if (n >= 4) x3_vmcnt<4>();
    else if (n >= 1) x3_vmcnt<1>();
    else x3_vmcnt<0>();
}
// compiler-only fence: vector-memory operations written after it are issued after the ones before it (the counts above rely
// on the issue order; loads from global memory and LDS-DMA writes do not alias, so nothing else orders them for the compiler)
#define X3_FENCE() asm volatile("" ::: "memory")
// acr_barrier_nofence (acr_common.h): __syncthreads()'s release fence made hipcc drain every outstanding DMA in front of the
// barrier of every second step (`s_waitcnt vmcnt(0)`, found in the ISA) -- the counted wait above it is the synchronisation
__device__ __forceinline__ void x3_barrier(int n_younger) {
    x3_wait_vm(n_younger);
    acr_barrier_nofence();
}

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

// ---- fragment addresses (LDS byte addresses of plane 0 of the tile at ring offset 0; lane-dependent part, computed once) -------
struct X3Lane { uint32_t rowb[4]; uint32_t trb[2][2]; };
__device__ __forceinline__ uint32_t x3_lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ X3Lane x3_lane(int lane, const char* smem) {
    X3Lane lb;
    const uint32_t lds0 = x3_lds_addr(smem);
    const int r = lane & 31, h = lane >> 5;
    const int fr = x3_swz(r);
#pragma unroll
    for (int s = 0; s < 4; ++s) lb.rowb[s] = lds0 + r * 128 + (((2 * s + h) ^ fr) << 4);
    const int i = lane & 15, g1 = (lane >> 4) & 1, q = i >> 2, p = i & 3;
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
        const int row = 8 * hi + 4 * h + q;                // + 16 s rows per k-step: f is unchanged by it
        const int f = x3_swz(row);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) lb.trb[hi][blk] = lds0 + row * 128 + (((4 * blk + 2 * g1 + (p >> 1)) ^ f) << 4) + 8 * (p & 1);
    }
    return lb;
}
// Every LDS read that follows an LDS-DMA in program order is INLINE ASM with its own counted lgkmcnt wait: hipcc cannot tell a
// DMA's LDS write from a read of another slot and puts s_waitcnt vmcnt(0) in front of every LDS load builtin behind one --
// which made each step wait for the tiles and score blocks it had just sent for (found in the first version's ISA; the same
// finding as gemm_bf16.hip).  LDS operations of a wave return in order, so "at most n outstanding" = "all but my n newest
// reads have landed"; operations the compiler adds around them only make a wait stricter.  Outputs are early-clobber and the
// waits name the registers that must have landed.
#define X3_RD128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "i"(OFF))
#define X3_RD32(dst, addr, OFF) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "i"(OFF))
#define X3_RDTR(lo, hi, alo, ahi, OFF)                                                                  \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"          \
                 : "=&v"(lo), "=&v"(hi) : "v"(alo), "v"(ahi), "i"(OFF))
#define X3_WAIT3(cnt, x) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]))
#define X3_WAIT6(cnt, l, h_) \
    asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(h_[0]), "+v"(h_[1]), "+v"(h_[2]))
// rows of the tile as an MFMA operand: element j of lane (r, h) = tile[r][16 S + 8 h + j], planes 0..2
#define X3_ROWFRAG(a, lb, TOFF, S) \
    { X3_RD128(a[0], lb.rowb[S], TOFF); X3_RD128(a[1], lb.rowb[S], (TOFF) + X3_PLANE_B); X3_RD128(a[2], lb.rowb[S], (TOFF) + 2 * X3_PLANE_B); }
// the tile transposed: element j of lane (r, h) = tile[16 S + 8 (j >> 2) + 4 h + (j & 3)][32 BLK + r] (lo: j < 4, hi: j >= 4)
#define X3_TRFRAG(l, h_, lb, TOFF, S, BLK)                                                      \
    { X3_RDTR(l[0], h_[0], lb.trb[0][BLK], lb.trb[1][BLK], (TOFF) + (S) * 2048);                 \
      X3_RDTR(l[1], h_[1], lb.trb[0][BLK], lb.trb[1][BLK], (TOFF) + X3_PLANE_B + (S) * 2048);    \
      X3_RDTR(l[2], h_[2], lb.trb[0][BLK], lb.trb[1][BLK], (TOFF) + 2 * X3_PLANE_B + (S) * 2048); }
#define X3_JOIN(a, l, h_)                                                                       \
    const bf16x8 a[3] = {__builtin_shufflevector(l[0], h_[0], 0, 1, 2, 3, 4, 5, 6, 7),           \
                         __builtin_shufflevector(l[1], h_[1], 0, 1, 2, 3, 4, 5, 6, 7),           \
                         __builtin_shufflevector(l[2], h_[2], 0, 1, 2, 3, 4, 5, 6, 7)}

// FILLERS.  Both waves of a SIMD run the same program from the same barrier: left alone they reach their matrix work together
// and their vector work together (first version: 56-59 % of a step's cycles were one wave waiting for its partner at the
// barrier, scripts/lab/attn_x3_phases.py).  A six-MFMA group keeps the matrix pipe for ~190 cycles; the operations below take
// callables f0..f2 that run BETWEEN the groups -- the step's DMA issue and the operand splits go there -- so that a wave's
// stream alternates matrix and vector work in ~200-cycle pieces and the partner's groups fall into the gaps.
struct X3Nop { __device__ __forceinline__ void operator()() const {} };

// acc[krow(reg,h)][r] += sum_d tile[krow][d] * y[r][d]: tile rows = A operand, y = the lane's row held as fragments y[plane][k-step].
// The reads of k-step S + 1 are in flight while the six MFMAs of k-step S run.
template <int TILE_OFF, class F0 = X3Nop, class F1 = X3Nop, class F2 = X3Nop>
__device__ __forceinline__ void x3_rowop(f32x16& acc, const X3Lane& lb, const bf16x8 (&y)[3][4], F0&& f0 = F0(), F1&& f1 = F1(),
                                         F2&& f2 = F2()) {
    bf16x8 a0[3], a1[3];
    X3_ROWFRAG(a0, lb, TILE_OFF, 0);
    X3_ROWFRAG(a1, lb, TILE_OFF, 1);
    X3_WAIT3(3, a0);
    { const bf16x8 b_[3] = {y[0][0], y[1][0], y[2][0]}; X3_MFMA6(acc, a0, b_); }
    f0();
    X3_ROWFRAG(a0, lb, TILE_OFF, 2);
    X3_WAIT3(3, a1);
    { const bf16x8 b_[3] = {y[0][1], y[1][1], y[2][1]}; X3_MFMA6(acc, a1, b_); }
    f1();
    X3_ROWFRAG(a1, lb, TILE_OFF, 3);
    X3_WAIT3(3, a0);
    { const bf16x8 b_[3] = {y[0][2], y[1][2], y[2][2]}; X3_MFMA6(acc, a0, b_); }
    f2();
    X3_WAIT3(0, a1);
    { const bf16x8 b_[3] = {y[0][3], y[1][3], y[2][3]}; X3_MFMA6(acc, a1, b_); }
}
// The accumulator tile z (pieces z0: rows 0-15, z1: rows 16-31) times the transposed tile, both 32-column blocks:
//   ZA = true :  acc_blk[i = z-lane][j = tile column 32 blk + r]      (z is the A operand)
//   ZA = false:  acc_blk[i = tile column 32 blk + krow][j = z-lane]   (z is the B operand)
// Group order (k-step, block): (0,0) (0,1) (1,0) (1,1) -- z1 is first read after f1, so f0 / f1 may still be producing it.
template <int TILE_OFF, bool ZA, class F0 = X3Nop, class F1 = X3Nop, class F2 = X3Nop>
__device__ __forceinline__ void x3_accop(f32x16& acc0, f32x16& acc1, const bf16x8 (&z0)[3], const bf16x8 (&z1)[3], const X3Lane& lb,
                                         F0&& f0 = F0(), F1&& f1 = F1(), F2&& f2 = F2()) {
    bf16x4 l0[3], h0[3], l1[3], h1[3];
    X3_TRFRAG(l0, h0, lb, TILE_OFF, 0, 0);
    X3_TRFRAG(l1, h1, lb, TILE_OFF, 0, 1);
    X3_WAIT6(6, l0, h0);
    { X3_JOIN(t, l0, h0); if (ZA) { X3_MFMA6(acc0, z0, t); } else { X3_MFMA6(acc0, t, z0); } }
    f0();
    X3_TRFRAG(l0, h0, lb, TILE_OFF, 1, 0);
    X3_WAIT6(6, l1, h1);
    { X3_JOIN(t, l1, h1); if (ZA) { X3_MFMA6(acc1, z0, t); } else { X3_MFMA6(acc1, t, z0); } }
    f1();
    X3_TRFRAG(l1, h1, lb, TILE_OFF, 1, 1);
    X3_WAIT6(6, l0, h0);
    { X3_JOIN(t, l0, h0); if (ZA) { X3_MFMA6(acc0, z1, t); } else { X3_MFMA6(acc0, t, z1); } }
    f2();
    X3_WAIT6(0, l1, h1);
    { X3_JOIN(t, l1, h1); if (ZA) { X3_MFMA6(acc1, z1, t); } else { X3_MFMA6(acc1, t, z1); } }
}
// the same with ONE fragment set (12 registers less): a group's reads are issued behind the previous group's filler, their
// latency is covered by the SIMD partner, not by this wave's own MFMAs (dK/dV body: 256 registers are all there is)
template <int TILE_OFF, bool ZA, class F0 = X3Nop, class F1 = X3Nop, class F2 = X3Nop>
__device__ __forceinline__ void x3_accop1(f32x16& acc0, f32x16& acc1, const bf16x8 (&z0)[3], const bf16x8 (&z1)[3], const X3Lane& lb,
                                          F0&& f0 = F0(), F1&& f1 = F1(), F2&& f2 = F2()) {
    bf16x4 l0[3], h0[3];
    X3_TRFRAG(l0, h0, lb, TILE_OFF, 0, 0);
    X3_WAIT6(0, l0, h0);
    { X3_JOIN(t, l0, h0); if (ZA) { X3_MFMA6(acc0, z0, t); } else { X3_MFMA6(acc0, t, z0); } }
    X3_TRFRAG(l0, h0, lb, TILE_OFF, 0, 1);
    f0();
    X3_WAIT6(0, l0, h0);
    { X3_JOIN(t, l0, h0); if (ZA) { X3_MFMA6(acc1, z0, t); } else { X3_MFMA6(acc1, t, z0); } }
    X3_TRFRAG(l0, h0, lb, TILE_OFF, 1, 0);
    f1();
    X3_WAIT6(0, l0, h0);
    { X3_JOIN(t, l0, h0); if (ZA) { X3_MFMA6(acc0, z1, t); } else { X3_MFMA6(acc0, t, z1); } }
    X3_TRFRAG(l0, h0, lb, TILE_OFF, 1, 1);
    f2();
    X3_WAIT6(0, l0, h0);
    { X3_JOIN(t, l0, h0); if (ZA) { X3_MFMA6(acc1, z1, t); } else { X3_MFMA6(acc1, t, z1); } }
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
                                                             float* __restrict__ sres, const float* __restrict__ q32,
                                                             const float* __restrict__ k32, const float* __restrict__ v32,
                                                             AttnGeom g32, int ntail, char* __restrict__ oimg) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * X3_SLOT_B];        // [slot][K planes | V planes]
    __shared__ float mlsh[4 * 64];                                             // split tail: (m | l) of the four partial sweeps
    // The last `ntail` workgroups of the grid (dispatched after every full one) are split-tail workgroups, one per (b, h): T = 785
    // is 25 blocks = 6 x 4 + ONE, which as a seventh workgroup with a single live wave costs a full workgroup's time.  They are
    // the resident-score generation's (attn_f32_sres_tails.h): exact-fp32 products from the fp32 operands on 1/NB of the rows.
    const int nmain = (int)gridDim.x - ntail;
    if ((int)blockIdx.x >= nmain) {
        attn_fwd_tail_body<4>(reinterpret_cast<float*>(smem), mlsh, g32, q32, k32, v32, o, lse2, sres, acr_xcd_remap((int)blockIdx.x - nmain, ntail));
        return;
    }
    const int NB = (g.T + 31) >> 5, nqt = ntail ? NB >> 2 : (NB + 3) >> 2;
    int id = acr_xcd_remap(blockIdx.x, nmain);
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
    const X3Lane lb = x3_lane(lane, smem);
    const int doff = x3_dma_off(g.D, wave, lane);
    const float c2 = g.scale * ACR_LOG2E;
    float* sblk = sres + x3_block(g.H, NB, b, hd, min(q0 >> 5, NB - 1), 0) + lane * 4;
    auto step = [&](int k0, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int KOFF = SLOT * X3_SLOT_B, VOFF = KOFF + X3_TILE_B;
        x3_barrier((k0 > 0 && live) ? 4 : 0);              // slot SLOT has landed (the previous step's 4 score stores may be in flight); the other slot is free
        // the next tile's DMA (K planes, then V planes) is issued between the MFMA groups of this tile's first product
        auto dma_next = [&](const bf16_t* base, int toff) {
            if (k0 + 64 <= g.T) x3_dma_tile_i(smem + (SLOT ^ 1) * X3_SLOT_B + toff, base + (int64_t)(k0 + 32) * g.D, g.plane, doff, wave);
            else if (k0 + 32 < g.T) x3_dma_tile(smem + (SLOT ^ 1) * X3_SLOT_B + toff, base, g.plane, g.D, k0 + 32, g.T, wave, lane);
            X3_FENCE();
        };
        if (!live) {                                       // waves past the end only help with the DMA
            dma_next(kb, 0);
            dma_next(vb, X3_TILE_B);
            return;
        }
        f32x16 s = {0};
        x3_rowop<KOFF>(s, lb, qf, [&] { dma_next(kb, 0); }, [&] { dma_next(vb, X3_TILE_B); });      // s[reg] = q.k of key k0 + krow, query q0 + r
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) s[reg] *= c2;   // scaled base-2 logits
        if (k0 + 32 > g.T) {                               // only the last key tile has keys beyond T (uniform branch)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (k0 + acr_krow(reg, h) >= g.T) s[reg] = -INFINITY;
        }
        float* sp = sblk + (int64_t)(k0 >> 5) * X3_SB_FLOATS;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {                   // (kept out of the second product's fillers: 16 more live registers
            const f32x4 t = {s[4 * gq], s[4 * gq + 1], s[4 * gq + 2], s[4 * gq + 3]};      //  cost the third wave per SIMD)
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
        x3_accop<VOFF, false>(o0, o1, p0, p1, lb, [&] { x3_split_acc<1>(p, p1); });      // o[reg] = O^T[d = 32*blk + krow][query = r]
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
        // The output as the NEXT product's split-product image (proj reads o as [token][feature] rows; include/acr_hip.h "split-product
        // images"): row = b T + query of the (B T) x D matrix, this head's 64 features = 8 chunks of 8.  Lane (r, h) holds features
        // 8 grp + 4 h .. + 3 of each quad: the two halves of a chunk sit in lanes (r, 0) and (r, 1), so the pair swaps one quad per two
        // chunks and lane h finishes the chunks of parity h.  Same split expression as the image pass (planes_split8): same bits.
        if (oimg != nullptr) {
            const int64_t row = (int64_t)b * g.T + q0 + r;
            const int nkb = g.D >> 4;
            char* ib = oimg + ((row >> 7) * nkb + hd * 4) * (int64_t)X3_TILE_B + (int)(row & 127) * 32;
            const int sw = (int)((row >> 3) & 1);
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float own[4], snd[4], rcv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = (half ? o1 : o0)[4 * (2 * j) + e] * inv, hi = (half ? o1 : o0)[4 * (2 * j + 1) + e] * inv;
                        own[e] = h ? hi : lo;                   // lane 0 keeps quad 2j (chunk 2j, low half); lane 1 quad 2j+1 (chunk 2j+1, high half)
                        snd[e] = h ? lo : hi;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) rcv[e] = __shfl_xor(snd[e], 32);
                    const float v8[8] = {h ? rcv[0] : own[0], h ? rcv[1] : own[1], h ? rcv[2] : own[2], h ? rcv[3] : own[3],
                                         h ? own[0] : rcv[0], h ? own[1] : rcv[1], h ? own[2] : rcv[2], h ? own[3] : rcv[3]};
                    bf16x8 p0, p1, p2;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        bf16_t h0, h1, h2;
                        x3_split1(v8[e], h0, h1, h2);
                        p0[e] = h0; p1[e] = h1; p2[e] = h2;
                    }
                    const int c8 = half * 4 + 2 * j + h;        // chunk of 8 features inside the head: stage hd*4 + (c8 >> 1), half c8 & 1
                    char* dst = ib + (c8 >> 1) * X3_TILE_B + (((c8 & 1) ^ sw) << 4);
                    *reinterpret_cast<bf16x8*>(dst) = p0;
                    *reinterpret_cast<bf16x8*>(dst + X3_PLANE_B) = p1;
                    *reinterpret_cast<bf16x8*>(dst + 2 * X3_PLANE_B) = p2;
                }
        }
    }
    // The image contract: rows past the matrix, up to the end of its last 128-row block, are ZERO (the weight gradient contracts over
    // image rows).  The wave that owns the last query block of the last sample writes them for its head's four stages -- (rows, 4
    // stages, 3 planes, 2 halves) 16-byte stores -- instead of a memset launch per layer on the host.
    if (oimg != nullptr && live && b == g.B - 1 && q0 + 32 >= g.T) {
        const int64_t rows = (int64_t)g.B * g.T, rend = (rows + 127) & ~(int64_t)127;
        const int nkb = g.D >> 4;
        const bf16x8 z = {0};
        for (int i = lane; i < (int)(rend - rows) * 24; i += 64) {
            const int64_t row = rows + i / 24;
            const int rem = i % 24, stage = rem / 6, pl = (rem % 6) >> 1, half = rem & 1;
            char* dst = oimg + ((row >> 7) * nkb + hd * 4 + stage) * (int64_t)X3_TILE_B + pl * X3_PLANE_B + (int)(row & 127) * 32 + half * 16;
            *reinterpret_cast<bf16x8*>(dst) = z;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward: EIGHT-wave workgroups (512 threads, one per CU, two waves per SIMD).  Measured on the first, four-wave version
// (scripts/lab/attn_x3_phases.py, profiles/r04_x3_phases.txt): a 6-term product needs only 768 matrix-pipe cycles per 32 x 32 x 64
// block, so what a wave pays per step is its vector-memory INSTRUCTIONS -- an LDS-DMA piece costs 150-240 cycles to issue under
// load, a dK/dV step issued 28 of them (3.3k cycles), and its 256 registers spilled (every scratch reload waits for all of them).
// With eight waves sharing each tile a wave issues 3 tile pieces per step instead of 6, and G travels through a private LDS tile
// (4 DMA pieces, no registers) instead of 2 x 16 registers filled by 16 scalar loads.
//   wave w: tile DMA of operand w >> 2 (dkdv: Q | dO, dq: K | V), rows 8 (w & 3) .. + 7 of its three planes.
// ---------------------------------------------------------------------------------------------------------------------------------
#define X3_BW 8                            // waves per backward workgroup

// dQ: workgroup = (b, h, 256 queries); the dO planes of the wave's 32 queries in registers; K / V tile planes stream through the
// ring; every wave streams ITS score blocks (its q-block x key block j, as stored: query on the lane) into a private two-slot
// LDS ring two steps ahead and ITS 32 x 32 block of G into a private single-slot tile one step ahead -- all by LDS-DMA (register
// prefetch rings become loop-carried copies that hipcc waits for right after issuing the loads).
//   dP^T = V dO^T (24 MFMAs)   dS^T = exp2(S - lse2) (dP^T + G/H - delta)   dQ += dS K (24 MFMAs)
__device__ __forceinline__ void attn_dq_x3_body(char* smem, float* ssm, float* gsm, int bid, int nblk, bool tails, const X3Geom& g,
                                                const bf16_t* __restrict__ kp, const bf16_t* __restrict__ vp,
                                                const bf16_t* __restrict__ dop, const float* __restrict__ lse2,
                                                const float* __restrict__ delta, const float* __restrict__ sres,
                                                const float* __restrict__ gm, int64_t gm_sb, int64_t gm_st, float* __restrict__ dq) {
    const int NB = (g.T + 31) >> 5, nqt = tails ? NB / X3_BW : (NB + X3_BW - 1) / X3_BW;
    int id = acr_xcd_remap(bid, nblk);
    const int qt = id % nqt; id /= nqt;
    const int hd = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int q0 = (qt * X3_BW + wave) * 32;
    const bool live = q0 < g.T;
    const int64_t pbase = (int64_t)b * g.T * g.D + (int64_t)hd * 64;
    const int op = wave >> 2, pc = wave & 3;               // this wave's share of the tile DMA: operand (K | V), 8-row piece
    const bf16_t* tb = (op ? vp : kp) + pbase;
    char* tdst = smem + op * X3_TILE_B;
    // score blocks of this wave: (qb = q0 / 32, kb = step), copied as stored (lane l of piece gq owns floats gq*256 + 4 l ..)
    const float* srow = sres + x3_block(g.H, NB, b, hd, min(q0 >> 5, NB - 1), 0) + lane * 4;
    float* sw = ssm + wave * 2 * X3_SB_FLOATS;
    auto dma_scores = [&](int kblk, int slot) {
        const float* src = srow + (int64_t)kblk * X3_SB_FLOATS;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            __builtin_amdgcn_global_load_lds((x3_glb_vp)(src + gq * 256), (x3_lds_vp)(sw + slot * X3_SB_FLOATS + gq * 256), 16, 0, 0);
    };
    // G block of the step: rows = this wave's 32 queries (fixed), 128 bytes = the step's 32 keys; 16-byte chunk c of row q is
    // stored in slot c ^ ((q >> 1) & 7) (the lane's row reads are then bank-conflict free); columns clamped into the row
    const float* gb0 = gm ? gm + (int64_t)b * gm_sb : nullptr;          // uniform
    float* gw = gsm + wave * X3_SB_FLOATS;
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
            __builtin_amdgcn_global_load_lds((x3_glb_vp)(grow[p] + min(k0 + gchunk[p], (int)gm_st - 4)), (x3_lds_vp)(gw + p * 256), 16, 0, 0);
    };
    x3_dma_tile(tdst, tb, g.plane, g.D, 0, g.T, pc, lane);
    if (live) {
        dma_g(0);
        dma_scores(0, 0);
        if (32 < g.T) dma_scores(1, 1);
    }
    bf16x8 dof[3][4];
    x3_rows_from_global(dof, dop + pbase, g.plane, g.D, q0, g.T, r, h);
    const bool qok = q0 + r < g.T;
    const float l2q = qok ? lse2[((int64_t)b * g.H + hd) * g.T + q0 + r] : INFINITY;    // queries beyond T: p = exp2(-inf) = 0
    const float dl = qok ? delta[((int64_t)b * g.H + hd) * g.T + q0 + r] : 0.f;
    const float invH = gb0 ? 1.f / (float)g.H : 0.f;
    f32x16 dq0 = {0}, dq1 = {0};
    const X3Lane lb = x3_lane(lane, smem);
    const int doff = x3_dma_off(g.D, pc, lane);
    const uint32_t saddr = x3_lds_addr(sw) + lane * 16;                                 // + slot * 4096 + gq * 1024
    uint32_t gaddr[4];                                                                  // quad gq = keys 8 gq + 4 h .. + 3 of row r
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) gaddr[gq] = x3_lds_addr(gw) + r * 128 + (((2 * gq + h) ^ ((r >> 1) & 7)) << 4);
    auto step = [&](int k0, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int KOFF = SLOT * X3_SLOT_B, VOFF = KOFF + X3_TILE_B;
        // Issue order of a live wave's step (younger to the right):  tile(t+1) [3] | G(t+1) [4]  scores(t+2) [4], all LDS-DMA.
        // At this barrier tile(t) must have landed; behind it the previous step issued G(t) and scores(t+1).
        x3_barrier((k0 > 0 && live) ? (gb0 ? 4 : 0) + (k0 + 32 < g.T ? 4 : 0) : 0);
        auto dma_next = [&] {                              // this wave's three pieces of tile(t+1)
            if (k0 + 64 <= g.T) x3_dma_tile_i(tdst + (SLOT ^ 1) * X3_SLOT_B, tb + (int64_t)(k0 + 32) * g.D, g.plane, doff, pc);
            else if (k0 + 32 < g.T) x3_dma_tile(tdst + (SLOT ^ 1) * X3_SLOT_B, tb, g.plane, g.D, k0 + 32, g.T, pc, lane);
            X3_FENCE();
        };
        if (!live) { dma_next(); return; }
        f32x16 dp = {0};
        x3_rowop<VOFF>(dp, lb, dof, dma_next);             // dP^T[key = krow][query = r]; tile(t+1) issued after the first MFMA group
        // G(t) (issued in the previous step) must have landed in this wave's tile: behind it are scores(t+1) [4] and this step's
        // tile(t+1) [3].  The score block (t) is older than it.
        if (k0 > 0) x3_wait_vm(k0 + 32 < g.T ? 7 : 0);
        f32x4 s4[4], g4[4];
        X3_RD128(s4[0], saddr, SLOT * 4096); X3_RD128(s4[1], saddr, SLOT * 4096 + 1024);
        X3_RD128(s4[2], saddr, SLOT * 4096 + 2048); X3_RD128(s4[3], saddr, SLOT * 4096 + 3072);
        if (gb0 != nullptr) {
            X3_RD128(g4[0], gaddr[0], 0); X3_RD128(g4[1], gaddr[1], 0); X3_RD128(g4[2], gaddr[2], 0); X3_RD128(g4[3], gaddr[3], 0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(g4[0]), "+v"(g4[1]), "+v"(g4[2]), "+v"(g4[3]));
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
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s4[0]), "+v"(s4[1]), "+v"(s4[2]), "+v"(s4[3]));
        f32x16 ds;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            ds[reg] = __builtin_amdgcn_exp2f(s4[reg >> 2][reg & 3] - l2q) * (dp[reg] + g4[reg >> 2][reg & 3] * invH - dl);
        // the private tiles have been read: G is refilled for the next step and the score slot for the step AFTER next -- between
        // the MFMA groups of the second product, like the second half of the split
        bf16x8 z0[3], z1[3];
        x3_split_acc<0>(ds, z0);
        x3_accop<KOFF, true>(dq0, dq1, z0, z1, lb,          // dQ[query = krow][d = 32*blk + r]
                             [&] { x3_split_acc<1>(ds, z1); },
                             [&] { if (k0 + 32 < g.T) dma_g(k0 + 32); X3_FENCE(); },
                             [&] { if (k0 + 64 < g.T) dma_scores((k0 >> 5) + 2, SLOT); X3_FENCE(); });
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

// dK, dV: workgroup = (b, h, 256 keys); the V planes of the wave's 32 keys in registers; Q / dO tile planes stream through the
// shared ring; every wave also streams ITS score blocks (q-block j x its key block) by LDS-DMA into a private two-slot ring and
// reads them transposed (key on the lane), exactly as attn_dkdv_sres_body does, and ITS 32 x 32 block of G into a private
// single-slot LDS tile (refilled right after the step's reads; natural [query][key] rows: the lane's reads are 32 consecutive
// floats); lse2 / delta of the step's 32 queries sit one per lane and reach the accumulator rows by ds_bpermute.
//   dP = dO V^T (24 MFMAs)   P = exp2(S - lse2)   dS = P (dP + G/H - delta)   dV += P^T dO (24)   dK += dS^T Q (24)
__device__ __forceinline__ void attn_dkdv_x3_body(char* smem, float* ssm, float* gsm, float* rcm, int bid, int nblk, bool tails,
                                                  const X3Geom& g,
                                                  const bf16_t* __restrict__ qp, const bf16_t* __restrict__ vp,
                                                  const bf16_t* __restrict__ dop, const float* __restrict__ lse2,
                                                  const float* __restrict__ delta, const float* __restrict__ sres,
                                                  const float* __restrict__ gm, int64_t gm_sb, int64_t gm_st, float* __restrict__ dk,
                                                  float* __restrict__ dv) {
    const int NB = (g.T + 31) >> 5, nkt = tails ? NB / X3_BW : (NB + X3_BW - 1) / X3_BW;
    int id = acr_xcd_remap(bid, nblk);
    const int ktile = id % nkt; id /= nkt;
    const int hd = id % g.H;
    const int b = id / g.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int key0 = (ktile * X3_BW + wave) * 32;
    const bool live = key0 < g.T;
    const int64_t pbase = (int64_t)b * g.T * g.D + (int64_t)hd * 64;
    const int op = wave >> 2, pc = wave & 3;               // this wave's share of the tile DMA: operand (Q | dO), 8-row piece
    const bf16_t* tb = (op ? dop : qp) + pbase;
    char* tdst = smem + op * X3_TILE_B;
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
    // G block of the step: rows = the 32 queries, 128 bytes = this wave's 32 keys.  Columns are clamped into the row (the last
    // key block reaches beyond T: those lanes' P is exactly 0 and whatever they compute never leaves their own accumulator row).
    const float* gb0 = gm ? gm + (int64_t)b * gm_sb : nullptr;          // uniform
    float* gw = gsm + wave * X3_SB_FLOATS;
    const int gcol = min(key0 + 4 * (lane & 7), (int)gm_st - 4);
    auto dma_g = [&](int q0) {
        if (gb0 == nullptr) return;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float* src = gb0 + (int64_t)min(q0 + 8 * p + (lane >> 3), g.T - 1) * gm_st + gcol;
            __builtin_amdgcn_global_load_lds((x3_glb_vp)src, (x3_lds_vp)(gw + p * 256), 16, 0, 0);
        }
    };
    // lse2 (lanes 0-31) and delta (lanes 32-63) of the step's 32 queries: one 256-byte DMA into a private LDS row
    float* rcw = rcm + wave * 64;
    auto dma_rc = [&](int q0) {
        const float* src = (lane < 32 ? lrow : drow) + min(q0 + (lane & 31), g.T - 1);
        __builtin_amdgcn_global_load_lds((x3_glb_vp)src, (x3_lds_vp)rcw, 4, 0, 0);
    };
    x3_dma_tile(tdst, tb, g.plane, g.D, 0, g.T, pc, lane);
    if (live) {
        dma_rc(0);
        dma_g(0);
        dma_scores(0, 0);
        if (32 < g.T) dma_scores(1, 1);
    }
    bf16x8 vf[3][4];
    x3_rows_from_global(vf, vp + pbase, g.plane, g.D, key0, g.T, r, h);
    const float invH = gb0 ? 1.f / (float)g.H : 0.f;
    f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
    const X3Lane lb = x3_lane(lane, smem);
    const int doff = x3_dma_off(g.D, pc, lane);
    // transposed score reads: lane (kappa = r, h): LDS byte address = tb4[reg & 3] + slot*4096 + 128*(reg >> 2)
    uint32_t tb4[4];
    {
        const int gk = r >> 3, hk = (r >> 2) & 1, ek = r & 3, mm = 2 * gk + hk;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            tb4[j] = x3_lds_addr(ssm) + ((wave * 2 * X3_SB_FLOATS) + gk * 256 + 128 * hk + ek + 4 * ((j + 4 * h) ^ mm)) * 4;
    }
    const uint32_t gaddr = x3_lds_addr(gsm) + (wave * X3_SB_FLOATS + 4 * h * 32 + r) * 4;      // + 128 * c_reg per register
    const uint32_t rcaddr = x3_lds_addr(rcw) + 16 * h;                                          // l4[gq] at + 32 gq, d4[gq] at + 128 + 32 gq
    auto step = [&](int q0, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr int QOFF = SLOT * X3_SLOT_B, DOOFF = QOFF + X3_TILE_B;
        // Issue order of a live wave's step (younger to the right):  tile(t+1) [3] | lse2-delta(t+1) [1]  G(t+1) [4]  scores(t+2) [4],
        // all of them LDS-DMA.  At this barrier tile(t) must have landed; behind it the previous step issued lse2-delta(t), G(t)
        // and scores(t+1).
        x3_barrier((q0 > 0 && live) ? 1 + (gb0 ? 4 : 0) + (q0 + 32 < g.T ? 4 : 0) : 0);
        auto dma_next = [&] {                              // this wave's three pieces of tile(t+1)
            if (q0 + 64 <= g.T) x3_dma_tile_i(tdst + (SLOT ^ 1) * X3_SLOT_B, tb + (int64_t)(q0 + 32) * g.D, g.plane, doff, pc);
            else if (q0 + 32 < g.T) x3_dma_tile(tdst + (SLOT ^ 1) * X3_SLOT_B, tb, g.plane, g.D, q0 + 32, g.T, pc, lane);
            X3_FENCE();
        };
        if (!live) { dma_next(); return; }
        f32x16 dp = {0};
        x3_rowop<DOOFF>(dp, lb, vf, dma_next);             // dP[query = krow][key = r]; tile(t+1) issued after the first MFMA group
        // lse2-delta(t) and G(t) (issued in the previous step) must have landed in this wave's private tiles: behind them are
        // scores(t+1) [4] and this step's tile(t+1) [3].  The score block (t) is older than both: it has landed with them.
        if (q0 > 0) x3_wait_vm(q0 + 32 < g.T ? 7 : 0);
        // this wave's private tiles, read by inline asm (see X3_RD128): score block transposed, G block, lse2 | delta
        float s[16], gv[16];
        f32x4 l4[4], d4[4];
#define X3_RDS(REG) X3_RD32(s[REG], tb4[(REG) & 3], SLOT * X3_SB_FLOATS * 4 + 128 * ((REG) >> 2))
#define X3_RDG(REG) X3_RD32(gv[REG], gaddr, 128 * (((REG) & 3) + 8 * ((REG) >> 2)))
        X3_RDS(0); X3_RDS(1); X3_RDS(2); X3_RDS(3); X3_RDS(4); X3_RDS(5); X3_RDS(6); X3_RDS(7);
        X3_RDS(8); X3_RDS(9); X3_RDS(10); X3_RDS(11); X3_RDS(12); X3_RDS(13); X3_RDS(14); X3_RDS(15);
        X3_RD128(l4[0], rcaddr, 0); X3_RD128(l4[1], rcaddr, 32); X3_RD128(l4[2], rcaddr, 64); X3_RD128(l4[3], rcaddr, 96);
        X3_RD128(d4[0], rcaddr, 128); X3_RD128(d4[1], rcaddr, 160); X3_RD128(d4[2], rcaddr, 192); X3_RD128(d4[3], rcaddr, 224);
        if (gb0 != nullptr) {
            X3_RDG(0); X3_RDG(1); X3_RDG(2); X3_RDG(3); X3_RDG(4); X3_RDG(5); X3_RDG(6); X3_RDG(7);
            X3_RDG(8); X3_RDG(9); X3_RDG(10); X3_RDG(11); X3_RDG(12); X3_RDG(13); X3_RDG(14); X3_RDG(15);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gv[0]), "+v"(gv[1]), "+v"(gv[2]), "+v"(gv[3]), "+v"(gv[4]), "+v"(gv[5]), "+v"(gv[6]),
                         "+v"(gv[7]), "+v"(gv[8]), "+v"(gv[9]), "+v"(gv[10]), "+v"(gv[11]), "+v"(gv[12]), "+v"(gv[13]), "+v"(gv[14]), "+v"(gv[15]));
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) gv[reg] = 0.f;
        }
#undef X3_RDS
#undef X3_RDG
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]),
                     "+v"(s[8]), "+v"(s[9]), "+v"(s[10]), "+v"(s[11]), "+v"(s[12]), "+v"(s[13]), "+v"(s[14]), "+v"(s[15]), "+v"(l4[0]),
                     "+v"(l4[1]), "+v"(l4[2]), "+v"(l4[3]), "+v"(d4[0]), "+v"(d4[1]), "+v"(d4[2]), "+v"(d4[3]));
        if (q0 + 32 > g.T) {                               // last query block: rows beyond T are junk, P = 0 there
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (q0 + acr_krow(reg, h) >= g.T) s[reg] = -INFINITY;
        }
        f32x16 p, ds;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {                   // krow(4 gq + e, h) = 8 gq + 4 h + e: four consecutive queries
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int reg = 4 * gq + e;
                const float pv = __builtin_amdgcn_exp2f(s[reg] - l4[gq][e]);
                p[reg] = pv;
                ds[reg] = pv * (dp[reg] + gv[reg] * invH - d4[gq][e]);
            }
        }
        // the private tiles have been read (their values are in p / ds): they are refilled between the MFMA groups below, like
        // the splits of P's second half and of dS; the DMAs land under the rest of the step and the next step's counted waits
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every read of the private tiles has returned
        bf16x8 z0[3], z1[3];
        x3_split_acc<0>(p, z0);
        x3_accop<DOOFF, true>(dv0, dv1, z0, z1, lb,
                              [&] { x3_split_acc<1>(p, z1); },
                              [&] { if (q0 + 32 < g.T) dma_rc(q0 + 32); X3_FENCE(); },
                              [&] { if (q0 + 32 < g.T) dma_g(q0 + 32); X3_FENCE(); });
        x3_split_acc<0>(ds, z0);
        x3_accop<QOFF, true>(dk0, dk1, z0, z1, lb,
                             [&] { x3_split_acc<1>(ds, z1); },
                             [&] { if (q0 + 64 < g.T) dma_scores((q0 >> 5) + 2, SLOT); X3_FENCE(); });
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

// dK/dV and dQ in ONE launch (one partly filled last round instead of two).  Grid: [dK/dV full | dQ full | dK/dV split tails | dQ
// split tails]: T = 785 is 25 blocks = 3 eight-wave workgroups + ONE block per (b, h) and sweep, which as a fourth workgroup with a
// single live wave took 3 of the launch's 12 rounds of the chip.  That block is a split-tail workgroup of the resident-score
// generation instead (attn_f32_sres_tails.h, eight waves dealing the other dimension; exact-fp32 products from the fp32 q, k, v,
// d_o), dispatched after every full workgroup.
__global__ __launch_bounds__(64 * X3_BW, 2) void attn_bwd_x3_kernel(X3Geom g, const bf16_t* __restrict__ qp, const bf16_t* __restrict__ kp,
                                                                   const bf16_t* __restrict__ vp, const bf16_t* __restrict__ dop,
                                                                   const float* __restrict__ lse2, const float* __restrict__ delta,
                                                                   const float* __restrict__ sres, const float* __restrict__ gm,
                                                                   int64_t gm_sb, int64_t gm_st, float* __restrict__ dq,
                                                                   float* __restrict__ dk, float* __restrict__ dv,
                                                                   const float* __restrict__ q32, const float* __restrict__ k32,
                                                                   const float* __restrict__ v32, const float* __restrict__ do32,
                                                                   AttnGeom g32, int ntail) {
    // one LDS block, carved: [slot][Q | dO planes] resp. [slot][K | V planes] (48 KB) | [wave][slot] score blocks (64 KB) | [wave] G
    // block (32 KB) | [wave][lse2 x 32 | delta x 32] (2 KB); the split tails use the first 128 KB as 8 tile images + 8 score blocks
    __shared__ __attribute__((aligned(1024))) char lds[2 * X3_SLOT_B + X3_BW * 3 * X3_SB_FLOATS * 4 + X3_BW * 256];
    char* smem = lds;
    float* ssm = reinterpret_cast<float*>(lds + 2 * X3_SLOT_B);
    float* gsm = ssm + X3_BW * 2 * X3_SB_FLOATS;
    float* rcm = gsm + X3_BW * X3_SB_FLOATS;
    const int half = ((int)gridDim.x - 2 * ntail) >> 1;
    const int bid = (int)blockIdx.x;
    if (bid < half)
        attn_dkdv_x3_body(smem, ssm, gsm, rcm, bid, half, ntail != 0, g, qp, vp, dop, lse2, delta, sres, gm, gm_sb, gm_st, dk, dv);
    else if (bid < 2 * half)
        attn_dq_x3_body(smem, ssm, gsm, bid - half, half, ntail != 0, g, kp, vp, dop, lse2, delta, sres, gm, gm_sb, gm_st, dq);
    else if (bid < 2 * half + ntail)
        attn_dkdv_tail_body<X3_BW>(reinterpret_cast<float*>(lds), reinterpret_cast<float*>(lds) + X3_BW * DT_FLOATS, g32, q32, v32, do32, lse2,
                                   delta, sres, gm, gm_sb, gm_st, dk, dv, acr_xcd_remap(bid - 2 * half, ntail));
    else
        attn_dq_tail_body<X3_BW>(reinterpret_cast<float*>(lds), g32, k32, v32, do32, lse2, delta, sres, gm, gm_sb, gm_st, dq,
                                 acr_xcd_remap(bid - 2 * half - ntail, ntail));
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
// one leftover 32-row block beyond a whole number of workgroups (nw blocks each), and at least one full workgroup
static bool x3_split_tail(int NB, int nw) { return acr_opt(ACR_OPT_ATTN_F32_NOSPLITTAIL) == 0 && (NB % nw) == 1 && NB > nw; }
static X3Geom x3_geom(const AttnGeom& g) {
    X3Geom x;
    x.B = g.B; x.H = g.H; x.T = g.T; x.D = g.H * 64; x.scale = g.scale;
    x.plane = (int64_t)g.B * g.T * x.D;
    x.osb = g.osb; x.ost = g.ost; x.osh = g.osh;
    x.sb = g.sb; x.st = g.st; x.sh = g.sh;
    return x;
}

// does the forward with an o-image epilogue keep this T's leftover block on a split-tail workgroup?  (then the image is not offered:
// the tail bodies write fp32 o only)
bool acr_attn_x3_fwd_uses_split_tail(int T) {
    const int NB = (T + 31) / 32;
    return x3_split_tail(NB, 4) && NB >= 33;
}

void acr_attn_fwd_f32_x3(const AttnGeom& g, const float* q, const float* k, const float* v, float* o, float* lse2, float* scores,
                         float* pmean, int64_t pmean_sb, int64_t pmean_st, hipStream_t st, char* oimg) {
    const X3Geom x = x3_geom(g);
    bf16_t* planes = reinterpret_cast<bf16_t*>(scores + x3_score_floats(g));
    X3SplitArgs a;
    a.src[0] = q; a.src[1] = k; a.src[2] = v;
    a.sb = g.sb; a.st = g.st; a.sh = g.sh; a.dst = planes; a.plane = x.plane; a.B = g.B; a.T = g.T; a.H = g.H;
    const int64_t n8 = (int64_t)g.B * g.T * g.H * 8;
    hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)((n8 + 255) / 256), 3), dim3(256), 0, st, a);
    const int NB = (g.T + 31) / 32;
    // measured (scripts/lab/attn_gen.py, A/B through ACR_OPT_ATTN_F32_NOSPLITTAIL): the forward's split tails pay from NB = 33 on
    // (T = 1025: 0.624 -> 0.612 ms, T = 2305 at B = 2: 0.355 -> 0.342 ms) but not at T = 785 (0.705 -> 0.724 ms: a tail workgroup's
    // 6-7 unpipelined exact-fp32 steps take about as long as the 25 pipelined split-product steps of a three-per-CU full one)
    // (the split-tail bodies write fp32 o only: a launch that also writes o's image keeps the leftover block on an ordinary workgroup)
    const int ntail = (acr_attn_x3_fwd_uses_split_tail(g.T) && oimg == nullptr) ? g.B * g.H : 0;
    hipLaunchKernelGGL(attn_fwd_x3_kernel, dim3(g.B * g.H * (ntail ? NB / 4 : (NB + 3) / 4) + ntail), dim3(256), 0, st, x, (const bf16_t*)planes,
                       (const bf16_t*)(planes + 3 * x.plane), (const bf16_t*)(planes + 6 * x.plane), o, lse2, scores, q, k, v, g, ntail, oimg);
    if (pmean) acr_attn_pmean_sres(g, scores, lse2, pmean, pmean_sb, pmean_st, st);
}

void acr_attn_bwd_f32_x3(const AttnGeom& g, const float* q, const float* k, const float* v, const float* o, const float* d_o,
                         const float* lse2, const float* scores, const float* gm, int64_t gm_sb, int64_t gm_st, float* dq, float* dk,
                         float* dv, float* delta_ws, hipStream_t st) {
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
    const int ntail = x3_split_tail(NB, X3_BW) ? g.B * g.H : 0;
    const int nmain = g.B * g.H * (ntail ? NB / X3_BW : (NB + X3_BW - 1) / X3_BW);
    hipLaunchKernelGGL(attn_bwd_x3_kernel, dim3(2 * nmain + 2 * ntail), dim3(64 * X3_BW), 0, st, x, planes, planes + 3 * x.plane,
                       planes + 6 * x.plane, (const bf16_t*)dop, lse2, (const float*)delta_ws, scores, gm, gm_sb, gm_st, dq, dk, dv, q, k, v, d_o, g,
                       ntail);
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
