// Tile helpers shared by the fp32 attention kernels (attn_f32_dma.hip: recompute generation; attn_f32_sres.hip: resident-score
// generation): LDS-DMA of 32-row x 64-float tiles into chunk-XOR-swizzled LDS images, the MFMA operand walks over them
// (rowop / accop) and their lane-base + immediate addressed forms.  Device code only; include after acr_common.h.
#pragma once
#include "acr_common.h"
#include "attn_f32.h"

#define DT_FLOATS 2048             // one 32-row x 64-float tile

typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

// DMA rows row0 .. row0+31 (clamped to Tn-1) of a (T, 64) matrix with row stride st into a swizzled LDS tile; the 8 pieces
// (4 rows each) are dealt over the workgroup's 4 waves.
__device__ __forceinline__ void dma_tile32(float* lds, const float* __restrict__ g, int64_t st, int row0, int Tn, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i;
        const int row = piece * 4 + (lane >> 4);
        const int c = (lane & 15) ^ (row & 15);              // logical 16-byte chunk that lands in physical slot lane & 15
        const float* src = g + (int64_t)min(row0 + row, Tn - 1) * st + c * 4;
        __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(lds + piece * 256), 16, 0, 0);
    }
}
// one 256-byte DMA by lanes 0..63 of ONE wave: dst[lane] = lane < 32 ? a[i0 + lane] : b[i0 + lane - 32] (clamped to n-1)
__device__ __forceinline__ void dma_rowconst(float* lds, const float* __restrict__ a, const float* __restrict__ b, int i0, int n, int lane) {
    const float* src = (lane < 32 ? a : b) + min(i0 + (lane & 31), n - 1);
    __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)lds, 4, 0, 0);
}

__device__ __forceinline__ f32x4 sw_row4(const float* tile, int row, int chunk) {
    return *reinterpret_cast<const f32x4*>(tile + row * 64 + ((chunk ^ (row & 15)) << 2));
}
__device__ __forceinline__ float sw_elem(const float* tile, int row, int col) {
    return tile[row * 64 + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3))];
}

// acc[reg] += sum_d tile[krow(reg,h)][d] * Y[l&31][d]   (tile rows are the A operand, y = 32 registers of the lane's row)
__device__ __forceinline__ void rowop(f32x16& acc, const float* tile, const float (&y)[32], int r, int h) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4 a = sw_row4(tile, r, 8 * h + i);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], y[4 * i + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], y[4 * i + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], y[4 * i + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], y[4 * i + 3], acc, 0, 0, 0);
    }
}
// z as A operand: acc[i = z-lane][j = tile column 32*blk + r] += sum_reg z[reg] * tile[krow(reg,h)][32*blk + r]
__device__ __forceinline__ void accop_a(f32x16& acc, const f32x16& z, const float* tile, int blk, int r, int h) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z[reg], sw_elem(tile, acr_krow(reg, h), 32 * blk + r), acc, 0, 0, 0);
}
// z as B operand: acc[i = tile column][j = z-lane]
__device__ __forceinline__ void accop_b(f32x16& acc, const f32x16& z, const float* tile, int blk, int r, int h) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sw_elem(tile, acr_krow(reg, h), 32 * blk + r), z[reg], acc, 0, 0, 0);
}
// lane (r, h) takes row r, columns [32h, 32h+32) of a swizzled tile into 32 registers
__device__ __forceinline__ void rows_from_lds(float (&reg)[32], const float* tile, int r, int h) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4 v = sw_row4(tile, r, 8 * h + i);
        reg[4 * i + 0] = v[0]; reg[4 * i + 1] = v[1]; reg[4 * i + 2] = v[2]; reg[4 * i + 3] = v[3];
    }
}
// the same from global memory (row clamped, scaled; rows beyond Tn are zero)
__device__ __forceinline__ void rows_from_global(float (&reg)[32], const float* __restrict__ g, int64_t st, int row0, int Tn, int r, int h,
                                                 float mul) {
    const float okm = (row0 + r < Tn) ? mul : 0.f;
    const float* p = g + (int64_t)min(row0 + r, Tn - 1) * st + 32 * h;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * i) * okm;
        reg[4 * i + 0] = v[0]; reg[4 * i + 1] = v[1]; reg[4 * i + 2] = v[2]; reg[4 * i + 3] = v[3];
    }
}

// ---- immediate-offset addressing --------------------------------------------------------------------------------
// VALU instructions do not overlap the fp32 MFMA, so the loops keep NO vector address arithmetic: every LDS read is
// (lane base register) + (compile-time immediate).  With the chunk-XOR swizzle the lane-dependent part of a read address
// takes only a few values, precomputed once per wave (byte offsets inside a tile):
//   row reads  (rowop, rows_from_lds): row = r, chunk 8h + i        ->  rowb[i] = r*256 + (((8h) ^ (r & 15) ^ i) << 4)
//   col reads  (accop): row = krow(reg, h) = c_reg + 4h, column 32*blk + r; with L = (r >> 2) ^ 4h and
//              C = 8*blk ^ (c_reg & 15) (bit 2 of C is always 0):  address = colb[C & 3] + c_reg*256 + (C & 8)*16,
//              colb[c] = 4h*256 + ((L ^ c) << 4) + (r & 3)*4
// and the ring slots are unrolled (SLOT is a template parameter), so tile base offsets are immediates too.
struct LaneBases { int rowb[8]; int colb[4]; };
__device__ __forceinline__ LaneBases lane_bases(int r, int h) {
    LaneBases lb;
#pragma unroll
    for (int i = 0; i < 8; ++i) lb.rowb[i] = r * 256 + ((((8 * h) ^ (r & 15)) ^ i) << 4);
    const int L = (r >> 2) ^ (4 * h);
#pragma unroll
    for (int c = 0; c < 4; ++c) lb.colb[c] = h * 1024 + ((L ^ c) << 4) + (r & 3) * 4;
    return lb;
}
template <int TILE_OFF>            // TILE_OFF: byte offset of the tile inside the workgroup's LDS block `sm`
__device__ __forceinline__ void rowop_i(f32x16& acc, const char* sm, const LaneBases& lb, const float (&y)[32]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(sm + lb.rowb[i] + TILE_OFF);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], y[4 * i + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], y[4 * i + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], y[4 * i + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], y[4 * i + 3], acc, 0, 0, 0);
    }
}
template <int TILE_OFF, int BLK, int REG>
__device__ __forceinline__ float col_elem(const char* sm, const LaneBases& lb) {
    constexpr int c_reg = (REG & 3) + 8 * (REG >> 2);
    constexpr int C = (8 * BLK) ^ (c_reg & 15);
    return *reinterpret_cast<const float*>(sm + lb.colb[C & 3] + (TILE_OFF + c_reg * 256 + (C & 8) * 16));
}
template <int TILE_OFF, int BLK, int REG = 0>
__device__ __forceinline__ void accop_b_i(f32x16& acc, const f32x16& z, const char* sm, const LaneBases& lb) {
    if constexpr (REG < 16) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(col_elem<TILE_OFF, BLK, REG>(sm, lb), z[REG], acc, 0, 0, 0);
        accop_b_i<TILE_OFF, BLK, REG + 1>(acc, z, sm, lb);
    }
}
template <int TILE_OFF, int BLK, int REG = 0>
__device__ __forceinline__ void accop_a_i(f32x16& acc, const f32x16& z, const char* sm, const LaneBases& lb) {
    if constexpr (REG < 16) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z[REG], col_elem<TILE_OFF, BLK, REG>(sm, lb), acc, 0, 0, 0);
        accop_a_i<TILE_OFF, BLK, REG + 1>(acc, z, sm, lb);
    }
}
template <int TILE_OFF>
__device__ __forceinline__ void rows_from_lds_i(float (&reg)[32], const char* sm, const LaneBases& lb) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sm + lb.rowb[i] + TILE_OFF);
        reg[4 * i + 0] = v[0]; reg[4 * i + 1] = v[1]; reg[4 * i + 2] = v[2]; reg[4 * i + 3] = v[3];
    }
}
// DMA of a 32-row tile with the lane part of the source address precomputed (element offsets of the wave's two pieces
// relative to row 0 of the tile): src = uniform row-0 pointer + off[i].  Only valid for tiles fully inside [0, T).
__device__ __forceinline__ void dma_offsets32(int (&off)[2], int64_t st, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 4 + (lane >> 4);
        off[i] = row * (int)st + (((lane & 15) ^ (row & 15)) << 2);
    }
}
__device__ __forceinline__ void dma_tile32_i(float* lds, const float* __restrict__ row0ptr, const int (&off)[2], int wave) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((glb_vp)(row0ptr + off[i]), (lds_vp)(lds + (wave * 2 + i) * 256), 16, 0, 0);
}


// ---- the same tile DMA dealt over NW waves (workgroups of NW x 32 rows): piece p of the 8 goes to wave p % NW ----
template <int NW>
__device__ __forceinline__ void dma_tile32_nw(float* lds, const float* __restrict__ g, int64_t st, int row0, int Tn, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < (8 + NW - 1) / NW; ++i) {
        const int piece = wave + i * NW;
        if (piece < 8) {                                     // wave-uniform
            const int row = piece * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (row & 15);
            const float* src = g + (int64_t)min(row0 + row, Tn - 1) * st + c * 4;
            __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(lds + piece * 256), 16, 0, 0);
        }
    }
}
template <int NW>
__device__ __forceinline__ void dma_offsets32_nw(int (&off)[(8 + NW - 1) / NW], int64_t st, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < (8 + NW - 1) / NW; ++i) {
        const int row = min(wave + i * NW, 7) * 4 + (lane >> 4);
        off[i] = row * (int)st + (((lane & 15) ^ (row & 15)) << 2);
    }
}
template <int NW>
__device__ __forceinline__ void dma_tile32_nw_i(float* lds, const float* __restrict__ row0ptr, const int (&off)[(8 + NW - 1) / NW], int wave) {
#pragma unroll
    for (int i = 0; i < (8 + NW - 1) / NW; ++i) {
        const int piece = wave + i * NW;
        if (piece < 8)
            __builtin_amdgcn_global_load_lds((glb_vp)(row0ptr + off[i]), (lds_vp)(lds + piece * 256), 16, 0, 0);
    }
}

// ONE wave DMAs a whole 32-row tile (8 pieces) into its private LDS image (split-tail workgroups: every wave has its own stream)
__device__ __forceinline__ void dma_tile32_one(float* lds, const float* __restrict__ g, int64_t st, int row0, int Tn, int lane) {
#pragma unroll
    for (int piece = 0; piece < 8; ++piece) {
        const int row = piece * 4 + (lane >> 4);
        const int c = (lane & 15) ^ (row & 15);
        const float* src = g + (int64_t)min(row0 + row, Tn - 1) * st + c * 4;
        __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(lds + piece * 256), 16, 0, 0);
    }
}
// lane bases shifted by a (wave-dependent) byte offset: the *_i walks can then take TILE_OFF = 0 on a private tile
__device__ __forceinline__ LaneBases lane_bases_at(int r, int h, int byte_off) {
    LaneBases lb = lane_bases(r, h);
#pragma unroll
    for (int i = 0; i < 8; ++i) lb.rowb[i] += byte_off;
#pragma unroll
    for (int c = 0; c < 4; ++c) lb.colb[c] += byte_off;
    return lb;
}

// ---- the same walks with INLINE-ASM LDS reads (round 4) ---------------------------------------------------------------------------
// hipcc cannot tell an LDS-DMA's LDS write from an LDS load of another slot: behind a global_load_lds it puts s_waitcnt vmcnt(0)
// in front of the next LDS load builtin -- in every step of the sweeps the wave then waited for the tile it had just sent for
// (found in the ISA while building attn_f32_x3.hip; the resident-score sweeps had carried it since round 3: their "DMA issue +
// loads" phase, profiles/r03_attn_bwd_phases.txt).  Reads written as inline asm are invisible to that pass; each carries its own
// counted lgkmcnt wait (LDS operations of a wave return in order: "at most n outstanding" = "all but my n newest have landed";
// operations the compiler adds only make a wait stricter).  Addresses are LDS byte addresses (tile base included).
struct LaneBasesA { uint32_t rowb[8]; uint32_t colb[4]; };
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ LaneBasesA lane_bases_a(int r, int h, const void* sm) {
    const LaneBases lb = lane_bases(r, h);
    LaneBasesA la;
    const uint32_t b = lds_addr_of(sm);
#pragma unroll
    for (int i = 0; i < 8; ++i) la.rowb[i] = b + lb.rowb[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) la.colb[c] = b + lb.colb[c];
    return la;
}
#define ACR_LDS_RD128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "i"(OFF))
#define ACR_LDS_RD32(dst, addr, OFF) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "i"(OFF))
#define ACR_LDS_WAIT4(cnt, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define ACR_LDS_WAIT1(cnt, a) asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(a))

// acc[reg] += sum_d tile[krow(reg,h)][d] * Y[l&31][d]: four 16-byte reads in flight ahead of the MFMAs that consume them
template <int TILE_OFF>
__device__ __forceinline__ void rowop_x(f32x16& acc, const LaneBasesA& lb, const float (&y)[32]) {
    f32x4 a[8];
    ACR_LDS_RD128(a[0], lb.rowb[0], TILE_OFF); ACR_LDS_RD128(a[1], lb.rowb[1], TILE_OFF);
    ACR_LDS_RD128(a[2], lb.rowb[2], TILE_OFF); ACR_LDS_RD128(a[3], lb.rowb[3], TILE_OFF);
#define ACR_ROWSTEP(I, CNT, NEXT)                                                                   \
    ACR_LDS_WAIT1(CNT, a[I]);                                                                       \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[I][0], y[4 * (I) + 0], acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[I][1], y[4 * (I) + 1], acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[I][2], y[4 * (I) + 2], acc, 0, 0, 0);              \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[I][3], y[4 * (I) + 3], acc, 0, 0, 0);              \
    NEXT
    ACR_ROWSTEP(0, 3, ACR_LDS_RD128(a[4], lb.rowb[4], TILE_OFF);)
    ACR_ROWSTEP(1, 3, ACR_LDS_RD128(a[5], lb.rowb[5], TILE_OFF);)
    ACR_ROWSTEP(2, 3, ACR_LDS_RD128(a[6], lb.rowb[6], TILE_OFF);)
    ACR_ROWSTEP(3, 3, ACR_LDS_RD128(a[7], lb.rowb[7], TILE_OFF);)
    ACR_ROWSTEP(4, 3, )
    ACR_ROWSTEP(5, 2, )
    ACR_ROWSTEP(6, 1, )
    ACR_ROWSTEP(7, 0, )
#undef ACR_ROWSTEP
}
// the 16 column elements of accop (col_elem) of one 32-column block, read 8 ahead
template <int TILE_OFF, int BLK, int REG>
__device__ __forceinline__ void col_read_x(float& dst, const LaneBasesA& lb) {
    constexpr int c_reg = (REG & 3) + 8 * (REG >> 2);
    constexpr int C = (8 * BLK) ^ (c_reg & 15);
    ACR_LDS_RD32(dst, lb.colb[C & 3], TILE_OFF + c_reg * 256 + (C & 8) * 16);
}
template <int TILE_OFF, int BLK, bool ZA>
__device__ __forceinline__ void accop_x(f32x16& acc, const f32x16& z, const LaneBasesA& lb) {
    float t[16];
#define ACR_CR(R) col_read_x<TILE_OFF, BLK, R>(t[R], lb)
#define ACR_MM(R)                                                                                   \
    acc = ZA ? __builtin_amdgcn_mfma_f32_32x32x2f32(z[R], t[R], acc, 0, 0, 0)                       \
             : __builtin_amdgcn_mfma_f32_32x32x2f32(t[R], z[R], acc, 0, 0, 0)
    ACR_CR(0); ACR_CR(1); ACR_CR(2); ACR_CR(3); ACR_CR(4); ACR_CR(5); ACR_CR(6); ACR_CR(7);
    ACR_LDS_WAIT4(4, t[0], t[1], t[2], t[3]);
    ACR_MM(0); ACR_MM(1); ACR_MM(2); ACR_MM(3);
    ACR_CR(8); ACR_CR(9); ACR_CR(10); ACR_CR(11);
    ACR_LDS_WAIT4(4, t[4], t[5], t[6], t[7]);
    ACR_MM(4); ACR_MM(5); ACR_MM(6); ACR_MM(7);
    ACR_CR(12); ACR_CR(13); ACR_CR(14); ACR_CR(15);
    ACR_LDS_WAIT4(4, t[8], t[9], t[10], t[11]);
    ACR_MM(8); ACR_MM(9); ACR_MM(10); ACR_MM(11);
    ACR_LDS_WAIT4(0, t[12], t[13], t[14], t[15]);
    ACR_MM(12); ACR_MM(13); ACR_MM(14); ACR_MM(15);
#undef ACR_CR
#undef ACR_MM
}
