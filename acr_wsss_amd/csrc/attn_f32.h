// Shared between attn_f32.hip (first-generation fp32 / fp32-math kernels, C entry points) and attn_f32_dma.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct AttnGeom {
    int B, H, T;
    float scale;
    int64_t sb, st, sh;       // q/k/v (and dq/dk/dv)
    int64_t osb, ost, osh;    // o / do
};

void acr_attn_fwd_f32_dma(const AttnGeom& g, const float* q, const float* k, const float* v, float* o, float* lse2, float* pmean,
                          int64_t pmean_sb, int64_t pmean_st, hipStream_t st);
void acr_attn_bwd_f32_dma(const AttnGeom& g, const float* q, const float* k, const float* v, const float* o, const float* d_o,
                          const float* lse2, const float* gm, int64_t gm_sb, int64_t gm_st, float* dq, float* dk, float* dv,
                          float* delta, hipStream_t st);

// resident-score generation (attn_f32_sres.hip)
void acr_attn_fwd_f32_sres(const AttnGeom& g, const float* q, const float* k, const float* v, float* o, float* lse2, float* scores,
                           float* pmean, int64_t pmean_sb, int64_t pmean_st, hipStream_t st);
void acr_attn_bwd_f32_sres(const AttnGeom& g, const float* q, const float* k, const float* v, const float* o, const float* d_o,
                           const float* lse2, const float* scores, const float* gm, int64_t gm_sb, int64_t gm_st, float* dq,
                           float* dk, float* dv, float* delta, hipStream_t st);
