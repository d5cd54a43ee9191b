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
// the two HBM-stream kernels of the resident-score generation on their own (shared with the split-product generation)
void acr_attn_pmean_sres(const AttnGeom& g, const float* scores, const float* lse2, float* pmean, int64_t pmean_sb, int64_t pmean_st,
                         hipStream_t st);
void acr_attn_delta_sres(const AttnGeom& g, const float* scores, const float* o, const float* d_o, const float* lse2, const float* gm,
                         int64_t gm_sb, int64_t gm_st, float* delta, hipStream_t st);

// split-product generation (attn_f32_x3.hip; acr_dtype ACR_F32_BF16X3): fp32 tensors, products as six bf16-MFMA terms.
// `scores` = acr_attn_x3_scores_floats floats (score blocks + the bf16 planes of q, k, v, kept for the backward), `delta_ws` =
// acr_attn_x3_bwd_ws_floats floats (delta + the bf16 planes of dO).
int64_t acr_attn_x3_scores_floats(const AttnGeom& g);
int64_t acr_attn_x3_bwd_ws_floats(const AttnGeom& g);
bool acr_attn_x3_fwd_uses_split_tail(int T);     // forward of this T runs split-tail workgroups (which write fp32 o only)
// oimg != nullptr: o additionally leaves as the split-product image of the (B T) x (H 64) matrix (acr_attn_fwd_scores_oimg)
void acr_attn_fwd_f32_x3(const AttnGeom& g, const float* q, const float* k, const float* v, float* o, float* lse2, float* scores,
                         float* pmean, int64_t pmean_sb, int64_t pmean_st, hipStream_t st, char* oimg = nullptr);
void acr_attn_bwd_f32_x3(const AttnGeom& g, const float* q, const float* k, const float* v, const float* o, const float* d_o,
                         const float* lse2, const float* scores, const float* gm, int64_t gm_sb, int64_t gm_st, float* dq, float* dk,
                         float* dv, float* delta_ws, hipStream_t st);
