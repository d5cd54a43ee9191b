// Batched input pipeline of the ACR training / CAM steps:
//   myTool.py:1158-1199 get_data_from_chunk_v2   decode -> RandomResizeLong (cv2.resize, bilinear) -> flip ->
//                                                (x/255 - mean)/std -> zero-padded RandomCrop to S x S
//   myTool.py:1364-1403 get_data_from_chunk_val  decode -> cv2.resize(S, S) -> (x/255 - mean)/std
// The host decodes (any decoder) and draws the geometry; this kernel does everything else for the WHOLE batch in one
// launch, reading the decoded uint8 HWC RGB images once and writing the (B,3,S,S) network input once: resize, flip,
// normalisation and crop are an index map plus a pointwise affine, so they fuse into one gather per output pixel.
// HBM-bound: 4 source pixels (L2-resident neighbours) + 12 or 6 bytes written per output pixel.
//
// cv2.resize(float image, INTER_LINEAR) semantics (the reference converts to float64 BEFORE resizing, so it is the
// float path, not the 11-bit fixed-point uint8 path): sample position (d + 0.5) * in/out - 0.5, clamped to the border
// -- identical to F.interpolate(bilinear, align_corners=False) without antialiasing.
#include "acr_common.h"

struct PreImg {            // mirrors acr_pre_image (include/acr_hip.h)
    int64_t offset;        // byte offset of the image in the packed uint8 buffer
    int32_t h, w;          // decoded size
    int32_t rh, rw;        // size after the resize step
    int32_t flip;          // 1: horizontal flip after the resize
    int32_t cont_top, cont_left, img_top, img_left, ch, cw;   // RandomCrop boxes (myTool.py:923-955)
};

template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ packed, const PreImg* __restrict__ tab,
                                                         T* __restrict__ out, int S, float m0, float m1, float m2, float s0,
                                                         float s1, float s2) {
    const int b = blockIdx.y;
    const PreImg im = tab[b];
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= S * S) return;
    const int y = pix / S, x = pix - y * S;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    const int cy = y - im.cont_top, cx = x - im.cont_left;
    if (cy >= 0 && cy < im.ch && cx >= 0 && cx < im.cw) {
        const int ry = im.img_top + cy;                       // pixel of the resized (and flipped) image
        int rx = im.img_left + cx;
        if (im.flip) rx = im.rw - 1 - rx;
        // sample position (d + 0.5) * in/out - 0.5 = ((2d + 1) * in - out) / (2 * out), in exact integer arithmetic: the
        // reference computes it in float64, and an fp32 product loses ~3e-5 of a pixel at x ~ 500, i.e. 2e-4 of the output
        const int ny = (2 * ry + 1) * im.h - im.rh, dy = 2 * im.rh;
        const int nx = (2 * rx + 1) * im.w - im.rw, dx = 2 * im.rw;
        const int y0 = ny < 0 ? 0 : min(ny / dy, im.h - 1), x0 = nx < 0 ? 0 : min(nx / dx, im.w - 1);
        const int y1 = min(y0 + 1, im.h - 1), x1 = min(x0 + 1, im.w - 1);
        const float ly = (ny < 0 || y0 >= im.h - 1) ? 0.f : (float)(ny - y0 * dy) / (float)dy;
        const float lx = (nx < 0 || x0 >= im.w - 1) ? 0.f : (float)(nx - x0 * dx) / (float)dx;
        const uint8_t* p = packed + im.offset;
        const uint8_t* p00 = p + ((int64_t)y0 * im.w + x0) * 3;
        const uint8_t* p01 = p + ((int64_t)y0 * im.w + x1) * 3;
        const uint8_t* p10 = p + ((int64_t)y1 * im.w + x0) * 3;
        const uint8_t* p11 = p + ((int64_t)y1 * im.w + x1) * 3;
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
        const float v0 = w00 * p00[0] + w01 * p01[0] + w10 * p10[0] + w11 * p11[0];
        const float v1 = w00 * p00[1] + w01 * p01[1] + w10 * p10[1] + w11 * p11[1];
        const float v2 = w00 * p00[2] + w01 * p01[2] + w10 * p10[2] + w11 * p11[2];
        r0 = (v0 / 255.f - m0) / s0;
        r1 = (v1 / 255.f - m1) / s1;
        r2 = (v2 / 255.f - m2) / s2;
    }
    T* o = out + (int64_t)b * 3 * S * S + pix;
    acr_store1<T>(o, r0);
    acr_store1<T>(o + (int64_t)S * S, r1);
    acr_store1<T>(o + 2 * (int64_t)S * S, r2);
}

extern "C" int acr_preprocess_batch(const void* packed_u8, const void* table, int32_t batch, int32_t S, const float* mean3,
                                    const float* std3, int32_t out_dtype, void* out, void* stream) {
    ACR_CHECK_ARG(packed_u8 && table && out && mean3 && std3, "acr_preprocess_batch: null pointer");
    ACR_CHECK_ARG(batch > 0 && S > 0, "acr_preprocess_batch: bad geometry batch=%d S=%d", batch, S);
    static_assert(sizeof(PreImg) == sizeof(acr_pre_image), "acr_pre_image layout");
    const dim3 grid((unsigned)((S * S + 255) / 256), (unsigned)batch);
    if (out_dtype == ACR_F32)
        hipLaunchKernelGGL((preprocess_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)packed_u8,
                           (const PreImg*)table, (float*)out, S, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    else if (out_dtype == ACR_BF16)
        hipLaunchKernelGGL((preprocess_kernel<__bf16>), grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)packed_u8,
                           (const PreImg*)table, (__bf16*)out, S, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    else {
        acr_set_error("acr_preprocess_batch: unknown output dtype %d", out_dtype);
        return ACR_ERR_UNSUPPORTED;
    }
    return acr_check_launch("acr_preprocess_batch");
}
