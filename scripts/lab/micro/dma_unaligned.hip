// Does global_load_lds_dwordx4 accept a global address that is only 4-byte aligned?  (shifted 3x3-convolution taps)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;
__global__ void k(const float* src, float* out, int shift) {
    __shared__ __attribute__((aligned(1024))) float s[256];
    const int lane = threadIdx.x;
    __builtin_amdgcn_global_load_lds((glb_vp)(src + shift + 4 * lane), (lds_vp)s, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[4 * lane + i] = s[4 * lane + i];
}
int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 4096); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 6; ++shift) {
        hipMemset(o, 0, 1024);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, shift);
        std::vector<float> r(256);
        hipError_t e = hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += r[i] != (float)(i + shift);
        printf("shift %d floats: %s (%d wrong, err %d) first %g %g %g %g\n", shift, bad ? "WRONG" : "ok", bad, (int)e, r[0], r[1], r[2], r[3]);
    }
    return 0;
}
