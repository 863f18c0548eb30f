// Input packing in front of the encoder.
//   slices2rgb (reference mst/models/dino.py:10-27; its call at dino.py:129 is commented out): three consecutive gray slices
//   become the three channels of one image, [B,1,D,H,W] -> [B*ceil(D/3), 3, H, W]; when D is not a multiple of 3 the volume is
//   padded along D with its own first 3 - D%3 slices.  Pure index arithmetic, HBM-bound: 16-byte loads and stores.
#include "mst_common.h"

namespace {

// one thread per 16 bytes of output; `hw16` = 16-byte pieces per slice
__global__ void slices2rgb_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int D, int Dp, int64_t hw16, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / hw16, p = i - s * hw16;      // s = output slice index b*Dp + d
        const int64_t b = s / Dp;
        int d = (int)(s - b * Dp);
        if (d >= D) d -= D;                                // padding = the first slices again (dino.py:19-20)
        out[i] = in[(b * D + d) * hw16 + p];
    }
}

__global__ void slices2rgb_scalar_kernel(const char* __restrict__ in, char* __restrict__ out, int D, int Dp, int64_t hw,
                                         int esz, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / hw, p = i - s * hw;
        const int64_t b = s / Dp;
        int d = (int)(s - b * Dp);
        if (d >= D) d -= D;
        const char* src = in + ((b * D + d) * hw + p) * esz;
        char* dst = out + i * esz;
        for (int e = 0; e < esz; ++e) dst[e] = src[e];
    }
}

}  // namespace

int launch_slices2rgb(const void* vol, int dt, int B, int D, int H, int W, void* out, hipStream_t s) {
    const int esz = dt == MST_F32 ? 4 : 2;
    const int pad = (D % 3) ? 3 - D % 3 : 0;
    MST_CHECK_ARG(pad <= D, "slices2rgb: D=%d cannot pad itself to a multiple of 3", D);
    const int Dp = D + pad;
    const int64_t hw = (int64_t)H * W;
    if ((hw * esz) % 16 == 0 && ((uintptr_t)vol % 16 == 0) && ((uintptr_t)out % 16 == 0)) {
        const int64_t hw16 = hw * esz / 16, total = (int64_t)B * Dp * hw16;
        const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        slices2rgb_kernel<<<dim3(grid), dim3(256), 0, s>>>((const uint4*)vol, (uint4*)out, D, Dp, hw16, total);
    } else {
        const int64_t total = (int64_t)B * Dp * hw;
        const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        slices2rgb_scalar_kernel<<<dim3(grid), dim3(256), 0, s>>>((const char*)vol, (char*)out, D, Dp, hw, esz, total);
    }
    return mst_check_launch("slices2rgb");
}
