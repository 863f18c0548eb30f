// Saliency volume of scripts/main_predict.py (`--get_attention`): what run_pred / _pred_trans do AFTER the forward
// (l.72-105, 147-165), on the device: head mean of get_attention_maps(), test-time-augmentation flips undone while
// accumulating at the patch-grid resolution, then ONE trilinear up-sampling (F.interpolate(..., mode='trilinear'),
// align_corners=False) of [1,1,D,g,g] to the volume's [D,H,W].  HBM-bound: the output (68.7 MB at 64 x 518^2) is
// written once with 16-byte stores, the 0.35 MB low-resolution grid stays in L2.
#include "mst_common.h"

namespace {

// low[d'][y'][x'] (+)= mean_h maps[d][h][y*gw + x]; (d',y',x') = (d,y,x) with the flipped axes mirrored back.
// slice_acc[d'] (+)= slice_attn[d] (get_slice_attention() is one value per slice: its volume is a broadcast).
__global__ void saliency_accumulate_kernel(const float* __restrict__ maps, const float* __restrict__ slice_attn, int D,
                                           int heads, int gh, int gw, int Np, int flip_mask, int accumulate,
                                           float* __restrict__ low, float* __restrict__ slice_acc) {
    const int d = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int dd = (flip_mask & 1) ? D - 1 - d : d;
    if (i == 0 && slice_acc) slice_acc[dd] = (accumulate ? slice_acc[dd] : 0.f) + slice_attn[d];
    if (i >= gh * gw) return;
    const int y = i / gw, x = i - y * gw;
    float s = 0.f;
    for (int h = 0; h < heads; ++h) s += maps[((int64_t)d * heads + h) * Np + i];
    s /= (float)heads;                                   // weight.mean(dim=1)   main_predict.py:76
    const int yy = (flip_mask & 2) ? gh - 1 - y : y, xx = (flip_mask & 4) ? gw - 1 - x : x;
    float* o = low + ((int64_t)dd * gh + yy) * gw + xx;
    *o = (accumulate ? *o : 0.f) + s;
}

struct Lin { int i0, i1; float l; };
__device__ __forceinline__ Lin lin_index(int dst, int n_in, int n_out) {   // area_pixel_compute_source_index, align_corners=False
    const float src = fmaxf(((float)dst + 0.5f) * ((float)n_in / (float)n_out) - 0.5f, 0.f);
    Lin r;
    r.i0 = min((int)src, n_in - 1);
    r.i1 = min(r.i0 + 1, n_in - 1);
    r.l = src - (float)r.i0;
    return r;
}

// out[z][y][x0..x0+3] = scale * trilinear(low); one thread per 4 output columns (depth, then height, then width)
__global__ void saliency_upsample_kernel(const float* __restrict__ low, int D, int gh, int gw, float scale, int Dout, int H,
                                         int W, float* __restrict__ out) {
    const int z = blockIdx.z, y = blockIdx.y * blockDim.y + threadIdx.y;
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (x0 >= W || y >= H) return;
    const Lin lz = lin_index(z, D, Dout), ly = lin_index(y, gh, H);
    const float* p00 = low + ((int64_t)lz.i0 * gh + ly.i0) * gw;
    const float* p01 = low + ((int64_t)lz.i0 * gh + ly.i1) * gw;
    const float* p10 = low + ((int64_t)lz.i1 * gh + ly.i0) * gw;
    const float* p11 = low + ((int64_t)lz.i1 * gh + ly.i1) * gw;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const Lin lx = lin_index(min(x0 + e, W - 1), gw, W);
        const float a0 = (1.f - lz.l) * p00[lx.i0] + lz.l * p10[lx.i0];   // depth
        const float a1 = (1.f - lz.l) * p01[lx.i0] + lz.l * p11[lx.i0];
        const float b0 = (1.f - lz.l) * p00[lx.i1] + lz.l * p10[lx.i1];
        const float b1 = (1.f - lz.l) * p01[lx.i1] + lz.l * p11[lx.i1];
        const float c0 = (1.f - ly.l) * a0 + ly.l * a1;                     // height
        const float c1 = (1.f - ly.l) * b0 + ly.l * b1;
        v[e] = scale * ((1.f - lx.l) * c0 + lx.l * c1);                      // width
    }
    float* o = out + ((int64_t)z * H + y) * W + x0;
    if (x0 + 3 < W && (W & 3) == 0) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    else
        for (int e = 0; e < 4 && x0 + e < W; ++e) o[e] = v[e];
}

}  // namespace

int launch_saliency_accumulate(const float* maps, const float* slice_attn, int D, int heads, int gh, int gw, int Np,
                               int flip_mask, int accumulate, float* low, float* slice_acc, hipStream_t s) {
    saliency_accumulate_kernel<<<dim3((gh * gw + 255) / 256, D), dim3(256), 0, s>>>(maps, slice_attn, D, heads, gh, gw, Np,
                                                                                    flip_mask, accumulate, low, slice_acc);
    return mst_check_launch("saliency_accumulate");
}

int launch_saliency_upsample(const float* low, int D, int gh, int gw, float scale, int Dout, int H, int W, float* out,
                             hipStream_t s) {
    MST_CHECK_ARG(Dout <= 65535 && H <= 8 * 65535, "saliency_upsample: volume %d x %d too large for the launch grid", Dout, H);
    // 64 x 8 threads = 8 rows of 256 columns per workgroup (whole 1 KiB row pieces per wave)
    saliency_upsample_kernel<<<dim3(((W + 3) / 4 + 63) / 64, (H + 7) / 8, Dout), dim3(64, 8), 0, s>>>(low, D, gh, gw, scale, Dout, H, W, out);
    return mst_check_launch("saliency_upsample");
}
