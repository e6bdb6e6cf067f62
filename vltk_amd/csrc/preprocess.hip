// Image pre-processing on the GPU: shortest-edge bilinear resize, (x - mean) / std,
// pad to the batch size -- replaces ResizeShortestEdge + Preprocess of the reference's
// vltk/legacy/processing.py:29-150 (torch F.interpolate(mode="bilinear", align_corners=False) + pad).
//
// HBM-bound: one lane per output pixel computes the 3 channels (12 input reads, 3 coalesced NCHW stores).
// Arithmetic contract: fp32, ATen's operation order for upsample_bilinear2d with align_corners=False --
// scale = in/out (float), src = fma(scale, dst+0.5, -0.5) clamped at 0, lambda1 = src - floor, lambda0 = 1 - lambda1,
// value = fma(bot, h1, top*h0) with top/bot = fma(p_1, w1, p_0*w0) -- explicit fmaf where ATen's FMA-enabled CPU
// build contracts, nothing else contracted (-ffp-contract=off).
#include "vk_common.h"

namespace vk {

constexpr int PRE_MAX_IMAGES = 32;

struct PreArgs {
    const float *raw[PRE_MAX_IMAGES];   // HWC f32 (BGR, 0-255), device pointers
    int raw_h[PRE_MAX_IMAGES], raw_w[PRE_MAX_IMAGES];
    int new_h[PRE_MAX_IMAGES], new_w[PRE_MAX_IMAGES];
    int n, Hmax, Wmax;
    float mean[3], stdv[3], pad_value;
    float *out;                          // [n, 3, Hmax, Wmax]
};

__global__ __launch_bounds__(256) void preprocess_kernel(PreArgs a) {
    const int n = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.Wmax || y >= a.Hmax) return;
    const long plane = (long)a.Hmax * a.Wmax;
    float *o = a.out + (long)n * 3 * plane + (long)y * a.Wmax + x;
    const int nh = a.new_h[n], nw = a.new_w[n];
    if (x >= nw || y >= nh) {
        o[0] = a.pad_value;
        o[plane] = a.pad_value;
        o[2 * plane] = a.pad_value;
        return;
    }
    const int H = a.raw_h[n], W = a.raw_w[n];
    const float rh = (float)H / (float)nh, rw = (float)W / (float)nw;
    // ATen's CPU kernels are built with FMA contraction: the source index is fma(scale, dst + 0.5, -0.5) and the
    // weighted sums are t0*w0 then fma(t1, w1, .) -- restated explicitly (this file is otherwise contraction-free)
    float sy = fmaf(rh, (float)y + 0.5f, -0.5f);
    float sx = fmaf(rw, (float)x + 0.5f, -0.5f);
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int yp = (y0 < H - 1) ? 1 : 0, xp = (x0 < W - 1) ? 1 : 0;
    const float h1 = sy - (float)y0, h0 = 1.f - h1;
    const float w1 = sx - (float)x0, w0 = 1.f - w1;
    const float *p00 = a.raw[n] + ((long)y0 * W + x0) * 3;
    const float *p01 = p00 + xp * 3;
    const float *p10 = p00 + (long)yp * W * 3;
    const float *p11 = p10 + xp * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = fmaf(p01[c], w1, p00[c] * w0), bot = fmaf(p11[c], w1, p10[c] * w0);
        const float v = fmaf(bot, h1, top * h0);
        o[c * plane] = (v - a.mean[c]) / a.stdv[c];
    }
}

}  // namespace vk

using namespace vk;

extern "C" int vk_preprocess(const float *const *raw_dev_ptrs_host, const int32_t *raw_hw_host, const int32_t *new_hw_host,
                             int N, int Hmax, int Wmax, const float *mean3_host, const float *std3_host, float pad_value,
                             float *out_nchw_dev, void *stream) {
    VK_REQUIRE(raw_dev_ptrs_host && raw_hw_host && new_hw_host && mean3_host && std3_host && out_nchw_dev, VK_EINVAL,
               "preprocess: null argument");
    VK_REQUIRE(N >= 1 && Hmax >= 1 && Wmax >= 1, VK_EINVAL, "preprocess: empty batch");
    for (int base = 0; base < N; base += PRE_MAX_IMAGES) {
        PreArgs a;
        memset(&a, 0, sizeof(a));
        a.n = N - base < PRE_MAX_IMAGES ? N - base : PRE_MAX_IMAGES;
        for (int i = 0; i < a.n; ++i) {
            a.raw[i] = raw_dev_ptrs_host[base + i];
            a.raw_h[i] = raw_hw_host[2 * (base + i)];
            a.raw_w[i] = raw_hw_host[2 * (base + i) + 1];
            a.new_h[i] = new_hw_host[2 * (base + i)];
            a.new_w[i] = new_hw_host[2 * (base + i) + 1];
            VK_REQUIRE(a.raw[i] && a.raw_h[i] >= 1 && a.raw_w[i] >= 1 && a.new_h[i] >= 1 && a.new_w[i] >= 1 &&
                           a.new_h[i] <= Hmax && a.new_w[i] <= Wmax, VK_EINVAL, "preprocess: bad geometry for image %d", base + i);
        }
        a.Hmax = Hmax;
        a.Wmax = Wmax;
        for (int c = 0; c < 3; ++c) {
            a.mean[c] = mean3_host[c];
            a.stdv[c] = std3_host[c];
        }
        a.pad_value = pad_value;
        a.out = out_nchw_dev + (long)base * 3 * Hmax * Wmax;
        hipLaunchKernelGGL(preprocess_kernel, dim3(ceil_div(Wmax, 64), ceil_div(Hmax, 4), a.n), dim3(256), 0,
                           (hipStream_t)stream, a);
        VK_CHECK_HIP(hipGetLastError());
    }
    return VK_OK;
}
