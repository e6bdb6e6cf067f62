// HBM-bound layout / pooling kernels of the FRCNN forward (gfx950).  Every kernel moves
// 16-byte vectors per lane along the channel axis of NHWC tensors (8 f16 / 4 f32), so a
// wavefront touches 1 KiB of contiguous channels per instruction.
//
//   stem_pack      NCHW f32 image -> zero-bordered NHWC4 (the stem conv's input image)
//   maxpool3x3s2   BasicStem's max-pool                      reference frcnn.py:875-878
//   roi_pool       torchvision.ops.RoIPool forward            reference frcnn.py:1179,1198
//   mean_pool      box_features.mean(dim=[2,3])               reference frcnn.py:1401
//   nchw<->nhwc    layout plumbing for the stage-level tests
#include <cfloat>
#include <cstdlib>

#include "vk_common.h"

namespace vk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <typename T>
struct V16;
template <>
struct V16<_Float16> {
    typedef half8 type;
    static constexpr int N = 8;
};
template <>
struct V16<float> {
    typedef floatx4 type;
    static constexpr int N = 4;
};

// ---------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float *__restrict__ x, T *__restrict__ y, int C, int HW, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long step = (long)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        int c = (int)(i % C);
        long p = i / C;            // n*HW + hw
        long n = p / HW;
        long hw = p - n * HW;
        y[i] = (T)x[(n * C + c) * HW + hw];
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T *__restrict__ x, float *__restrict__ y, int C, int HW, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long step = (long)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        long hw = i % HW;
        long p = i / HW;           // n*C + c
        long n = p / C;
        int c = (int)(p - n * C);
        y[i] = (float)x[(n * HW + hw) * C + c];
    }
}

static inline int grid_for(long total, int block) {
    long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

// ---------------------------------------------------------------------------
// x [N,3,H,W] f32 -> y [N,Hp,Wp,4] T; y[n, h+3, w+3, c] = x[n,c,h,w], everything else 0.
template <typename T>
__global__ void stem_pack_kernel(const float *__restrict__ x, T *__restrict__ y, int H, int W, int Hp, int Wp,
                                 long total /* N*Hp*Wp */, int *__restrict__ nonfinite) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long step = (long)gridDim.x * blockDim.x;
    const long HW = (long)H * W;
    bool bad = false;
    for (; i < total; i += step) {
        int wp = (int)(i % Wp);
        long t = i / Wp;
        int hp = (int)(t % Hp);
        long n = t / Hp;
        int h = hp - 3, w = wp - 3;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
            const float *s = x + n * 3 * HW + (long)h * W + w;
            v0 = s[0];
            v1 = s[HW];
            v2 = s[2 * HW];
            // |v| < inf is false for inf and NaN.  (f16 mode: a finite pixel beyond the f16 range becomes inf in the cast below and
            // is flagged as well: the conversion is what the stem convolves)
            bad |= !(__builtin_fabsf((float)(T)v0) < INFINITY && __builtin_fabsf((float)(T)v1) < INFINITY && __builtin_fabsf((float)(T)v2) < INFINITY);
        }
        typedef T out4 __attribute__((ext_vector_type(4)));
        out4 o = {(T)v0, (T)v1, (T)v2, (T)0.f};
        reinterpret_cast<out4 *>(y)[i] = o;
    }
    // A non-finite pixel makes the reference raise: the NaN / inf spreads through every convolution (torch's relu and max_pool2d
    // keep NaN) into the RPN logits, torch.sort ranks NaN first, and _clip_box asserts on the selected boxes (frcnn.py:148).
    // The ReLU epilogues here (v_max / v_pk_max) return the non-NaN operand, so the flag is raised at the source instead.
    if (nonfinite && bad) atomicOr(nonfinite, 1);
}

// ---------------------------------------------------------------------------
// 3x3 stride-2 max-pool on NHWC.  caffe: pad 0 + ceil_mode (window clipped at the bottom/right
// edge; torch guarantees the last window starts inside the input); else pad 1 floor mode.
template <typename T>
__global__ void maxpool3x3s2_kernel(const T *__restrict__ x, T *__restrict__ y, int H, int W, int C, int Ho, int Wo,
                                    int pad, long total /* N*Ho*Wo*(C/VN) */) {
    typedef typename V16<T>::type vec;
    constexpr int VN = V16<T>::N;
    const int cv = C / VN;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long step = (long)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        int c = (int)(i % cv);
        long t = i / cv;
        int wo = (int)(t % Wo);
        t /= Wo;
        int ho = (int)(t % Ho);
        long n = t / Ho;
        int h0 = ho * 2 - pad, w0 = wo * 2 - pad;
        float m[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) m[e] = -INFINITY;
        for (int dh = 0; dh < 3; ++dh) {
            int h = h0 + dh;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int dw = 0; dw < 3; ++dw) {
                int w = w0 + dw;
                if ((unsigned)w >= (unsigned)W) continue;
                vec v = reinterpret_cast<const vec *>(x + ((n * H + h) * W + w) * C)[c];
#pragma unroll
                for (int e = 0; e < VN; ++e) m[e] = fmaxf(m[e], (float)v[e]);
            }
        }
        vec o;
#pragma unroll
        for (int e = 0; e < VN; ++e) o[e] = (T)m[e];
        reinterpret_cast<vec *>(y + ((n * Ho + ho) * Wo + wo) * C)[c] = o;
    }
}

// ---------------------------------------------------------------------------
// RoIPool, torchvision semantics (see oracle/tv_ops.c): one workgroup per (RoI, group of RPW output rows);
// the bin geometry is wave-uniform, lanes run over (pw, 16-B channel chunk).
template <typename T, int RPW>
__global__ void roi_pool_kernel(const T *__restrict__ feat, const float *__restrict__ rois, T *__restrict__ out, int H,
                                int W, int C, int P, float scale) {
    typedef typename V16<T>::type vec;
    constexpr int VN = V16<T>::N;
    const int groups = (P + RPW - 1) / RPW;
    const int k = blockIdx.x / groups, ph0 = (blockIdx.x % groups) * RPW;
    const float *r = rois + 5 * (long)k;
    const int b = (int)r[0];
    const int rsw = (int)roundf(r[1] * scale), rsh = (int)roundf(r[2] * scale);
    const int rew = (int)roundf(r[3] * scale), reh = (int)roundf(r[4] * scale);
    const int roi_w = max(rew - rsw + 1, 1), roi_h = max(reh - rsh + 1, 1);
    const float bin_h = (float)roi_h / (float)P, bin_w = (float)roi_w / (float)P;
    const int cv = C / VN;
    const T *fb = feat + (long)b * H * W * C;
    for (int ph = ph0; ph < min(ph0 + RPW, P); ++ph) {
    int hs = (int)floorf((float)ph * bin_h), he = (int)ceilf((float)(ph + 1) * bin_h);
    hs = min(max(hs + rsh, 0), H);
    he = min(max(he + rsh, 0), H);
    T *ob = out + ((long)k * P + ph) * P * C;
    for (int i = threadIdx.x; i < P * cv; i += blockDim.x) {
        int pw = i / cv, c = i - pw * cv;
        int ws = (int)floorf((float)pw * bin_w), we = (int)ceilf((float)(pw + 1) * bin_w);
        ws = min(max(ws + rsw, 0), W);
        we = min(max(we + rsw, 0), W);
        const bool empty = (he <= hs) || (we <= ws);
        vec o;
        if constexpr (sizeof(T) == 2) {
            // f16: the maximum taken on the packed halves (v_pk_max_f16: 4 instructions per 16-byte load instead of 24 for
            // convert / compare / select per element); same value as the fp32 form: a NaN never wins, an all-NaN bin gives -inf
            // (what -FLT_MAX rounds to)
#pragma unroll
            for (int e = 0; e < VN; ++e) o[e] = empty ? (T)0.f : (T)(-__builtin_inff());
            for (int h = hs; h < he; ++h)
                for (int w = ws; w < we; ++w) o = __builtin_elementwise_max(o, reinterpret_cast<const vec *>(fb + ((long)h * W + w) * C)[c]);
        } else {
            float m[VN];
#pragma unroll
            for (int e = 0; e < VN; ++e) m[e] = empty ? 0.f : -FLT_MAX;
            for (int h = hs; h < he; ++h)
                for (int w = ws; w < we; ++w) {
                    vec v = reinterpret_cast<const vec *>(fb + ((long)h * W + w) * C)[c];
#pragma unroll
                    for (int e = 0; e < VN; ++e) m[e] = ((float)v[e] > m[e]) ? (float)v[e] : m[e];
                }
#pragma unroll
            for (int e = 0; e < VN; ++e) o[e] = (T)m[e];
        }
        reinterpret_cast<vec *>(ob + (long)pw * C)[c] = o;
    }
    }
}

// (Measured and not kept, round 3: a SEPARABLE form -- one workgroup per RoI, a thread walks the 14 bins of its (pw, channel chunk)
// items top to bottom, takes each input row's maximum over the bin's columns once and carries the boundary row into the next bin:
// rows x (roi_w + 14) cell reads instead of (roi_h + 14) x (roi_w + 14).  Bit-identical, and slower: 1.45 - 1.50 ms against 1.20 -
// 1.21 ms per 9600 RoIs of the bench (mean proposal 126 x 112 pixels = 8 x 7 cells: windows of one or two cells, so little is
// shared, and the walk down the bins is a chain of dependent loads where the form above has 14 x the workgroups in flight).)

// ---------------------------------------------------------------------------
// out[k][c] = (sum_s x[k][s][c]) / S in f32.  One workgroup per RoI, 4 row-groups x 64 chunk-lanes
// accumulate partial sums that are combined through LDS in a fixed order (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void mean_pool_kernel(const T *__restrict__ x, float *__restrict__ out, int S, int C) {
    typedef typename V16<T>::type vec;
    constexpr int VN = V16<T>::N;
    __shared__ float part[4][64 * VN];
    const int k = blockIdx.x;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int cv = C / VN;
    const T *xb = x + (long)k * S * C;
    for (int c0 = 0; c0 < cv; c0 += 64) {
        const int c = c0 + lane;
        float acc[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[e] = 0.f;
        if (c < cv) {
            for (int s = grp; s < S; s += 4) {
                vec v = reinterpret_cast<const vec *>(xb + (long)s * C)[c];
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[e] += (float)v[e];
            }
        }
#pragma unroll
        for (int e = 0; e < VN; ++e) part[grp][lane * VN + e] = acc[e];
        __syncthreads();
        if (grp == 0 && c < cv) {
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                float sum = ((part[0][lane * VN + e] + part[1][lane * VN + e]) + part[2][lane * VN + e]) +
                            part[3][lane * VN + e];
                out[(long)k * C + c * VN + e] = sum / (float)S;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
int launch_stem_pack(const float *x, void *y, int N, int H, int W, int Hp, int Wp, vk_dtype dt, hipStream_t s, int32_t *nonfinite) {
    long total = (long)N * Hp * Wp;
    int g = grid_for(total, 256);
    if (dt == VK_F16)
        hipLaunchKernelGGL(stem_pack_kernel<_Float16>, dim3(g), dim3(256), 0, s, x, (_Float16 *)y, H, W, Hp, Wp, total, nonfinite);
    else
        hipLaunchKernelGGL(stem_pack_kernel<float>, dim3(g), dim3(256), 0, s, x, (float *)y, H, W, Hp, Wp, total, nonfinite);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

int launch_maxpool(const void *x, void *y, int N, int H, int W, int C, int caffe, vk_dtype dt, hipStream_t s) {
    int Ho, Wo;
    const int pad = caffe ? 0 : 1;
    if (caffe) {
        Ho = (H - 3 + 1) / 2 + 1;   // ceil((H-3)/2)+1
        Wo = (W - 3 + 1) / 2 + 1;
        if ((Ho - 1) * 2 >= H) --Ho;   // last window must start inside the input
        if ((Wo - 1) * 2 >= W) --Wo;
    } else {
        Ho = (H + 2 - 3) / 2 + 1;
        Wo = (W + 2 - 3) / 2 + 1;
    }
    const int vn = dt == VK_F16 ? 8 : 4;
    VK_REQUIRE(C % vn == 0, VK_EINVAL, "maxpool: C=%d must be a multiple of %d", C, vn);
    long total = (long)N * Ho * Wo * (C / vn);
    int g = grid_for(total, 256);
    if (dt == VK_F16)
        hipLaunchKernelGGL(maxpool3x3s2_kernel<_Float16>, dim3(g), dim3(256), 0, s, (const _Float16 *)x, (_Float16 *)y,
                           H, W, C, Ho, Wo, pad, total);
    else
        hipLaunchKernelGGL(maxpool3x3s2_kernel<float>, dim3(g), dim3(256), 0, s, (const float *)x, (float *)y, H, W, C,
                           Ho, Wo, pad, total);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

}  // namespace vk

using namespace vk;

extern "C" {

void vk_stem_out_hw(int H, int W, int caffe_maxpool, int *Ho, int *Wo) {
    int h1 = (H + 6 - 7) / 2 + 1, w1 = (W + 6 - 7) / 2 + 1;   // 7x7 s2 p3
    int h2, w2;
    if (caffe_maxpool) {
        h2 = (h1 - 3 + 1) / 2 + 1;
        w2 = (w1 - 3 + 1) / 2 + 1;
        if ((h2 - 1) * 2 >= h1) --h2;
        if ((w2 - 1) * 2 >= w1) --w2;
    } else {
        h2 = (h1 + 2 - 3) / 2 + 1;
        w2 = (w1 + 2 - 3) / 2 + 1;
    }
    *Ho = h2;
    *Wo = w2;
}

int vk_nchw_to_nhwc(const float *x, int N, int C, int H, int W, void *y, vk_dtype dt, void *stream) {
    long total = (long)N * C * H * W;
    VK_REQUIRE(total > 0, VK_EINVAL, "nchw_to_nhwc: empty tensor");
    int g = grid_for(total, 256);
    hipStream_t s = (hipStream_t)stream;
    if (dt == VK_F16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<_Float16>, dim3(g), dim3(256), 0, s, x, (_Float16 *)y, C, H * W, total);
    else if (dt == VK_F32)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(g), dim3(256), 0, s, x, (float *)y, C, H * W, total);
    else
        VK_REQUIRE(false, VK_EINVAL, "nchw_to_nhwc: bad dtype");
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

int vk_nhwc_to_nchw(const void *x, int N, int C, int H, int W, float *y, vk_dtype dt, void *stream) {
    long total = (long)N * C * H * W;
    VK_REQUIRE(total > 0, VK_EINVAL, "nhwc_to_nchw: empty tensor");
    int g = grid_for(total, 256);
    hipStream_t s = (hipStream_t)stream;
    if (dt == VK_F16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<_Float16>, dim3(g), dim3(256), 0, s, (const _Float16 *)x, y, C, H * W, total);
    else if (dt == VK_F32)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(g), dim3(256), 0, s, (const float *)x, y, C, H * W, total);
    else
        VK_REQUIRE(false, VK_EINVAL, "nhwc_to_nchw: bad dtype");
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

int vk_maxpool3x3s2(const void *x, int N, int H, int W, int C, int caffe, void *y, vk_dtype dt, void *stream) {
    VK_REQUIRE(dt == VK_F16 || dt == VK_F32, VK_EINVAL, "maxpool: bad dtype");
    VK_REQUIRE(H >= 3 && W >= 3, VK_EINVAL, "maxpool: input %dx%d too small", H, W);
    return launch_maxpool(x, y, N, H, W, C, caffe, dt, (hipStream_t)stream);
}

int vk_roi_pool(const void *feat, int N, int H, int W, int C, const float *rois, int K, float spatial_scale, int P,
                void *out, vk_dtype dt, void *stream) {
    (void)N;
    VK_REQUIRE(dt == VK_F16 || dt == VK_F32, VK_EINVAL, "roi_pool: bad dtype");
    const int vn = dt == VK_F16 ? 8 : 4;
    VK_REQUIRE(C % vn == 0 && P > 0, VK_EINVAL, "roi_pool: C=%d must be a multiple of %d", C, vn);
    if (K == 0) return VK_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dt == VK_F16)
        hipLaunchKernelGGL((roi_pool_kernel<_Float16, 1>), dim3(K * P), dim3(256), 0, s, (const _Float16 *)feat, rois,
                           (_Float16 *)out, H, W, C, P, spatial_scale);
    else
        hipLaunchKernelGGL((roi_pool_kernel<float, 1>), dim3(K * P), dim3(256), 0, s, (const float *)feat, rois, (float *)out,
                           H, W, C, P, spatial_scale);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

int vk_mean_pool(const void *x, int K, int S, int C, float *out, vk_dtype dt, void *stream) {
    VK_REQUIRE(dt == VK_F16 || dt == VK_F32, VK_EINVAL, "mean_pool: bad dtype");
    const int vn = dt == VK_F16 ? 8 : 4;
    VK_REQUIRE(C % vn == 0 && S > 0, VK_EINVAL, "mean_pool: C=%d must be a multiple of %d", C, vn);
    if (K == 0) return VK_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dt == VK_F16)
        hipLaunchKernelGGL(mean_pool_kernel<_Float16>, dim3(K), dim3(256), 0, s, (const _Float16 *)x, out, S, C);
    else
        hipLaunchKernelGGL(mean_pool_kernel<float>, dim3(K), dim3(256), 0, s, (const float *)x, out, S, C);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

}  // extern "C"
