// N4: the FPN-side kernels north_star names -- RoIAlign over a feature pyramid, level assignment, the
// neck's top-down nearest-2x upsample + add, the top blocks' stride-2 pick and ReLU copy.  All HBM-bound, NHWC.
//   roi_align_kernel     torchvision roi_align semantics (`aligned` as detectron2 ROIAlignV2), multi-level: each RoI reads
//                        the map of its level (assign_boxes_to_levels, frcnn.py:444-460; ROIPooler loop :1214-1222)
//   assign_levels_kernel floor(canonical_level + log2(sqrt(area)/canonical_size + 1e-8)) clamped, minus min_level
//   upsample2x_add       y = lateral + nearest2x(top)   (detectron2 FPN top-down path; absent from the reference)
//   subsample2           LastLevelMaxPool: max_pool2d(k=1, s=2) = every second pixel (frcnn.py:835-836)
//   relu_copy            the ReLU between LastLevelP6P7's two convs (frcnn.py:852-853)
#include "vk_common.h"

namespace vk {

constexpr int FPN_MAX_LEVELS = 6;

struct PyramidArgs {
    const void *map[FPN_MAX_LEVELS];
    int H[FPN_MAX_LEVELS], W[FPN_MAX_LEVELS];
    float scale[FPN_MAX_LEVELS];
    int levels;
};

template <typename T>
__device__ __forceinline__ float fld(const T *p) { return (float)*p; }

// one workgroup per (RoI, output bin); lanes over channels.  fp32 arithmetic in the published op order.
template <typename T>
__global__ __launch_bounds__(256) void roi_align_kernel(PyramidArgs py, int C, const float *__restrict__ rois,
                                                       const int32_t *__restrict__ levels, int P, int sampling_ratio, int aligned,
                                                       T *__restrict__ out) {
    const int k = blockIdx.x / (P * P), bin = blockIdx.x % (P * P), ph = bin / P, pw = bin % P;
    const float *r = rois + 5 * (long)k;
    const int lv = levels ? levels[k] : 0;
    const T *map = (const T *)py.map[lv];
    const int H = py.H[lv], W = py.W[lv];
    const float s = py.scale[lv], off = aligned ? 0.5f : 0.f;
    const int b = (int)r[0];
    const float sw = r[1] * s - off, sh = r[2] * s - off, ew = r[3] * s - off, eh = r[4] * s - off;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) {
        rw = fmaxf(rw, 1.f);
        rh = fmaxf(rh, 1.f);
    }
    const float bh = rh / (float)P, bw = rw / (float)P;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
    const float count = (float)max(gh * gw, 1);
    const T *base = map + (long)b * H * W * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int iy = 0; iy < gh; ++iy) {
            float y = sh + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                float yy = y;
                if (yy < -1.0f || yy > (float)H || x < -1.0f || x > (float)W) continue;
                if (yy <= 0.f) yy = 0.f;
                if (x <= 0.f) x = 0.f;
                int yl = (int)yy, xl = (int)x, yh, xh;
                if (yl >= H - 1) {
                    yh = yl = H - 1;
                    yy = (float)yl;
                } else
                    yh = yl + 1;
                if (xl >= W - 1) {
                    xh = xl = W - 1;
                    x = (float)xl;
                } else
                    xh = xl + 1;
                const float ly = yy - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                acc += w1 * fld(base + ((long)yl * W + xl) * C + c) + w2 * fld(base + ((long)yl * W + xh) * C + c) +
                       w3 * fld(base + ((long)yh * W + xl) * C + c) + w4 * fld(base + ((long)yh * W + xh) * C + c);
            }
        }
        out[((long)k * P * P + bin) * C + c] = (T)(acc / count);
    }
}

// Vector form (C a multiple of 16 B worth of channels): one workgroup per RoI; a group of C/V lanes (V = 16 B of
// channels per lane) owns one output bin at a time, so every tap of the bilinear stencil is one coalesced row read and
// every output one coalesced row write.  Same arithmetic and op order as roi_align_kernel.
template <typename T>
__global__ __launch_bounds__(256) void roi_align_vec_kernel(PyramidArgs py, int C, const float *__restrict__ rois,
                                                           const int32_t *__restrict__ levels, int P, int sampling_ratio, int aligned,
                                                           T *__restrict__ out) {
    constexpr int V = 16 / sizeof(T);
    typedef T vecT __attribute__((ext_vector_type(V)));
    const int k = blockIdx.x;
    const float *r = rois + 5 * (long)k;
    const int lv = levels ? levels[k] : 0;
    const T *map = (const T *)py.map[lv];
    const int H = py.H[lv], W = py.W[lv];
    const float s = py.scale[lv], off = aligned ? 0.5f : 0.f;
    const int b = (int)r[0];
    const float sw = r[1] * s - off, sh = r[2] * s - off, ew = r[3] * s - off, eh = r[4] * s - off;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) {
        rw = fmaxf(rw, 1.f);
        rh = fmaxf(rh, 1.f);
    }
    const float bh = rh / (float)P, bw = rw / (float)P;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
    const float count = (float)max(gh * gw, 1);
    const T *base = map + (long)b * H * W * C;
    const int lanes = C / V;                         // lanes per bin (<= 256)
    const int groups = 256 / lanes, grp = threadIdx.x / lanes, c0 = (threadIdx.x % lanes) * V;
    if (grp >= groups) return;
    for (int bin = grp; bin < P * P; bin += groups) {
        const int ph = bin / P, pw = bin % P;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int iy = 0; iy < gh; ++iy) {
            const float y0 = sh + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                float yy = y0;
                if (yy < -1.0f || yy > (float)H || x < -1.0f || x > (float)W) continue;
                if (yy <= 0.f) yy = 0.f;
                if (x <= 0.f) x = 0.f;
                int yl = (int)yy, xl = (int)x, yh, xh;
                if (yl >= H - 1) {
                    yh = yl = H - 1;
                    yy = (float)yl;
                } else
                    yh = yl + 1;
                if (xl >= W - 1) {
                    xh = xl = W - 1;
                    x = (float)xl;
                } else
                    xh = xl + 1;
                const float ly = yy - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                const vecT v1 = *reinterpret_cast<const vecT *>(base + ((long)yl * W + xl) * C + c0);
                const vecT v2 = *reinterpret_cast<const vecT *>(base + ((long)yl * W + xh) * C + c0);
                const vecT v3 = *reinterpret_cast<const vecT *>(base + ((long)yh * W + xl) * C + c0);
                const vecT v4 = *reinterpret_cast<const vecT *>(base + ((long)yh * W + xh) * C + c0);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += w1 * (float)v1[e] + w2 * (float)v2[e] + w3 * (float)v3[e] + w4 * (float)v4[e];
            }
        }
        vecT o;
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = (T)(acc[e] / count);
        *reinterpret_cast<vecT *>(out + ((long)k * P * P + bin) * C + c0) = o;
    }
}

// Separable form of the vector kernel (round 3; the default when the tables fit).  A bin's value is
//   (1/count) * sum over samples (iy, ix) of the bilinear interpolation at (y_iy, x_ix)
// and bilinear weights factor per axis, so it equals  (1/count) * sum_Y sum_X WY[Y] * WX[X] * F[Y][X]  with
// WY[Y] = the summed weights the samples' rows give pixel row Y (likewise WX): the same sample positions, validity test and
// clamping as roi_align_kernel, evaluated ONCE per (axis, bin index) by one thread into LDS, after which a bin reads each pixel
// of its footprint once -- (gh+1..gh+2) x (gw+1..gw+2) row reads instead of 4 * gh * gw (adaptive sampling at 800x1333 has gh, gw
// up to 6: 49-64 reads instead of 144).  fp32 sums in a different order than the per-sample form: equal to ~1e-7 relative
// (tests: <= 1e-5 in fp32).  A RoI whose footprint exceeds the table (RA_T rows) takes the per-sample loop.
constexpr int RA_T = 12, RA_MAXP = 16;
template <typename T>
__global__ __launch_bounds__(256) void roi_align_sep_kernel(PyramidArgs py, int C, const float *__restrict__ rois,
                                                           const int32_t *__restrict__ levels, int P, int sampling_ratio, int aligned,
                                                           int groups, T *__restrict__ out) {
    constexpr int V = 16 / sizeof(T);
    typedef T vecT __attribute__((ext_vector_type(V)));
    __shared__ float wtab[2][RA_MAXP][RA_T];
    __shared__ int first[2][RA_MAXP], cnt[2][RA_MAXP];
    const int k = blockIdx.x;
    const float *r = rois + 5 * (long)k;
    const int lv = levels ? levels[k] : 0;
    const T *map = (const T *)py.map[lv];
    const int H = py.H[lv], W = py.W[lv];
    const float s = py.scale[lv], off = aligned ? 0.5f : 0.f;
    const int b = (int)r[0];
    const float sw = r[1] * s - off, sh = r[2] * s - off, ew = r[3] * s - off, eh = r[4] * s - off;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) {
        rw = fmaxf(rw, 1.f);
        rh = fmaxf(rh, 1.f);
    }
    const float bh = rh / (float)P, bw = rw / (float)P;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
    const float count = (float)max(gh * gw, 1);
    const T *base = map + (long)b * H * W * C;
    const int lanes = C / V;
    const int grp = threadIdx.x / lanes, c0 = (threadIdx.x % lanes) * V;
    // rows a bin index can touch: its samples span bsz * (g-1)/g, so at most floor(span) + 2 distinct floor rows plus the upper
    // neighbour of the last.  The tables serve a RoI when that fits AND is not more reads than the samples' own 4 taps (a fixed
    // sampling_ratio over large bins leaves gaps between the samples); degenerate boxes (negative or NaN sizes) keep the per-sample loop.
    const float nyb = gh > 1 ? floorf(bh * (float)(gh - 1) / (float)gh) + 3.f : 2.f;
    const float nxb = gw > 1 ? floorf(bw * (float)(gw - 1) / (float)gw) + 3.f : 2.f;
    const bool tables = bh >= 0.f && bw >= 0.f && nyb <= (float)RA_T && nxb <= (float)RA_T && nyb * nxb <= 4.f * (float)gh * (float)gw + 4.f;
    if (!tables) {                                 // (uniform per workgroup)
        if (grp >= groups) return;
        for (int bin = grp; bin < P * P; bin += groups) {
            const int ph = bin / P, pw = bin % P;
            float acc[V];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = 0.f;
            for (int iy = 0; iy < gh; ++iy) {
                const float y0 = sh + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                for (int ix = 0; ix < gw; ++ix) {
                    float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                    float yy = y0;
                    if (yy < -1.0f || yy > (float)H || x < -1.0f || x > (float)W) continue;
                    if (yy <= 0.f) yy = 0.f;
                    if (x <= 0.f) x = 0.f;
                    int yl = (int)yy, xl = (int)x, yh, xh;
                    if (yl >= H - 1) {
                        yh = yl = H - 1;
                        yy = (float)yl;
                    } else
                        yh = yl + 1;
                    if (xl >= W - 1) {
                        xh = xl = W - 1;
                        x = (float)xl;
                    } else
                        xh = xl + 1;
                    const float ly = yy - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
                    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                    const vecT v1 = *reinterpret_cast<const vecT *>(base + ((long)yl * W + xl) * C + c0);
                    const vecT v2 = *reinterpret_cast<const vecT *>(base + ((long)yl * W + xh) * C + c0);
                    const vecT v3 = *reinterpret_cast<const vecT *>(base + ((long)yh * W + xl) * C + c0);
                    const vecT v4 = *reinterpret_cast<const vecT *>(base + ((long)yh * W + xh) * C + c0);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] += w1 * (float)v1[e] + w2 * (float)v2[e] + w3 * (float)v3[e] + w4 * (float)v4[e];
                }
            }
            vecT o;
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] = (T)(acc[e] / count);
            *reinterpret_cast<vecT *>(out + ((long)k * P * P + bin) * C + c0) = o;
        }
        return;
    }
    // ---- per-axis weight tables: thread (axis, p) walks its bin index's samples in order ----
    if (threadIdx.x < 2 * P) {
        const int axis = threadIdx.x / P, p = threadIdx.x % P;
        const float start = axis ? sw : sh, bsz = axis ? bw : bh;
        const int g = axis ? gw : gh, lim = axis ? W : H;
        float *w = wtab[axis][p];
#pragma unroll
        for (int t = 0; t < RA_T; ++t) w[t] = 0.f;
        int f = -1, n = 0;
        for (int i = 0; i < g; ++i) {
            float v = start + (float)p * bsz + ((float)i + 0.5f) * bsz / (float)g;
            if (v < -1.0f || v > (float)lim) continue;
            if (v <= 0.f) v = 0.f;
            int lo = (int)v, hi;
            if (lo >= lim - 1) {
                hi = lo = lim - 1;
                v = (float)lo;
            } else
                hi = lo + 1;
            const float l = v - (float)lo, h = 1.f - l;
            if (f < 0) f = lo;                       // positions ascend with i: the first valid sample has the lowest row
            w[lo - f] += h;
            w[hi - f] += l;
            n = hi - f + 1;
        }
        first[axis][p] = max(f, 0);
        cnt[axis][p] = n;
    }
    __syncthreads();
    if (grp >= groups) return;
    const float inv = 1.f / count;
    for (int bin = grp; bin < P * P; bin += groups) {
        const int ph = bin / P, pw = bin % P;
        const int nY = cnt[0][ph], nX = cnt[1][pw];
        const T *p0 = base + ((long)first[0][ph] * W + first[1][pw]) * C + c0;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int yy = 0; yy < nY; ++yy) {
            const float wy = wtab[0][ph][yy];
            const T *prow = p0 + (long)yy * W * C;
            for (int xx = 0; xx < nX; xx += 4) {   // four independent row reads in flight; columns past nX: weight 0, address clamped
                vecT v[4];
                float wx[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int xi = min(xx + q, nX - 1);
                    wx[q] = xx + q < nX ? wy * wtab[1][pw][xi] : 0.f;
                    v[q] = *reinterpret_cast<const vecT *>(prow + (long)xi * C);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] += wx[q] * (float)v[q][e];
            }
        }
        vecT o;
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = (T)(acc[e] * inv);
        *reinterpret_cast<vecT *>(out + ((long)k * P * P + bin) * C + c0) = o;
    }
}

__global__ void assign_levels_kernel(const float *__restrict__ boxes, int ld, int K, int min_level, int max_level, float canonical_size,
                                     int canonical_level, int32_t *__restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float *b = boxes + (long)k * ld;
    const float area = (b[2] - b[0]) * (b[3] - b[1]);
    float lv = floorf((float)canonical_level + log2f(sqrtf(area) / canonical_size + 1e-8f));
    lv = fminf(fmaxf(lv, (float)min_level), (float)max_level);      // NaN (negative area) clamps like torch.clamp: stays NaN -> cast
    out[k] = (int32_t)lv - min_level;
}

template <typename T>
__global__ void upsample2x_add_kernel(const T *__restrict__ lat, const T *__restrict__ top, T *__restrict__ y, int H, int W, int Ht, int Wt,
                                      int C, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const long px = i / C;
    const int w = (int)(px % W), h = (int)((px / W) % H);
    const long n = px / ((long)W * H);
    const int ht = min(h >> 1, Ht - 1), wt = min(w >> 1, Wt - 1);
    y[i] = (T)((float)lat[i] + (float)top[((n * Ht + ht) * Wt + wt) * C + c]);
}

// 16 bytes per lane (8 channels of a 16-bit type) for channel counts that are multiples of 8: the scalar form above moves 2 bytes
// per lane (P2 of 32 images: 1.7 ms for 2.7 GB; this one runs at the copy rate).  Same arithmetic per element: fp32 add, one rounding.
template <typename T>
__global__ void upsample2x_add_vec8_kernel(const T *__restrict__ lat, const T *__restrict__ top, T *__restrict__ y, int H, int W, int Ht, int Wt,
                                           int C8, long total8) {
    typedef T vec8 __attribute__((ext_vector_type(8)));
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int c = (int)(i % C8);
    const long px = i / C8;
    const int w = (int)(px % W), h = (int)((px / W) % H);
    const long n = px / ((long)W * H);
    const int ht = min(h >> 1, Ht - 1), wt = min(w >> 1, Wt - 1);
    const vec8 a = reinterpret_cast<const vec8 *>(lat)[i];
    const vec8 b = reinterpret_cast<const vec8 *>(top)[((n * Ht + ht) * Wt + wt) * C8 + c];
    vec8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (T)((float)a[e] + (float)b[e]);
    reinterpret_cast<vec8 *>(y)[i] = o;
}

template <typename T>
__global__ void subsample2_kernel(const T *__restrict__ x, T *__restrict__ y, int H, int W, int Ho, int Wo, int C, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const long px = i / C;
    const int wo = (int)(px % Wo), ho = (int)((px / Wo) % Ho);
    const long n = px / ((long)Wo * Ho);
    y[i] = x[((n * H + 2 * ho) * W + 2 * wo) * C + c];
}

template <typename T>
__global__ void relu_copy_kernel(const T *__restrict__ x, T *__restrict__ y, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const float v = (float)x[i];
        y[i] = (T)(v > 0.f ? v : 0.f);
    }
}

}  // namespace vk

using namespace vk;

#define VK_DT_SWITCH(dt, CALL)                                                          \
    switch (dt) {                                                                       \
        case VK_F32: { typedef float T; CALL; } break;                                  \
        case VK_F16: { typedef _Float16 T; CALL; } break;                               \
        case VK_BF16: { typedef __bf16 T; CALL; } break;                                \
        default: VK_REQUIRE(false, VK_EINVAL, "dtype must be f32, f16 or bf16");        \
    }

extern "C" {

int vk_roi_align(const void *const *maps, const int32_t *Hs, const int32_t *Ws, const float *scales, int levels, int N, int C,
                 const float *rois, const int32_t *roi_levels, int K, int P, int sampling_ratio, int aligned, void *out, vk_dtype dt,
                 void *stream) {
    VK_REQUIRE(maps && Hs && Ws && scales && rois && out, VK_EINVAL, "roi_align: null argument");
    VK_REQUIRE(levels >= 1 && levels <= FPN_MAX_LEVELS && (levels == 1 || roi_levels), VK_EINVAL, "roi_align: 1..%d levels, level ids needed beyond one",
               FPN_MAX_LEVELS);
    VK_REQUIRE(N > 0 && C > 0 && P > 0 && K >= 0 && sampling_ratio >= 0, VK_EINVAL, "roi_align: bad sizes");
    if (K == 0) return VK_OK;
    PyramidArgs py;
    memset(&py, 0, sizeof(py));
    py.levels = levels;
    for (int i = 0; i < levels; ++i) {
        VK_REQUIRE(maps[i] && Hs[i] > 0 && Ws[i] > 0, VK_EINVAL, "roi_align: level %d is empty", i);
        py.map[i] = maps[i];
        py.H[i] = Hs[i];
        py.W[i] = Ws[i];
        py.scale[i] = scales[i];
    }
    const int vec = 16 / (int)dtype_size(dt);
    if (C % vec == 0 && C / vec <= 256) {            // one workgroup per RoI, 16-byte lanes (the FPN case: C = 256)
        static const bool per_sample = getenv("VK_ROIALIGN_TABLES") && atoi(getenv("VK_ROIALIGN_TABLES")) == 0;   // A/B switch
        if (P <= RA_MAXP && !per_sample) {
            int groups = 256 / (C / vec);
            while ((P * P) % groups) --groups;       // no idle group in the last pass (P = 7, 32 lanes per bin: 7 groups)
            VK_DT_SWITCH(dt, hipLaunchKernelGGL(roi_align_sep_kernel<T>, dim3(K), dim3(256), 0, (hipStream_t)stream, py, C, rois, roi_levels, P,
                                                sampling_ratio, aligned, groups, (T *)out));
        } else {
            VK_DT_SWITCH(dt, hipLaunchKernelGGL(roi_align_vec_kernel<T>, dim3(K), dim3(256), 0, (hipStream_t)stream, py, C, rois, roi_levels, P,
                                                sampling_ratio, aligned, (T *)out));
        }
    } else {
        const dim3 grid((unsigned)((long)K * P * P)), block(C >= 256 ? 256 : 64);
        VK_DT_SWITCH(dt, hipLaunchKernelGGL(roi_align_kernel<T>, grid, block, 0, (hipStream_t)stream, py, C, rois, roi_levels, P, sampling_ratio,
                                            aligned, (T *)out));
    }
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

int vk_assign_levels(const float *boxes, int ld, int K, int min_level, int max_level, float canonical_box_size, int canonical_level,
                     int32_t *levels_out, void *stream) {
    VK_REQUIRE(boxes && levels_out && ld >= 4 && K >= 0 && min_level <= max_level, VK_EINVAL, "assign_levels: bad arguments");
    if (K == 0) return VK_OK;
    hipLaunchKernelGGL(assign_levels_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, boxes, ld, K, min_level, max_level,
                       canonical_box_size, canonical_level, levels_out);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

int vk_upsample2x_add(const void *lateral, const void *top, void *y, int N, int H, int W, int Ht, int Wt, int C, vk_dtype dt, void *stream) {
    VK_REQUIRE(lateral && top && y && N > 0 && H > 0 && W > 0 && C > 0, VK_EINVAL, "upsample2x_add: bad arguments");
    VK_REQUIRE(2 * Ht >= H && 2 * Wt >= W, VK_EINVAL, "upsample2x_add: the %dx%d top map does not cover %dx%d", Ht, Wt, H, W);
    const long total = (long)N * H * W * C;
    const bool aligned = (((uintptr_t)lateral | (uintptr_t)top | (uintptr_t)y) & 15) == 0;
    if (C % 8 == 0 && aligned && (dt == VK_F16 || dt == VK_BF16)) {
        const long total8 = total / 8;
        if (dt == VK_F16)
            hipLaunchKernelGGL(upsample2x_add_vec8_kernel<_Float16>, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                               (const _Float16 *)lateral, (const _Float16 *)top, (_Float16 *)y, H, W, Ht, Wt, C / 8, total8);
        else
            hipLaunchKernelGGL(upsample2x_add_vec8_kernel<__bf16>, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                               (const __bf16 *)lateral, (const __bf16 *)top, (__bf16 *)y, H, W, Ht, Wt, C / 8, total8);
        VK_CHECK_HIP(hipGetLastError());
        return VK_OK;
    }
    VK_DT_SWITCH(dt, hipLaunchKernelGGL(upsample2x_add_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                        (const T *)lateral, (const T *)top, (T *)y, H, W, Ht, Wt, C, total));
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

int vk_subsample2(const void *x, void *y, int N, int H, int W, int C, vk_dtype dt, void *stream) {
    VK_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0, VK_EINVAL, "subsample2: bad arguments");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)N * Ho * Wo * C;
    VK_DT_SWITCH(dt, hipLaunchKernelGGL(subsample2_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                        (const T *)x, (T *)y, H, W, Ho, Wo, C, total));
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

int vk_relu_copy(const void *x, void *y, long n, vk_dtype dt, void *stream) {
    VK_REQUIRE(x && y && n >= 0, VK_EINVAL, "relu_copy: bad arguments");
    if (n == 0) return VK_OK;
    VK_DT_SWITCH(dt, hipLaunchKernelGGL(relu_copy_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const T *)x,
                                        (T *)y, n));
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

}  // extern "C"
