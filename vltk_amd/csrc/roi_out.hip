// Box-head post-processing on gfx950: class / attribute soft-max + arg-max, the attribute
// branch's embedding gather, the arg-max class's box regression, and the per-image
// "class-max NMS with a threshold list" that selects the final detections.
//
// Replaces (reference vltk/modeling/frcnn.py):
//   FastRCNNOutputLayers.forward (arg-max, embedding, concat)     :1731-1734
//   ROIOutputs._predict_objs / _predict_attrs / _predict_boxes     :1242-1260
//   ROIOutputs.inference (threshold loop, scales, gathers)         :1262-1294
//   do_nms (+ torchvision.ops.nms)                                 :116-143
//
// The reference decodes all C class-specific boxes per RoI and then keeps only the arg-max
// class's box (`idxs = arange(R)*C + max_classes`, :128-129); here only that one box is
// decoded (same arithmetic, same result), so the [K, 4C] delta matrix never has to exist:
// `chosen_deltas_kernel` computes just the 4 needed rows of bbox_pred per RoI.  The
// reference's finite-ness assert (:148) is therefore evaluated on the boxes that are used.
//
// fp32 box math, reference op order, no FMA contraction (-ffp-contract=off for this file).
#include <cfloat>

#include "vk_common.h"

namespace vk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t desc_key32(float v) {
    uint32_t u = __float_as_uint(v);
    if (u == 0x80000000u) u = 0u;
    uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~asc;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// (value, index) arg-max with "first index wins" on ties
__device__ __forceinline__ void wave_argmax(float &v, int &i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(v, o);
        int oi = __shfl_xor(i, o);
        if (ov > v || (ov == v && oi < i)) {
            v = ov;
            i = oi;
        }
    }
}

// One wavefront per RoI.  softmax over the first n_soft logits, max/arg-max of the resulting
// probabilities over the first n_max (<= n_soft); optionally also the arg-max of the raw logits
// over n_soft entries (the attribute branch's `scores.max(-1)`, which includes background).
__global__ __launch_bounds__(256) void softmax_argmax_kernel(const float *__restrict__ logits, int ld, int K, int n_soft,
                                                             int n_max, float *__restrict__ prob_out,
                                                             int32_t *__restrict__ cls_out,
                                                             int32_t *__restrict__ raw_argmax_out) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= K) return;
    const float *x = logits + (long)k * ld;
    float m = -INFINITY;
    int mi = 0x7fffffff;
    for (int c = lane; c < n_soft; c += 64) {
        float v = x[c];
        if (v > m) {
            m = v;
            mi = c;
        }
    }
    wave_argmax(m, mi);
    float s = 0.f;
    for (int c = lane; c < n_soft; c += 64) s += expf(x[c] - m);
    s = wave_sum(s);
    float bp = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < n_max; c += 64) {
        float p = expf(x[c] - m) / s;
        if (p > bp) {
            bp = p;
            bi = c;
        }
    }
    wave_argmax(bp, bi);
    if (lane == 0) {
        prob_out[k] = bp;
        cls_out[k] = bi;
        if (raw_argmax_out) raw_argmax_out[k] = mi;
    }
}

// out[k] = [ (T)feat[k][0:F] | emb[cls[k]][0:E] ]   (torch.cat([roi_features, cls_emb], -1), :1734)
template <typename T>
__global__ void concat_embed_kernel(const float *__restrict__ feat, const T *__restrict__ emb,
                                    const int32_t *__restrict__ cls, int F, int E, T *__restrict__ out) {
    const int k = blockIdx.x;
    const float *f = feat + (long)k * F;
    T *o = out + (long)k * (F + E);
    for (int i = threadIdx.x; i < F; i += blockDim.x) o[i] = (T)f[i];
    if (E > 0) {
        const T *e = emb + (long)cls[k] * E;
        for (int i = threadIdx.x; i < E; i += blockDim.x) o[F + i] = e[i];
    }
}

// deltas[k][j] = bias[row] + <x[k], W[row]>, row = cls[k]*4 + j (class-specific) or j (agnostic).
// One wavefront per RoI; x is the first F entries of the row-major [K, ldx] matrix of type T.
template <typename T>
__global__ __launch_bounds__(256) void chosen_deltas_kernel(const T *__restrict__ x, int ldx, const T *__restrict__ w,
                                                            const float *__restrict__ bias,
                                                            const int32_t *__restrict__ cls, int agnostic, int F, int K,
                                                            float *__restrict__ out) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= K) return;
    const int row0 = agnostic ? 0 : cls[k] * 4;
    const T *xr = x + (long)k * ldx;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < F; i += 64) {
        float xv = (float)xr[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += xv * (float)w[(long)(row0 + j) * F + i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = wave_sum(acc[j]);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[(long)k * 4 + j] = acc[j] + bias[row0 + j];
    }
}


__device__ __forceinline__ void apply_deltas_roi(const float a[4], const float d[4], float wx, float wy, float ww,
                                                 float wh, float clampv, float o[4]) {
    float widths = a[2] - a[0];
    float heights = a[3] - a[1];
    float ctr_x = a[0] + 0.5f * widths;
    float ctr_y = a[1] + 0.5f * heights;
    float dx = d[0] / wx;
    float dy = d[1] / wy;
    float dw = d[2] / ww;
    float dh = d[3] / wh;
    dw = dw > clampv ? clampv : dw;
    dh = dh > clampv ? clampv : dh;
    float pcx = dx * widths + ctr_x;
    float pcy = dy * heights + ctr_y;
    float pw = expf(dw) * widths;
    float ph = expf(dh) * heights;
    o[0] = pcx - 0.5f * pw;
    o[1] = pcy - 0.5f * ph;
    o[2] = pcx + 0.5f * pw;
    o[3] = pcy + 0.5f * ph;
}

// One workgroup per image.  LDS: keys[Rp2] u64 | box[R][4] f32 | removed[R] i32 | kept[D] i32
__global__ __launch_bounds__(256) void roi_final_kernel(RoiFinalArgs a, int Rp2) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem_raw);
    float *sbox = reinterpret_cast<float *>(keys + Rp2);
    int *removed = reinterpret_cast<int *>(sbox + (size_t)a.R * 4);
    int *kept = removed + a.R;

    const int n = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
    const int cnt = min(a.counts[n], a.R);
    const float img_h = (float)a.image_hw[2 * n], img_w = (float)a.image_hw[2 * n + 1];
    const long k0 = (long)n * a.R;

    // 1. arg-max class's box: decode + clip (do_nms :116-129)
    bool bad = false;
    for (int r = tid; r < Rp2; r += T) {
        if (r < cnt) {
            const float *p = a.proposals + (k0 + r) * 4;
            float pr[4] = {p[0], p[1], p[2], p[3]};
            const float *dp = a.box_deltas + (k0 + r) * a.ld_box + (a.delta_mode == 0 ? a.obj_cls[k0 + r] * 4 : 0);
            float d[4] = {dp[0], dp[1], dp[2], dp[3]};
            float b[4];
            apply_deltas_roi(pr, d, a.wx, a.wy, a.ww, a.wh, a.clampv, b);
            if (!(isfinite(b[0]) && isfinite(b[1]) && isfinite(b[2]) && isfinite(b[3]))) bad = true;
            b[0] = fminf(fmaxf(b[0], 0.f), img_w);
            b[1] = fminf(fmaxf(b[1], 0.f), img_h);
            b[2] = fminf(fmaxf(b[2], 0.f), img_w);
            b[3] = fminf(fmaxf(b[3], 0.f), img_h);
            sbox[4 * r + 0] = b[0];
            sbox[4 * r + 1] = b[1];
            sbox[4 * r + 2] = b[2];
            sbox[4 * r + 3] = b[3];
            keys[r] = ((unsigned long long)desc_key32(a.obj_prob[k0 + r]) << 32) | (uint32_t)r;
        } else {
            keys[r] = ~0ull;
        }
    }
    if (bad) atomicOr(a.nonfinite, 1);
    __syncthreads();

    // 2. rank by probability, descending, ties -> lower RoI index (stable)
    for (int k2 = 2; k2 <= Rp2; k2 <<= 1)
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (int i = tid; i < Rp2; i += T) {
                int ixj = i ^ j2;
                if (ixj > i) {
                    unsigned long long x = keys[i], y = keys[ixj];
                    bool up = (i & k2) == 0;
                    if ((x > y) == up) {
                        keys[i] = y;
                        keys[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }

    // 3. greedy NMS per threshold until the kept count lands in [mind, maxd] (:1274-1278)
    int nk = 0;
    for (int ti = 0; ti < a.n_thresh; ++ti) {
        const double thr = a.thresh[ti];
        for (int i = tid; i < cnt; i += T) removed[i] = 0;
        nk = 0;
        for (int i = 0; i < cnt; ++i) {
            __syncthreads();
            if (removed[i]) continue;
            if (tid == 0) kept[nk] = i;
            ++nk;
            if (nk == a.maxd) break;   // keep[:maxd]
            const int ri = (int)(keys[i] & 0xFFFFFFFFull);
            const float ix1 = sbox[4 * ri], iy1 = sbox[4 * ri + 1], ix2 = sbox[4 * ri + 2], iy2 = sbox[4 * ri + 3];
            const float ia = (ix2 - ix1) * (iy2 - iy1);
            for (int j = i + 1 + tid; j < cnt; j += T) {
                if (removed[j]) continue;
                const int rj = (int)(keys[j] & 0xFFFFFFFFull);
                const float jx1 = sbox[4 * rj], jy1 = sbox[4 * rj + 1], jx2 = sbox[4 * rj + 2], jy2 = sbox[4 * rj + 3];
                const float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1);
                const float xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
                const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
                const float inter = w * h;
                const float ja = (jx2 - jx1) * (jy2 - jy1);
                const float ovr = inter / (ia + ja - inter);
                if ((double)ovr > thr) removed[j] = 1;
            }
        }
        __syncthreads();
        if (nk >= a.mind && nk <= a.maxd) break;
    }
    // (nk is computed identically by every thread; all LDS lives in the one dynamic array so its
    //  base stays 16-byte aligned)

    // 4. gather the outputs (rows >= nk are zero)
    const float sy = a.scales_yx ? a.scales_yx[2 * n] : 1.f, sx = a.scales_yx ? a.scales_yx[2 * n + 1] : 1.f;
    const long o0 = (long)n * a.D;
    for (int d = tid; d < a.D; d += T) {
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, op = 0.f, ap = 0.f;
        int64_t oc = 0, ac = 0, kid = 0;
        if (d < nk) {
            const int r = (int)(keys[kept[d]] & 0xFFFFFFFFull);
            b0 = sbox[4 * r];
            b1 = sbox[4 * r + 1];
            b2 = sbox[4 * r + 2];
            b3 = sbox[4 * r + 3];
            if (a.scales_yx) {   // boxes[:,0::2] *= scale_yx[1]; boxes[:,1::2] *= scale_yx[0]  (:1280-1283)
                b0 *= sx;
                b2 *= sx;
                b1 *= sy;
                b3 *= sy;
            }
            op = a.obj_prob[k0 + r];
            oc = a.obj_cls[k0 + r];
            ap = a.attr_prob ? a.attr_prob[k0 + r] : 0.f;
            ac = a.attr_cls ? a.attr_cls[k0 + r] : 0;
            kid = r;
        }
        a.out.boxes[(o0 + d) * 4 + 0] = b0;
        a.out.boxes[(o0 + d) * 4 + 1] = b1;
        a.out.boxes[(o0 + d) * 4 + 2] = b2;
        a.out.boxes[(o0 + d) * 4 + 3] = b3;
        a.out.obj_probs[o0 + d] = op;
        a.out.obj_ids[o0 + d] = oc;
        a.out.attr_probs[o0 + d] = ap;
        a.out.attr_ids[o0 + d] = ac;
        if (a.keep_ids) a.keep_ids[o0 + d] = kid;
    }
    if (tid == 0) a.out.preds_per_image[n] = nk;
    const int F4 = a.F / 4;
    for (int d = 0; d < a.D; ++d) {
        floatx4 *dst = reinterpret_cast<floatx4 *>(a.out.roi_features + (o0 + d) * a.F);
        if (d < nk) {
            const int r = (int)(keys[kept[d]] & 0xFFFFFFFFull);
            const floatx4 *src = reinterpret_cast<const floatx4 *>(a.features + (k0 + r) * a.F);
            for (int i = tid; i < F4; i += T) dst[i] = src[i];
        } else {
            const floatx4 z = {0.f, 0.f, 0.f, 0.f};
            for (int i = tid; i < F4; i += T) dst[i] = z;
        }
    }
}

// rois[k] = (batch index, x1, y1, x2, y2)   (convert_boxes_to_pooler_format, frcnn.py:426-441)
__global__ void make_rois_kernel(const float *__restrict__ boxes, int R, long total, float *__restrict__ rois) {
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    rois[5 * k] = (float)(k / R);
    rois[5 * k + 1] = boxes[4 * k];
    rois[5 * k + 2] = boxes[4 * k + 1];
    rois[5 * k + 3] = boxes[4 * k + 2];
    rois[5 * k + 4] = boxes[4 * k + 3];
}

int launch_make_rois(const float *boxes, int N, int R, float *rois, hipStream_t s) {
    long total = (long)N * R;
    hipLaunchKernelGGL(make_rois_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, boxes, R, total, rois);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

static int next_pow2i(int v) {
    int p = 2;
    while (p < v) p <<= 1;
    return p;
}

int launch_softmax_argmax(const float *logits, int ld, int K, int n_soft, int n_max, float *prob, int32_t *cls,
                          int32_t *raw_argmax, hipStream_t s) {
    if (K == 0) return VK_OK;
    hipLaunchKernelGGL(softmax_argmax_kernel, dim3(ceil_div(K, 4)), dim3(256), 0, s, logits, ld, K, n_soft, n_max, prob,
                       cls, raw_argmax);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

int launch_concat_embed(const float *feat, const void *emb, const int32_t *cls, int F, int E, int K, void *out,
                        vk_dtype dt, hipStream_t s) {
    if (K == 0) return VK_OK;
    if (dt == VK_F16)
        hipLaunchKernelGGL(concat_embed_kernel<_Float16>, dim3(K), dim3(256), 0, s, feat, (const _Float16 *)emb, cls, F, E,
                           (_Float16 *)out);
    else
        hipLaunchKernelGGL(concat_embed_kernel<float>, dim3(K), dim3(256), 0, s, feat, (const float *)emb, cls, F, E,
                           (float *)out);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

int launch_chosen_deltas(const void *x, int ldx, const void *w, const float *bias, const int32_t *cls, int agnostic, int F,
                         int K, float *out, vk_dtype dt, hipStream_t s) {
    if (K == 0) return VK_OK;
    if (dt == VK_F16)
        hipLaunchKernelGGL(chosen_deltas_kernel<_Float16>, dim3(ceil_div(K, 4)), dim3(256), 0, s, (const _Float16 *)x, ldx,
                           (const _Float16 *)w, bias, cls, agnostic, F, K, out);
    else
        hipLaunchKernelGGL(chosen_deltas_kernel<float>, dim3(ceil_div(K, 4)), dim3(256), 0, s, (const float *)x, ldx,
                           (const float *)w, bias, cls, agnostic, F, K, out);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

int launch_roi_final(RoiFinalArgs &a, int N, hipStream_t s) {
    VK_REQUIRE(a.R >= 1 && a.R <= 1024, VK_EINVAL, "roi_outputs: R=%d must be in 1..1024", a.R);
    VK_REQUIRE(a.D >= 1 && a.D <= a.R, VK_EINVAL, "roi_outputs: max_detections=%d must be in 1..R", a.D);
    VK_REQUIRE(a.F % 4 == 0, VK_EINVAL, "roi_outputs: F must be a multiple of 4");
    const int Rp2 = next_pow2i(a.R);
    const size_t smem = (size_t)Rp2 * 8 + (size_t)a.R * 16 + (size_t)a.R * 4 + (size_t)a.D * 4;
    hipLaunchKernelGGL(roi_final_kernel, dim3(N), dim3(256), smem, s, a, Rp2);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

}  // namespace vk
