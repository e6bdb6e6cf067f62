/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into or called from the
 * product path (vltk_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this.
 *
 * CPU restatement of the three third-party (torchvision) operators the
 * reference hot path calls:
 *   RoIPool      vltk/modeling/frcnn.py:30, :1179, :1198
 *   nms          vltk/modeling/frcnn.py:31, :132
 *   batched_nms  vltk/modeling/frcnn.py:31, :383   (single level => == nms)
 * torchvision is an UNPINNED dependency of the reference (requirements.txt:53)
 * and is not installed in this image, and the reference's own tests hold no
 * vectors for these ops: PARITY UNPINNED at this boundary.  The semantics
 * below restate torchvision's published CPU kernels (ops/cpu/roi_pool_kernel.cpp,
 * ops/cpu/nms_kernel.cpp) as summarised in SURVEY.md §8a rows 11/13/17.
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC; -ffp-contract=off so that
 * no FMA contraction changes the float arithmetic).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* RoIPool forward.  input NCHW float32, rois [K,5] = (batch_idx,x1,y1,x2,y2),
 * out [K,C,PH,PW].  argmax is not produced (inference only). */
void vko_roi_pool(const float *input, int N, int C, int H, int W,
                  const float *rois, int K, float spatial_scale,
                  int PH, int PW, float *out)
{
    (void)N;
    for (int k = 0; k < K; ++k) {
        const float *r = rois + 5 * k;
        int b = (int)r[0];
        int rsw = (int)roundf(r[1] * spatial_scale);
        int rsh = (int)roundf(r[2] * spatial_scale);
        int rew = (int)roundf(r[3] * spatial_scale);
        int reh = (int)roundf(r[4] * spatial_scale);
        int roi_w = imax(rew - rsw + 1, 1);   /* malformed RoIs forced to 1x1 */
        int roi_h = imax(reh - rsh + 1, 1);
        float bin_h = (float)roi_h / (float)PH;
        float bin_w = (float)roi_w / (float)PW;
        for (int ph = 0; ph < PH; ++ph) {
            int hs = (int)floorf((float)ph * bin_h);
            int he = (int)ceilf((float)(ph + 1) * bin_h);
            hs = imin(imax(hs + rsh, 0), H);
            he = imin(imax(he + rsh, 0), H);
            for (int pw = 0; pw < PW; ++pw) {
                int ws = (int)floorf((float)pw * bin_w);
                int we = (int)ceilf((float)(pw + 1) * bin_w);
                ws = imin(imax(ws + rsw, 0), W);
                we = imin(imax(we + rsw, 0), W);
                int empty = (he <= hs) || (we <= ws);
                for (int c = 0; c < C; ++c) {
                    const float *p = input + ((size_t)b * C + c) * H * W;
                    float m = empty ? 0.f : -FLT_MAX;
                    for (int h = hs; h < he; ++h)
                        for (int w = ws; w < we; ++w)
                            if (p[h * W + w] > m) m = p[h * W + w];
                    out[(((size_t)k * C + c) * PH + ph) * PW + pw] = m;
                }
            }
        }
    }
}

/* RoIAlign forward (torchvision.ops.roi_align CPU kernel semantics, `aligned` as in detectron2's ROIAlignV2).
 * torchvision is NOT a dependency the reference's FRCNN calls for this (it uses RoIPool): north_star names RoIAlign
 * (SURVEY.md 8f row N4), no reference vector exists -> parity unpinned; restated from the published kernel:
 *   offset = aligned ? 0.5 : 0; start = coord*scale - offset; size = end - start (min 1 when !aligned);
 *   grid = sampling_ratio > 0 ? sampling_ratio : ceil(size / pooled); sample (iy + .5) * bin / grid; bilinear with
 *   samples beyond [-1, H] x [-1, W] contributing 0 and coordinates clamped to [0, H-1]; mean over the grid.
 * input NCHW float32, rois [K,5], out [K,C,PH,PW]. */
static float vko_bilinear(const float *p, int H, int W, float y, float x)
{
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    float ly = y - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
    float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return w1 * p[yl * W + xl] + w2 * p[yl * W + xh] + w3 * p[yh * W + xl] + w4 * p[yh * W + xh];
}

void vko_roi_align(const float *input, int N, int C, int H, int W, const float *rois, int K, float spatial_scale,
                   int PH, int PW, int sampling_ratio, int aligned, float *out)
{
    (void)N;
    for (int k = 0; k < K; ++k) {
        const float *r = rois + 5 * k;
        int b = (int)r[0];
        float off = aligned ? 0.5f : 0.f;
        float sw = r[1] * spatial_scale - off, sh = r[2] * spatial_scale - off;
        float ew = r[3] * spatial_scale - off, eh = r[4] * spatial_scale - off;
        float rw = ew - sw, rh = eh - sh;
        if (!aligned) { rw = rw > 1.f ? rw : 1.f; rh = rh > 1.f ? rh : 1.f; }
        float bh = rh / (float)PH, bw = rw / (float)PW;
        int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
        int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
        float count = (float)(gh * gw > 1 ? gh * gw : 1);
        for (int c = 0; c < C; ++c) {
            const float *p = input + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    float acc = 0.f;
                    for (int iy = 0; iy < gh; ++iy) {
                        float y = sh + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                            acc += vko_bilinear(p, H, W, y, x);
                        }
                    }
                    out[(((size_t)k * C + c) * PH + ph) * PW + pw] = acc / count;
                }
        }
    }
}

/* stable descending argsort (ties -> lower index first): the build's defined
 * tie order. */
typedef struct { float s; int64_t i; } vko_si;
static int cmp_desc(const void *a, const void *b)
{
    const vko_si *x = (const vko_si *)a, *y = (const vko_si *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

void vko_argsort_desc(const float *scores, int64_t n, int64_t *order)
{
    vko_si *t = (vko_si *)malloc(sizeof(vko_si) * (size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; ++i) { t[i].s = scores[i]; t[i].i = i; }
    qsort(t, (size_t)n, sizeof(vko_si), cmp_desc);
    for (int64_t i = 0; i < n; ++i) order[i] = t[i].i;
    free(t);
}

/* Greedy NMS.  boxes [n,4] (x1,y1,x2,y2) f32, scores [n]; keep (capacity n)
 * receives indices in score order; returns their number.  area has no +1;
 * suppress j when (double)iou > thr, iou = inter / (a_i + a_j - inter) in f32
 * (torchvision's iou_threshold is a C++ double compared with a float). */
int64_t vko_nms(const float *boxes, const float *scores, int64_t n, double thr,
                int64_t *keep)
{
    if (n <= 0) return 0;
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    unsigned char *sup = (unsigned char *)calloc((size_t)n, 1);
    float *area = (float *)malloc(sizeof(float) * (size_t)n);
    vko_argsort_desc(scores, n, order);
    for (int64_t i = 0; i < n; ++i)
        area[i] = (boxes[4 * i + 2] - boxes[4 * i + 0]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    int64_t nk = 0;
    for (int64_t a = 0; a < n; ++a) {
        int64_t i = order[a];
        if (sup[i]) continue;
        keep[nk++] = i;
        float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        float ia = area[i];
        for (int64_t b = a + 1; b < n; ++b) {
            int64_t j = order[b];
            if (sup[j]) continue;
            float xx1 = fmaxf(ix1, boxes[4 * j]), yy1 = fmaxf(iy1, boxes[4 * j + 1]);
            float xx2 = fminf(ix2, boxes[4 * j + 2]), yy2 = fminf(iy2, boxes[4 * j + 3]);
            float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
            float inter = w * h;
            float ovr = inter / (ia + area[j] - inter);
            if ((double)ovr > thr) sup[j] = 1;
        }
    }
    free(order); free(sup); free(area);
    return nk;
}
