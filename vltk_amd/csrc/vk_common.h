// Internal helpers shared by the translation units of libvltk_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "vltk_hip.h"

namespace vk {

void set_error(const char *fmt, ...);

#define VK_CHECK_HIP(expr)                                                                    \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            vk::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return VK_EHIP;                                                                   \
        }                                                                                     \
    } while (0)

#define VK_REQUIRE(cond, code, ...)       \
    do {                                  \
        if (!(cond)) {                    \
            vk::set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)

#define VK_TRY(expr)              \
    do {                          \
        int _s = (expr);          \
        if (_s != VK_OK) return _s; \
    } while (0)

static inline size_t dtype_size(vk_dtype dt) {
    switch (dt) {
        case VK_F32: return 4;
        case VK_F16: return 2;
        case VK_I64: return 8;
        case VK_I32: return 4;
        case VK_BF16: return 2;
    }
    return 0;
}

constexpr int VK_MAX_DEVICES = 64;      // per-device statics of the launchers (zero pages, function attributes)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- convolution as implicit GEMM (conv_mfma.hip) --------------------------
constexpr int CONV_BM = 128;        // output pixels per workgroup tile
constexpr int CONV_KTILE_BYTES = 128;  // bytes of K per row per K-tile (64 f16 / 32 f32)
constexpr int CONV_COUT_ALIGN = 128;   // packed weight rows are padded to this

struct ConvArgs {
    const void *x;       // NHWC input (or the padded NHWC4 image in stem mode)
    const void *w;       // packed weights [cout_pad][ktiles * KTILE_BYTES]
    const float *bias;   // [cout_pad]
    const void *res;     // residual [M, ldy] or nullptr
    void *y;             // output [M, ldy]
    int N, H, W, Cin;    // input geometry (stem mode: padded Hp, Wp, 4)
    int Ho, Wo, Cout, ldy;
    int kh, kw, stride, pad, dil;
    int groups;          // 0 / 1: dense; > 1: slice-diagonal weights (vk_pack_conv_weight), Cin == Cout
    int concurrent;      // launched beside another stream's kernels (timing bucket 6)
    float *pool_part;    // fused spatial mean (Res5 `.mean(dim=[2,3])`): per-tile column sums go here, y is not written
    const void *x2;      // dual-source 1x1 (conv3 + projection shortcut in one GEMM): second input [M, Cin2], K = Cin | Cin2
    int Cin2;
    int relu;
    int stem;            // 1: K-tiles are runs of consecutive input pixels (7x7 s2 stem)
    vk_dtype dt, out_dt;
};
int launch_conv(const ConvArgs &a, hipStream_t stream);
bool conv256_eligible(const ConvArgs &a);                 // conv_mfma256.hip
int launch_conv256(const ConvArgs &a, hipStream_t stream);
bool conv3x3_panel_eligible(const ConvArgs &a);           // conv3x3_panel.hip (LDS-resident input panel, 9 taps per fetch)
int launch_conv3x3_panel(const ConvArgs &a, hipStream_t stream);
bool conv_gemm4_eligible(const ConvArgs &a);              // conv_gemm4.hip (1x1, K >= 1024, one or two inputs: 256x256 tile, four waves of 128x128)
int launch_conv_gemm4(const ConvArgs &a, hipStream_t stream);
int acquire_tile_counter(unsigned **ctr);     // a zeroed device word for one launch's dynamic tile tail; its last fetch zeroes it again (conv_gemm4.hip)
bool conv256_dual_ok(const ConvArgs &a);                  // conv_mfma256.hip (dual-source 1x1 with a long K: 256x256 tile)
bool conv_duo_eligible(const ConvArgs &a);                // conv_mfma_duo.hip (1x1 convs: 128x256 tile, two workgroups per CU)
int launch_conv_duo(const ConvArgs &a, hipStream_t stream);
bool conv_ws_eligible(const ConvArgs &a);                 // conv_ws.hip (1x1, K <= 512: weight-stationary, weights in registers)
int launch_conv_ws(const ConvArgs &a, hipStream_t stream);
bool conv3x3_blk_eligible(const ConvArgs &a);             // conv3x3_blk.hip (narrow channel blocks: ResNeXt grouped 3x3, dense 64 -> 64)
int launch_conv3x3_blk(const ConvArgs &a, hipStream_t stream);
// bneck_fused.hip: a whole res2 BottleneckBlock (64 bottleneck channels, stride 1) as one kernel
bool bneck_fused_eligible(int cin, int cmid, int cout, int stride, int groups, bool proj, long N, int H, int W, vk_dtype dt);
int launch_bneck_fused(const void *x, int N, int H, int W, int cin, bool proj, const void *w1, const float *b1, const void *w2,
                       const float *b2, const void *w3, const float *b3, void *y, bool concurrent, hipStream_t stream);
bool conv_duo_dual_ok(const ConvArgs &a);
bool conv_duo_pool_ok(const ConvArgs &a);                 // fused-mean form (pool_part set)
size_t conv_duo_pool_part_bytes(long M, int Cout);
bool conv_pool_sums_f64(long N, int HW, int Cin, int Cout, bool dual);   // which form the fused mean's workspace holds (conv_ws.hip)
bool conv_ws_pool_ok(const ConvArgs &a);
int launch_pool_finish(const float *part, int N, int HoWo, int Cin, int Cout, bool dual, float *out, hipStream_t stream);

// optional per-launch event timing (set by vk_forward when enabled; thread-local)
struct KernelTimer {
    struct Rec {
        int bucket;
        double flops;
        hipEvent_t e0, e1;
        int M, cout, cin, k, stride;
        double bytes;
    };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;   // free events
    std::vector<hipEvent_t> all;    // every event ever created (destroyed with the timer)
    int64_t launches[VK_NUM_KERNEL_BUCKETS] = {};
    double ms[VK_NUM_KERNEL_BUCKETS] = {};
    double flops[VK_NUM_KERNEL_BUCKETS] = {};
    double bytes[VK_NUM_KERNEL_BUCKETS] = {};   // algorithmic: input + output (+ residual) + weights, once each
    hipEvent_t get();
    void collect();   // accumulates every launch whose end event has completed; the rest stay pending
    ~KernelTimer();
};
extern thread_local KernelTimer *g_timer;

// ---- pool.hip ----
int launch_stem_pack(const float *x, void *y, int N, int H, int W, int Hp, int Wp, vk_dtype dt, hipStream_t s, int32_t *nonfinite = nullptr);
int launch_maxpool(const void *x, void *y, int N, int H, int W, int C, int caffe, vk_dtype dt, hipStream_t s);
bool stem_pool_eligible(int cout, vk_dtype dt);           // stem_pool.hip: 7x7 conv + BN + ReLU + max-pool as one kernel (f16, 64 channels)
int launch_stem_pool(const void *x, int N, int Hp, int Wp, int H1, int W1, const void *w, const float *bias, int caffe, void *y,
                     hipStream_t stream);

// ---- roi_out.hip ----
struct RoiFinalArgs {
    const float *obj_prob;
    const int32_t *obj_cls;
    const float *attr_prob;
    const int32_t *attr_cls;
    const float *box_deltas;
    int ld_box;
    int delta_mode;   // 0: full [K,4C] (index cls*4), 1: already the chosen/agnostic 4 deltas
    const float *proposals;
    const int32_t *counts;
    const float *features;
    int F, R, D;
    const int32_t *image_hw;
    const float *scales_yx;
    float wx, wy, ww, wh, clampv;
    int n_thresh;
    double thresh[VK_MAX_NMS_THRESH];
    int mind, maxd;
    vk_outputs out;
    int64_t *keep_ids;
    int32_t *nonfinite;
};
int launch_softmax_argmax(const float *logits, int ld, int K, int n_soft, int n_max, float *prob, int32_t *cls,
                          int32_t *raw_argmax, hipStream_t s);
int launch_concat_embed(const float *feat, const void *emb, const int32_t *cls, int F, int E, int K, void *out,
                        vk_dtype dt, hipStream_t s);
int launch_chosen_deltas(const void *x, int ldx, const void *w, const float *bias, const int32_t *cls, int agnostic, int F,
                         int K, float *out, vk_dtype dt, hipStream_t s);
int launch_roi_final(RoiFinalArgs &a, int N, hipStream_t s);
int launch_make_rois(const float *boxes, int N, int R, float *rois, hipStream_t s);

}  // namespace vk
