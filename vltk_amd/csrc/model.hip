// Model handle, weight loading / BN folding / repacking, workspace arena and the forward
// pass orchestration of libvltk_hip.so.
//
// Replaces (reference vltk/modeling/frcnn.py): FRCNN.__init__ :1744-1755, build_backbone
// :200-261, Res5ROIHeads.__init__ :1312-1363, the local branch of from_pretrained's
// load_state_dict :1862-1881, and FRCNN.inference :1942-2004 (call order of backbone ->
// proposal generator -> roi heads -> roi outputs).
//
// Data layout in HBM: every activation is NHWC in the handle's precision; one arena
// (single hipMalloc, re-grown only when a larger problem arrives) holds all intermediates
// so the steady state allocates nothing.  The Res5 head runs over RoI *chunks* so that the
// chunk's intermediates (pooled 14x14x1024 -> ... -> 14x14x2048) stay resident in the
// 256 MiB Infinity Cache between consecutive convolutions instead of round-tripping HBM.
#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "vk_common.h"

namespace vk {

thread_local KernelTimer *g_timer = nullptr;

hipEvent_t KernelTimer::get() {
    if (pool.empty()) {
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        all.push_back(e);
        return e;
    }
    hipEvent_t e = pool.back();
    pool.pop_back();
    return e;
}
void KernelTimer::collect() {
    const char *logp = getenv("VK_CONV_LOG");   // debug: one line per conv launch (shape, ms, TFLOP/s)
    FILE *lf = logp ? fopen(logp, "a") : nullptr;
    std::vector<Rec> pending;       // launches of a forward that is still in flight (vk_forward_begin without its _end yet)
    for (auto &r : recs) {
        float t = 0.f;
        if (hipEventQuery(r.e1) != hipSuccess) {
            pending.push_back(r);
            continue;
        }
        if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
            if (lf) fprintf(lf, "%d %d %d %d %d %d %.5f %.1f\n", r.bucket, r.M, r.cout, r.cin, r.k, r.stride, t, r.flops / (t * 1e-3) / 1e12);
            launches[r.bucket] += 1;
            ms[r.bucket] += t;
            flops[r.bucket] += r.flops;
            bytes[r.bucket] += r.bytes;
        }
        pool.push_back(r.e0);
        pool.push_back(r.e1);
    }
    if (lf) fclose(lf);
    recs.swap(pending);
    (void)hipGetLastError();        // hipEventQuery's "not ready" must not show up in a later launch check
}
KernelTimer::~KernelTimer() {
    for (auto e : all) (void)hipEventDestroy(e);
}

static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    bool loaded = false;
};

struct ConvLayer {
    std::string prefix;
    int cin = 0, cout = 0, k = 1, stride = 1, pad = 0, dil = 1;
    bool bn = true, relu = false;
    void *w = nullptr;       // device, packed
    float *b = nullptr;      // device, [packed_cout]
    int groups = 1;          // conv2 of a ResNeXt bottleneck (frcnn.py:950)
    int dt = -1;             // storage / arithmetic type of this layer; -1: the model's (vk_handle::dt)
};

struct Block {
    ConvLayer conv1, conv2, conv3, shortcut;
    bool has_shortcut = false;
    // fp16 fast mode: conv3 and a stride-1 projection shortcut run as ONE GEMM over K = [conv3 input | block
    // input] (`out += shortcut`, frcnn.py:970-977, without storing the shortcut): conv3.w / conv3.b then hold
    // the concatenated rows and the summed bias, and shortcut.w stays null
    bool fused_shortcut = false;
};

static const int kBlocks[3][4] = {{3, 4, 6, 3}, {3, 4, 23, 3}, {3, 8, 36, 3}};   // frcnn.py:226

}  // namespace vk

using namespace vk;

struct vk_handle {
    vk_config cfg;
    int device = 0;
    vk_dtype dt = VK_F16;
    // the box predictor (three small GEMMs + the box-delta rows, frcnn.py:1726-1740) runs in fp32 in BOTH modes: its
    // inputs are the fp32 RoI features, and fp16 weights / inputs alone would put the class and attribute logits at
    // 1.1e-3 / 2.0e-3 of the fp32 reference (measured) where north_star asks 1e-3; costs ~1 ms per 9600 RoIs
    vk_dtype pdt = VK_F32;
    bool finalized = false;
    std::vector<std::string> names;                 // expected state-dict tensors (strict load)
    std::map<std::string, HostTensor> host;

    ConvLayer stem;
    std::vector<Block> stages[3];                   // res2, res3, res4
    std::vector<Block> res5;
    ConvLayer rpn_conv, rpn_heads;                  // rpn_heads = [objectness | anchor_deltas] fused 1x1
    ConvLayer cls_score, fc_attr, attr_score;       // plain GEMMs (1x1 "convs" over K RoIs)
    void *bbox_w = nullptr;                         // [4C][F] in dt, unpadded rows (gathered per RoI)
    float *bbox_b = nullptr;
    void *emb = nullptr;                            // [C+1][F/8] in dt
    float *cell_anchors = nullptr;                  // [A][4]
    int A = 0, res4_c = 0, res5_c = 0, hid = 0, emb_dim = 0;
    std::vector<void *> owned;                      // device allocations to free

    // arena
    char *arena = nullptr;
    size_t arena_bytes = 0;
    int head_chunk = 9600;                           // RoIs per Res5 chunk (vk_set_option "head_chunk")
    int backbone_streams = 2;                        // 2: res3/res4 as two half-batches on two streams (option "backbone_streams")
    int backbone_split_min_batch = 8;                // ... from this batch size on (option "backbone_split_min_batch")
    int head_streams = 1;                            // 2: each Res5 chunk as two half-chunks on two streams (option "head_streams")
    int head_split_min_rois = 512;                   // ... for chunks of at least this many RoIs (option "head_split_min_rois")

    // stage bookkeeping of the last forward
    struct Stage {
        const void *ptr;
        vk_dtype dt;
        int64_t shape[4];
        int ndim;
    };
    std::map<std::string, Stage> stages_out;
    KernelTimer *ktimer = nullptr;
    bool timing = false;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipStream_t side = nullptr;                     // second stream of the res4 stage (half-batch pipelining)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t more_sides[2] = {nullptr, nullptr};  // third / fourth stream when backbone_streams is 3 / 4
    hipEvent_t more_joins[2] = {nullptr, nullptr};
    bool ev_valid = false;
    // forwards in flight (vk_forward_begin .. vk_forward_end): ticket t uses slot t % VK_MAX_INFLIGHT
    static constexpr int VK_MAX_INFLIGHT = 4;
    int32_t *flag_host = nullptr;                   // pinned [VK_MAX_INFLIGHT]: the non-finite flag of each forward
    char *meta_host = nullptr;                      // pinned [VK_MAX_INFLIGHT][meta_cap]: image_hw + scales_yx of each forward
    size_t meta_cap = 0;
    hipEvent_t ev_done[VK_MAX_INFLIGHT] = {nullptr, nullptr, nullptr, nullptr};
    int64_t next_ticket = 0, oldest_open = 0;       // tickets [oldest_open, next_ticket) have not been ended
};

namespace vk {

static void add_conv_names(std::vector<std::string> &n, const std::string &p, bool bn) {
    n.push_back(p + ".weight");
    if (bn) {
        n.push_back(p + ".norm.weight");
        n.push_back(p + ".norm.bias");
        n.push_back(p + ".norm.running_mean");
        n.push_back(p + ".norm.running_var");
    } else {
        n.push_back(p + ".bias");
    }
}

static Block make_block(const std::string &p, int cin, int cmid, int cout, int stride, int dil, bool stride_in_1x1,
                        int groups, std::vector<std::string> &names) {
    Block b;
    const int s1 = stride_in_1x1 ? stride : 1, s3 = stride_in_1x1 ? 1 : stride;   // frcnn.py:932
    b.has_shortcut = cin != cout;
    if (b.has_shortcut) {
        b.shortcut = ConvLayer{p + ".shortcut", cin, cout, 1, stride, 0, 1, true, false};
        add_conv_names(names, b.shortcut.prefix, true);
    }
    b.conv1 = ConvLayer{p + ".conv1", cin, cmid, 1, s1, 0, 1, true, true};
    b.conv2 = ConvLayer{p + ".conv2", cmid, cmid, 3, s3, dil, dil, true, true};
    b.conv2.groups = groups;                                                        // frcnn.py:950
    b.conv3 = ConvLayer{p + ".conv3", cmid, cout, 1, 1, 0, 1, true, true};   // relu after the residual add
    add_conv_names(names, b.conv1.prefix, true);
    add_conv_names(names, b.conv2.prefix, true);
    add_conv_names(names, b.conv3.prefix, true);
    return b;
}

static int dev_alloc(vk_handle *h, size_t bytes, void **out) {
    void *p = nullptr;
    VK_CHECK_HIP(hipMalloc(&p, bytes ? bytes : 16));
    h->owned.push_back(p);
    *out = p;
    return VK_OK;
}

static int upload(vk_handle *h, const void *host, size_t bytes, void **out) {
    VK_TRY(dev_alloc(h, bytes, out));
    VK_CHECK_HIP(hipMemcpy(*out, host, bytes, hipMemcpyHostToDevice));
    return VK_OK;
}

static const HostTensor *get_t(vk_handle *h, const std::string &name, std::vector<int64_t> shape) {
    auto it = h->host.find(name);
    if (it == h->host.end() || !it->second.loaded) {
        set_error("missing weight '%s' (strict load)", name.c_str());
        return nullptr;
    }
    if (it->second.shape != shape) {
        std::string got, want;
        for (auto v : it->second.shape) got += std::to_string(v) + ",";
        for (auto v : shape) want += std::to_string(v) + ",";
        set_error("weight '%s' has shape [%s], expected [%s]", name.c_str(), got.c_str(), want.c_str());
        return nullptr;
    }
    return &it->second;
}

static int finalize_conv(vk_handle *h, ConvLayer &L) {
    const HostTensor *w = get_t(h, L.prefix + ".weight", {L.cout, L.cin / L.groups, L.k, L.k});
    if (!w) return VK_EWEIGHTS;
    std::vector<float> bn;
    const float *bnp = nullptr, *bias = nullptr;
    if (L.bn) {
        const char *parts[4] = {".norm.weight", ".norm.bias", ".norm.running_mean", ".norm.running_var"};
        bn.resize(4 * (size_t)L.cout);
        for (int i = 0; i < 4; ++i) {
            const HostTensor *t = get_t(h, L.prefix + parts[i], {L.cout});
            if (!t) return VK_EWEIGHTS;
            memcpy(bn.data() + (size_t)i * L.cout, t->data.data(), sizeof(float) * L.cout);
        }
        bnp = bn.data();
    } else {
        const HostTensor *t = get_t(h, L.prefix + ".bias", {L.cout});
        if (!t) return VK_EWEIGHTS;
        bias = t->data.data();
    }
    const vk_dtype ldt = L.dt >= 0 ? (vk_dtype)L.dt : h->dt;
    const size_t wb = vk_packed_weight_bytes(L.cout, L.cin, L.k, L.k, L.groups, ldt);
    std::vector<char> packed(wb);
    std::vector<float> pb(vk_packed_cout(L.cout));
    VK_TRY(vk_pack_conv_weight(w->data.data(), bnp, bias, L.cout, L.cin, L.k, L.k, L.groups, ldt, packed.data(), pb.data()));
    VK_TRY(upload(h, packed.data(), wb, &L.w));
    VK_TRY(upload(h, pb.data(), pb.size() * sizeof(float), (void **)&L.b));
    return VK_OK;
}

// BN-folded packed rows + bias of one conv on the host (finalize_conv without the upload)
static int pack_conv_host(vk_handle *h, const ConvLayer &L, std::vector<char> &packed, std::vector<float> &pb) {
    const HostTensor *w = get_t(h, L.prefix + ".weight", {L.cout, L.cin / L.groups, L.k, L.k});
    if (!w) return VK_EWEIGHTS;
    std::vector<float> bn(4 * (size_t)L.cout);
    const char *parts[4] = {".norm.weight", ".norm.bias", ".norm.running_mean", ".norm.running_var"};
    for (int i = 0; i < 4; ++i) {
        const HostTensor *t = get_t(h, L.prefix + parts[i], {L.cout});
        if (!t) return VK_EWEIGHTS;
        memcpy(bn.data() + (size_t)i * L.cout, t->data.data(), sizeof(float) * L.cout);
    }
    packed.resize(vk_packed_weight_bytes(L.cout, L.cin, L.k, L.k, L.groups, h->dt));
    pb.resize(vk_packed_cout(L.cout));
    return vk_pack_conv_weight(w->data.data(), bn.data(), nullptr, L.cout, L.cin, L.k, L.k, L.groups, h->dt, packed.data(), pb.data());
}

// The same rule is restated in oracle/frcnn_oracle.py (fp16 emulation): keep the two in step.
static bool can_fuse_shortcut(const vk_handle *h, const Block &b) {
    static const bool off = getenv("VK_NO_FUSED_SHORTCUT") != nullptr;      // A/B switch
    return !off && h->dt == VK_F16 && b.has_shortcut && b.shortcut.stride == 1 && b.conv3.cout % 256 == 0 &&
           b.conv3.cin % 32 == 0 && b.shortcut.cin % 32 == 0;
}

static int finalize_block(vk_handle *h, Block &b) {
    VK_TRY(finalize_conv(h, b.conv1));
    VK_TRY(finalize_conv(h, b.conv2));
    b.fused_shortcut = can_fuse_shortcut(h, b);
    if (!b.fused_shortcut) {
        if (b.has_shortcut) VK_TRY(finalize_conv(h, b.shortcut));
        return finalize_conv(h, b.conv3);
    }
    std::vector<char> w3, wsc;
    std::vector<float> b3, bsc;
    VK_TRY(pack_conv_host(h, b.conv3, w3, b3));
    VK_TRY(pack_conv_host(h, b.shortcut, wsc, bsc));
    const size_t r3 = (size_t)b.conv3.cin * 2, rsc = (size_t)b.shortcut.cin * 2, rows = b3.size();
    std::vector<char> cat(rows * (r3 + rsc));
    for (size_t r = 0; r < rows; ++r) {
        memcpy(cat.data() + r * (r3 + rsc), w3.data() + r * r3, r3);
        memcpy(cat.data() + r * (r3 + rsc) + r3, wsc.data() + r * rsc, rsc);
        b3[r] += bsc[r];
    }
    VK_TRY(upload(h, cat.data(), cat.size(), &b.conv3.w));
    VK_TRY(upload(h, b3.data(), b3.size() * sizeof(float), (void **)&b.conv3.b));
    return VK_OK;
}

static void to_dt(const float *src, size_t n, vk_dtype dt, void *dst) {
    if (dt == VK_F16) {
        _Float16 *d = (_Float16 *)dst;
        for (size_t i = 0; i < n; ++i) d[i] = (_Float16)src[i];
    } else if (dt == VK_BF16) {          // round to nearest even, like torch's float -> bfloat16
        uint16_t *d = (uint16_t *)dst;
        for (size_t i = 0; i < n; ++i) {
            uint32_t u;
            memcpy(&u, &src[i], 4);
            if ((u & 0x7fffffffu) > 0x7f800000u)
                d[i] = (uint16_t)((u >> 16) | 0x40);              // NaN stays NaN
            else
                d[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        }
    } else {
        memcpy(dst, src, n * sizeof(float));
    }
}

// ---- arena ----
// fp16 fast mode: `box_features.mean(dim=[2,3])` (frcnn.py:1401) is folded into the last Res5 conv3's epilogue when
// a 128-row tile cannot span more than two RoIs (14x14 maps; RES5HALVE's 7x7 maps take the separate kernel)
// `rows`: pixels of one Res5 chunk.  The kernels of the fused form address the conv3 input with 32-bit byte offsets
// (conv_duo_pool_ok): a chunk whose [rows x mid channels] f16 tensor reaches 4 GiB (ResNeXt-152 32x8d at 9600 RoIs: 7.7 GB)
// keeps the separate mean kernel instead of failing the forward.
static bool fused_mean_ok(const vk_handle *h, int P, size_t rows) {
    static const bool off = getenv("VK_NO_FUSED_MEAN") != nullptr;      // A/B switch
    const long mid5 = (long)h->cfg.num_groups * h->cfg.width_per_group * 8;
    return !off && h->dt == VK_F16 && h->cfg.res5_halve == 0 && P * P >= 128 && P * P <= 255 && h->res5_c % 256 == 0 &&
           (long)rows * mid5 * 2 < (1L << 32);
}

struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(char *b) : base(b) {}
    void *take(size_t bytes) {
        void *p = base ? base + off : nullptr;
        off += align_up(bytes ? bytes : 16, 256);
        return p;
    }
};

struct Plan {
    // geometry
    int N, H, W, Hp, Wp, H1, W1, Hs[3], Ws[3], Hf, Wf, R, K, P, chunk;
    // buffers
    void *img_pad, *stem_out, *bufA, *bufB, *bufSC, *bufT1, *bufT2;
    void *rpn_hid;
    float *rpn_out, *prop_boxes, *prop_logits, *rois;
    int32_t *prop_counts, *image_hw, *nonfinite;
    float *scales;
    void *rpn_ws;
    size_t rpn_ws_bytes;
    void *pooled, *h_t1, *h_t2, *h_a, *h_b, *h_sc;
    float *pool_part;        // per-tile column sums of the last Res5 conv3 (fused spatial mean), or nullptr
    float *feat;
    void *featT, *concat, *attr_hid;
    float *cls_logits, *attr_logits, *obj_prob, *attr_prob, *chosen;
    int32_t *obj_cls, *attr_cls, *max_class;
    int64_t *keep_ids;
    size_t total;
};

static void conv_out_hw(int H, int W, int k, int s, int p, int d, int *Ho, int *Wo) {
    *Ho = (H + 2 * p - (d * (k - 1) + 1)) / s + 1;
    *Wo = (W + 2 * p - (d * (k - 1) + 1)) / s + 1;
}

static Plan make_plan(vk_handle *h, char *base, int N, int H, int W, int D) {
    Plan p;
    memset(&p, 0, sizeof(p));
    const vk_config &c = h->cfg;
    const size_t es = dtype_size(h->dt);
    p.N = N;
    p.H = H;
    p.W = W;
    conv_out_hw(H, W, 7, 2, 3, 1, &p.H1, &p.W1);
    p.Hp = std::max(H + 6, 2 * p.H1 + 6);
    p.Wp = std::max(W + 6, 2 * p.W1 + 6);
    p.Wp = (p.Wp + 1) & ~1;
    int h2, w2;
    vk_stem_out_hw(H, W, c.caffe_maxpool, &h2, &w2);
    p.Hs[0] = h2;
    p.Ws[0] = w2;
    for (int s = 1; s < 3; ++s) conv_out_hw(p.Hs[s - 1], p.Ws[s - 1], 1, 2, 0, 1, &p.Hs[s], &p.Ws[s]);
    p.Hf = p.Hs[2];
    p.Wf = p.Ws[2];
    p.R = c.post_nms_topk;
    p.K = N * p.R;
    p.P = c.pooler_resolution;
    p.chunk = std::min(h->head_chunk > 0 ? h->head_chunk : p.K, p.K);

    Carver cv(base);
    p.img_pad = cv.take((size_t)N * p.Hp * p.Wp * 4 * es);
    p.stem_out = cv.take((size_t)N * p.H1 * p.W1 * c.stem_out_channels * es);
    size_t max_out = 0, max_mid = 0;
    int cout = c.res2_out_channels, cmid = c.num_groups * c.width_per_group;
    for (int s = 0; s < 3; ++s) {
        const size_t px_out = (size_t)N * p.Hs[s] * p.Ws[s];
        const size_t px_in = s == 0 ? px_out : (size_t)N * p.Hs[s - 1] * p.Ws[s - 1];
        max_out = std::max(max_out, px_out * cout * es);
        max_mid = std::max(max_mid, (c.stride_in_1x1 ? px_out : px_in) * cmid * es);
        cout *= 2;
        cmid *= 2;
    }
    p.bufA = cv.take(max_out);
    p.bufB = cv.take(max_out);
    p.bufSC = cv.take(max_out);
    p.bufT1 = cv.take(max_mid);
    p.bufT2 = cv.take(max_mid);
    const size_t Mf = (size_t)N * p.Hf * p.Wf;
    p.rpn_hid = cv.take(Mf * h->hid * es);
    p.rpn_out = (float *)cv.take(Mf * (size_t)((5 * h->A + 7) / 8 * 8) * sizeof(float));
    p.prop_boxes = (float *)cv.take((size_t)p.K * 4 * sizeof(float));
    p.prop_logits = (float *)cv.take((size_t)p.K * sizeof(float));
    p.rois = (float *)cv.take((size_t)p.K * 5 * sizeof(float));
    p.prop_counts = (int32_t *)cv.take((size_t)N * sizeof(int32_t));
    p.image_hw = (int32_t *)cv.take((size_t)N * 2 * sizeof(int32_t));
    p.scales = (float *)cv.take((size_t)N * 2 * sizeof(float));
    p.nonfinite = (int32_t *)cv.take(sizeof(int32_t));
    p.rpn_ws_bytes = vk_rpn_workspace_bytes(N, p.Hf * p.Wf * h->A, c.pre_nms_topk);
    p.rpn_ws = cv.take(p.rpn_ws_bytes);
    const size_t rows = (size_t)p.chunk * p.P * p.P;
    const int mid5 = c.num_groups * c.width_per_group * 8;
    p.pooled = cv.take(rows * h->res4_c * es);
    p.h_t1 = cv.take(rows * mid5 * es);
    p.h_t2 = cv.take(rows * mid5 * es);
    p.h_a = cv.take(rows * h->res5_c * es);
    p.h_b = cv.take(rows * h->res5_c * es);
    p.h_sc = cv.take(rows * h->res5_c * es);
    p.pool_part = nullptr;
    // + one tile: two half-chunks on two streams keep separate partials and each rounds its tile count up
    if (fused_mean_ok(h, p.P, rows)) p.pool_part = (float *)cv.take(conv_duo_pool_part_bytes((long)rows + 128, h->res5_c));
    p.feat = (float *)cv.take((size_t)p.K * h->res5_c * sizeof(float));
    const size_t pes = dtype_size(h->pdt);
    p.featT = cv.take((size_t)p.K * h->res5_c * pes);
    p.concat = cv.take((size_t)p.K * (h->res5_c + h->emb_dim) * pes);
    p.attr_hid = cv.take((size_t)p.K * (h->res5_c / 4) * pes);
    p.cls_logits = (float *)cv.take((size_t)p.K * ((c.num_classes + 1 + 7) / 8 * 8) * sizeof(float));
    p.attr_logits = (float *)cv.take((size_t)p.K * ((c.num_attrs + 1 + 7) / 8 * 8) * sizeof(float));
    p.obj_prob = (float *)cv.take((size_t)p.K * sizeof(float));
    p.attr_prob = (float *)cv.take((size_t)p.K * sizeof(float));
    p.chosen = (float *)cv.take((size_t)p.K * 4 * sizeof(float));
    p.obj_cls = (int32_t *)cv.take((size_t)p.K * sizeof(int32_t));
    p.attr_cls = (int32_t *)cv.take((size_t)p.K * sizeof(int32_t));
    p.max_class = (int32_t *)cv.take((size_t)p.K * sizeof(int32_t));
    p.keep_ids = (int64_t *)cv.take((size_t)N * D * sizeof(int64_t));
    p.total = cv.off;
    return p;
}

static int run_conv(vk_handle *h, const ConvLayer &L, const void *x, int N, int H, int W, const void *res, void *y,
                    bool relu, vk_dtype out_dt, int ldy, hipStream_t s, int *Ho = nullptr, int *Wo = nullptr,
                    const void *x2 = nullptr, int cin2 = 0, float *pool_part = nullptr, bool concurrent = false) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.concurrent = concurrent ? 1 : 0;
    a.x = x;
    a.x2 = x2;
    a.Cin2 = cin2;
    a.pool_part = pool_part;
    a.w = L.w;
    a.bias = L.b;
    a.res = res;
    a.y = y;
    a.N = N;
    a.H = H;
    a.W = W;
    a.Cin = L.cin;
    conv_out_hw(H, W, L.k, L.stride, L.pad, L.dil, &a.Ho, &a.Wo);
    a.Cout = L.cout;
    a.ldy = ldy > 0 ? ldy : (L.cout + 7) / 8 * 8;
    a.kh = a.kw = L.k;
    a.stride = L.stride;
    a.pad = L.pad;
    a.dil = L.dil;
    a.groups = L.groups;
    a.relu = relu;
    a.stem = 0;
    a.dt = L.dt >= 0 ? (vk_dtype)L.dt : h->dt;
    a.out_dt = out_dt;
    if (Ho) *Ho = a.Ho;
    if (Wo) *Wo = a.Wo;
    return launch_conv(a, s);
}

// BottleneckBlock.forward frcnn.py:963-979.  x [N,H,W,cin] -> y [N,Ho,Wo,cout]
// n0 / nb: run images [n0, n0 + nb) of the full-batch buffers (every buffer is [N, H, W, C]: the half-batch pointers are
// plain offsets).  nb < 0: the whole batch.
static int run_block(vk_handle *h, const Block &b, const void *x, int N, int H, int W, void *t1, void *t2, void *sc,
                     void *y, hipStream_t s, int *Ho, int *Wo, float *pool_part = nullptr, int n0 = 0, int nb = -1) {
    int h1, w1, h2, w2;
    const bool cc = nb >= 0;            // half-batch beside the other half on a second stream
    if (nb >= 0) {
        const size_t es = dtype_size(h->dt);
        int ho, wo;
        conv_out_hw(H, W, b.conv1.k, b.conv1.stride, b.conv1.pad, b.conv1.dil, &h1, &w1);
        conv_out_hw(h1, w1, b.conv2.k, b.conv2.stride, b.conv2.pad, b.conv2.dil, &h2, &w2);
        conv_out_hw(h2, w2, 1, 1, 0, 1, &ho, &wo);
        x = (const char *)x + (size_t)n0 * H * W * b.conv1.cin * es;
        t1 = (char *)t1 + (size_t)n0 * h1 * w1 * b.conv1.cout * es;
        t2 = (char *)t2 + (size_t)n0 * h2 * w2 * b.conv2.cout * es;
        sc = (char *)sc + (size_t)n0 * ho * wo * b.conv3.cout * es;
        y = (char *)y + (size_t)n0 * ho * wo * b.conv3.cout * es;
        N = nb;
    }
    // res2: the whole block as one kernel (bneck_fused.hip) -- x is read once, t1 / t2 never leave the CU
    if (h->dt == VK_F16 && !pool_part && b.conv1.stride == 1 && b.conv2.stride == 1 && b.conv2.dil == 1 && (!b.has_shortcut || b.fused_shortcut) &&
        bneck_fused_eligible(b.conv1.cin, b.conv1.cout, b.conv3.cout, 1, b.conv2.groups, b.has_shortcut, N, H, W, h->dt)) {
        if (Ho) *Ho = H;
        if (Wo) *Wo = W;
        return launch_bneck_fused(x, N, H, W, b.conv1.cin, b.has_shortcut, b.conv1.w, b.conv1.b, b.conv2.w, b.conv2.b, b.conv3.w, b.conv3.b, y,
                                  cc, s);
    }
    const void *res = x;
    if (b.has_shortcut && !b.fused_shortcut) {
        VK_TRY(run_conv(h, b.shortcut, x, N, H, W, nullptr, sc, false, h->dt, 0, s, nullptr, nullptr, nullptr, 0, nullptr, cc));
        res = sc;
    }
    VK_TRY(run_conv(h, b.conv1, x, N, H, W, nullptr, t1, true, h->dt, 0, s, &h1, &w1, nullptr, 0, nullptr, cc));
    VK_TRY(run_conv(h, b.conv2, t1, N, h1, w1, nullptr, t2, true, h->dt, 0, s, &h2, &w2, nullptr, 0, nullptr, cc));
    if (b.fused_shortcut)      // stride-1 block: t2 and x cover the same pixels
        return run_conv(h, b.conv3, t2, N, h2, w2, nullptr, y, true, h->dt, 0, s, Ho, Wo, x, b.shortcut.cin, pool_part, cc);
    VK_TRY(run_conv(h, b.conv3, t2, N, h2, w2, res, y, true, h->dt, 0, s, Ho, Wo, nullptr, 0, pool_part, cc));
    return VK_OK;
}

static void set_stage(vk_handle *h, const char *name, const void *ptr, vk_dtype dt, std::initializer_list<int64_t> shape) {
    vk_handle::Stage st;
    st.ptr = ptr;
    st.dt = dt;
    st.ndim = 0;
    for (auto v : shape) st.shape[st.ndim++] = v;
    h->stages_out[name] = st;
}

}  // namespace vk

extern "C" {

const char *vk_last_error(void) { return g_err; }
int vk_version(void) { return 1; }

int vk_packed_cout(int cout) { return (cout + CONV_COUT_ALIGN - 1) / CONV_COUT_ALIGN * CONV_COUT_ALIGN; }

int vk_conv_slice_channels(int cin, int groups) {
    if (groups <= 1) return cin;
    if (cin <= 0 || cin % groups != 0) return -1;
    const int cpg = cin / groups;
    if (cpg & (cpg - 1)) return -1;                     // power of two: slices and groups nest
    return std::min(std::max(cpg, 64), cin);
}

size_t vk_packed_weight_bytes(int cout, int cin, int kh, int kw, int groups, vk_dtype dt) {
    const int sw = vk_conv_slice_channels(cin, groups);
    return sw <= 0 ? 0 : (size_t)vk_packed_cout(cout) * kh * kw * sw * dtype_size(dt);
}

int vk_pack_conv_weight(const float *w, const float *bn, const float *bias, int cout, int cin, int kh, int kw, int groups,
                        vk_dtype dt, void *w_packed, float *bias_packed) {
    VK_REQUIRE(dt == VK_F16 || dt == VK_F32 || dt == VK_BF16, VK_EINVAL, "pack: dtype must be f16, bf16 or f32");
    VK_REQUIRE(groups >= 1, VK_EINVAL, "pack: groups=%d", groups);
    const int sw = vk_conv_slice_channels(cin, groups);     // K channels per tap in the packed row (== cin when dense)
    VK_REQUIRE(sw > 0, VK_EINVAL, "pack: cin=%d / groups=%d must be a power of two", cin, groups);
    VK_REQUIRE(groups == 1 || cin == cout, VK_EINVAL, "pack: grouped convolution needs cin == cout (got %d, %d)", cin, cout);
    VK_REQUIRE((sw * (int)dtype_size(dt)) % CONV_KTILE_BYTES == 0, VK_EINVAL,
               "pack: %d channels per tap is not a whole number of 128-byte K-tiles for this dtype", sw);
    const int cp = vk_packed_cout(cout);
    const int cpg = cin / groups;                           // input channels of one group (= row length of w_oihw)
    const size_t K = (size_t)kh * kw * sw;
    std::vector<float> row(K);
    for (int co = 0; co < cp; ++co) {
        double s = 1.0;
        float b = 0.f;
        std::fill(row.begin(), row.end(), 0.f);
        if (co < cout) {
            if (bn) {   // eval BatchNorm folded into the conv: eps 1e-5 (nn.BatchNorm2d default)
                const double g = bn[co], be = bn[cout + co], mu = bn[2 * (size_t)cout + co], var = bn[3 * (size_t)cout + co];
                s = g / std::sqrt(var + 1e-5);
                b = (float)(be - mu * s);
            } else if (bias) {
                b = bias[co];
            }
            // the slice this channel's 64-wide output tile reads starts at slice0; its own group at g0
            const int slice0 = groups == 1 ? 0 : (co / 64 * 64) / sw * sw;
            const int g0 = groups == 1 ? 0 : co / cpg * cpg;
            for (int c = 0; c < cpg; ++c)
                for (int y = 0; y < kh; ++y)
                    for (int x = 0; x < kw; ++x)
                        row[((size_t)y * kw + x) * sw + (g0 + c - slice0)] =
                            (float)((double)w[(((size_t)co * cpg + c) * kh + y) * kw + x] * s);
        }
        bias_packed[co] = b;
        to_dt(row.data(), K, dt, (char *)w_packed + (size_t)co * K * dtype_size(dt));
    }
    return VK_OK;
}

size_t vk_packed_stem_bytes(int cout, vk_dtype dt) {
    const int ktiles = dt == VK_F16 ? 4 : 7;
    return (size_t)vk_packed_cout(cout) * ktiles * CONV_KTILE_BYTES;
}

int vk_pack_stem_weight(const float *w, const float *bn, int cout, vk_dtype dt, void *w_packed, float *bias_packed) {
    VK_REQUIRE(dt == VK_F16 || dt == VK_F32, VK_EINVAL, "pack_stem: dtype must be f16 or f32");
    const int cp = vk_packed_cout(cout);
    const int ktiles = dt == VK_F16 ? 4 : 7;
    const size_t K = (size_t)ktiles * CONV_KTILE_BYTES / dtype_size(dt);   // 256 (f16) / 224 (f32)
    std::vector<float> row(K);
    for (int co = 0; co < cp; ++co) {
        std::fill(row.begin(), row.end(), 0.f);
        float b = 0.f;
        if (co < cout) {
            double s = 1.0;
            if (bn) {
                const double g = bn[co], be = bn[cout + co], mu = bn[2 * (size_t)cout + co], var = bn[3 * (size_t)cout + co];
                s = g / std::sqrt(var + 1e-5);
                b = (float)(be - mu * s);
            }
            for (int c = 0; c < 3; ++c)
                for (int y = 0; y < 7; ++y)
                    for (int x = 0; x < 7; ++x)   // K index = kernel row * 32 + (kernel col * 4 + channel)
                        row[(size_t)y * 32 + x * 4 + c] = (float)((double)w[(((size_t)co * 3 + c) * 7 + y) * 7 + x] * s);
        }
        bias_packed[co] = b;
        to_dt(row.data(), K, dt, (char *)w_packed + (size_t)co * K * dtype_size(dt));
    }
    return VK_OK;
}

int vk_conv2d(const void *x, int N, int H, int W, int cin, const void *w_packed, const float *bias_packed,
              const void *residual, void *y, int cout, int ldy, int kh, int kw, int stride, int pad, int dil, int groups,
              int relu, vk_dtype dt, vk_dtype out_dt, void *stream) {
    VK_REQUIRE(kh == kw && kh >= 1 && stride >= 1 && dil >= 1 && pad >= 0 && groups >= 1, VK_EINVAL, "conv2d: bad geometry");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.w = w_packed;
    a.bias = bias_packed;
    a.res = residual;
    a.y = y;
    a.N = N;
    a.H = H;
    a.W = W;
    a.Cin = cin;
    conv_out_hw(H, W, kh, stride, pad, dil, &a.Ho, &a.Wo);
    VK_REQUIRE(a.Ho > 0 && a.Wo > 0 && N > 0, VK_EINVAL, "conv2d: empty output");
    a.Cout = cout;
    a.ldy = ldy;
    a.kh = kh;
    a.kw = kw;
    a.stride = stride;
    a.pad = pad;
    a.dil = dil;
    a.groups = groups;
    a.relu = relu;
    a.dt = dt;
    a.out_dt = out_dt;
    return launch_conv(a, (hipStream_t)stream);
}

int vk_conv1x1_dual(const void *x1, int cin1, const void *x2, int cin2, long M, const void *w_packed, const float *bias_packed,
                    const void *residual, void *y, int cout, int relu, void *stream) {
    VK_REQUIRE(x1 && x2 && M > 0 && M < (1L << 31), VK_EINVAL, "conv1x1_dual: bad arguments");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x1;
    a.x2 = x2;
    a.Cin = cin1;
    a.Cin2 = cin2;
    a.w = w_packed;
    a.bias = bias_packed;
    a.res = residual;
    a.y = y;
    a.N = 1;
    a.H = a.Ho = 1;
    a.W = a.Wo = (int)M;
    a.Cout = cout;
    a.ldy = cout;
    a.kh = a.kw = 1;
    a.stride = 1;
    a.dil = 1;
    a.groups = 1;
    a.relu = relu;
    a.dt = a.out_dt = VK_F16;
    return launch_conv(a, (hipStream_t)stream);
}

int vk_linear(const void *x, long M, int K, const void *w_packed, const float *bias_packed, const void *residual, void *y, int N, int ldy,
              int act, vk_dtype dt, vk_dtype out_dt, void *stream) {
    VK_REQUIRE(x && w_packed && bias_packed && y && M > 0 && M < (1L << 31), VK_EINVAL, "linear: bad arguments");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.Cin = K;
    a.w = w_packed;
    a.bias = bias_packed;
    a.res = residual;
    a.y = y;
    a.N = 1;
    a.H = a.Ho = 1;
    a.W = a.Wo = (int)M;
    a.Cout = N;
    a.ldy = ldy;
    a.kh = a.kw = 1;
    a.stride = 1;
    a.dil = 1;
    a.groups = 1;
    a.relu = act;
    a.dt = dt;
    a.out_dt = out_dt;
    return launch_conv(a, (hipStream_t)stream);
}

int vk_bottleneck64(const void *x, int N, int H, int W, int cin, int proj, const void *w1, const float *b1, const void *w2,
                    const float *b2, const void *w3, const float *b3, void *y, void *stream) {
    VK_REQUIRE(x && w1 && b1 && w2 && b2 && w3 && b3 && y && N > 0 && H > 0 && W > 0, VK_EINVAL, "bottleneck64: bad arguments");
    VK_REQUIRE((cin == 256 && !proj) || (cin == 64 && proj), VK_EINVAL, "bottleneck64: cin must be 256 (identity) or 64 (projection)");
    VK_REQUIRE(bneck_fused_eligible(cin, 64, 256, 1, 1, proj != 0, N, H, W, VK_F16) || getenv("VK_BNECK_FUSED"), VK_EINVAL,
               "bottleneck64: tensor beyond the 32-bit byte offsets");
    VK_REQUIRE((long)N * H * W * 512 < (1L << 32) - (1L << 20), VK_EINVAL, "bottleneck64: tensor beyond the 32-bit byte offsets");
    return launch_bneck_fused(x, N, H, W, cin, proj != 0, w1, b1, w2, b2, w3, b3, y, false, (hipStream_t)stream);
}

size_t vk_conv1x1_meanpool_workspace_bytes(int N, int HW, int cout) { return conv_duo_pool_part_bytes((long)N * HW, cout); }

int vk_conv1x1_meanpool(const void *x, int N, int HW, int cin, const void *w_packed, const float *bias_packed,
                        const void *residual, int cout, int relu, float *out_mean, void *workspace, size_t workspace_bytes,
                        void *stream) {
    VK_REQUIRE(x && out_mean && workspace && N > 0 && HW > 0, VK_EINVAL, "conv1x1_meanpool: bad arguments");
    VK_REQUIRE(workspace_bytes >= vk_conv1x1_meanpool_workspace_bytes(N, HW, cout), VK_EINVAL, "conv1x1_meanpool: workspace too small");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.Cin = cin;
    a.w = w_packed;
    a.bias = bias_packed;
    a.res = residual;
    a.pool_part = (float *)workspace;
    a.N = N;
    a.H = a.Ho = 1;
    a.W = a.Wo = HW;
    a.Cout = cout;
    a.ldy = cout;
    a.kh = a.kw = 1;
    a.stride = 1;
    a.dil = 1;
    a.groups = 1;
    a.relu = relu;
    a.dt = a.out_dt = VK_F16;
    VK_TRY(launch_conv(a, (hipStream_t)stream));
    return launch_pool_finish((const float *)workspace, N, HW, cin, cout, false, out_mean, (hipStream_t)stream);
}

static void stem_geom(int H, int W, int *H1, int *W1, int *Hp, int *Wp) {
    conv_out_hw(H, W, 7, 2, 3, 1, H1, W1);
    *Hp = std::max(H + 6, 2 * *H1 + 6);
    *Wp = (std::max(W + 6, 2 * *W1 + 6) + 1) & ~1;
}

size_t vk_stem_workspace_bytes(int N, int H, int W, int cout, vk_dtype dt) {
    int H1, W1, Hp, Wp;
    stem_geom(H, W, &H1, &W1, &Hp, &Wp);
    return align_up((size_t)N * Hp * Wp * 4 * dtype_size(dt), 256) + align_up((size_t)N * H1 * W1 * cout * dtype_size(dt), 256);
}

static int stem_impl(const float *x, int N, int H, int W, const void *w, const float *b, int cout, int caffe, void *y,
                     vk_dtype dt, void *img_pad, void *stem_out, hipStream_t s, int32_t *nonfinite = nullptr) {
    int H1, W1, Hp, Wp;
    stem_geom(H, W, &H1, &W1, &Hp, &Wp);
    VK_TRY(launch_stem_pack(x, img_pad, N, H, W, Hp, Wp, dt, s, nonfinite));
    if (stem_pool_eligible(cout, dt)) return launch_stem_pool(img_pad, N, Hp, Wp, H1, W1, w, b, caffe, y, s);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = img_pad;
    a.w = w;
    a.bias = b;
    a.y = stem_out;
    a.N = N;
    a.H = Hp;
    a.W = Wp;
    a.Cin = 4;
    a.Ho = H1;
    a.Wo = W1;
    a.Cout = cout;
    a.ldy = (cout + 7) / 8 * 8;
    a.kh = a.kw = 7;
    a.stride = 2;
    a.pad = 0;
    a.dil = 1;
    a.relu = 1;
    a.stem = 1;
    a.dt = a.out_dt = dt;
    VK_TRY(launch_conv(a, s));
    return launch_maxpool(stem_out, y, N, H1, W1, cout, caffe, dt, s);
}

int vk_stem(const float *x, int N, int H, int W, const void *w_packed, const float *bias_packed, int cout,
            int caffe_maxpool, void *y, vk_dtype dt, void *workspace, size_t workspace_bytes, void *stream) {
    VK_REQUIRE(dt == VK_F16 || dt == VK_F32, VK_EINVAL, "stem: bad dtype");
    VK_REQUIRE(cout % 8 == 0, VK_EINVAL, "stem: cout must be a multiple of 8");
    VK_REQUIRE(H >= 16 && W >= 16, VK_EINVAL, "stem: image %dx%d too small", H, W);
    VK_REQUIRE(workspace && workspace_bytes >= vk_stem_workspace_bytes(N, H, W, cout, dt), VK_EINVAL, "stem: workspace too small");
    int H1, W1, Hp, Wp;
    stem_geom(H, W, &H1, &W1, &Hp, &Wp);
    char *img_pad = (char *)workspace;
    char *stem_out = img_pad + align_up((size_t)N * Hp * Wp * 4 * dtype_size(dt), 256);
    return stem_impl(x, N, H, W, w_packed, bias_packed, cout, caffe_maxpool, y, dt, img_pad, stem_out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
int vk_create(const vk_config *cfg, int device, vk_handle **out) {
    VK_REQUIRE(cfg && out, VK_EINVAL, "create: null argument");
    VK_REQUIRE(cfg->depth == 50 || cfg->depth == 101 || cfg->depth == 152, VK_EINVAL, "create: depth %d unsupported", cfg->depth);
    VK_REQUIRE(cfg->num_groups >= 1 && cfg->width_per_group >= 1 &&
                   (cfg->num_groups == 1 || (cfg->width_per_group & (cfg->width_per_group - 1)) == 0),
               VK_EINVAL, "create: NUM_GROUPS=%d needs WIDTH_PER_GROUP (%d) to be a power of two", cfg->num_groups, cfg->width_per_group);
    // RES5HALVE=false only resets conv1/shortcut strides (frcnn.py:1351-1352): with the stride on conv2 the
    // reference's block 0 adds a 7x7 main path to a 14x14 shortcut and raises
    VK_REQUIRE(cfg->stride_in_1x1 != 0 || cfg->res5_halve != 0, VK_EINVAL,
               "create: STRIDE_IN_1X1=false with RES5HALVE=false leaves a stride-2 conv2 in res5 (frcnn.py:1351-1355): "
               "the reference's residual add fails on the shapes");
    VK_REQUIRE(cfg->precision == VK_F16 || cfg->precision == VK_F32, VK_EINVAL, "create: precision must be VK_F16 or VK_F32");
    VK_REQUIRE(cfg->num_sizes >= 1 && cfg->num_sizes <= VK_MAX_ANCHOR_DIM && cfg->num_ratios >= 1 &&
                   cfg->num_ratios <= VK_MAX_ANCHOR_DIM, VK_EINVAL, "create: bad anchor configuration");
    VK_REQUIRE(cfg->pre_nms_topk >= 1 && cfg->pre_nms_topk <= 8192, VK_EINVAL, "create: PRE_NMS_TOPK_TEST must be in 1..8192");
    VK_REQUIRE(cfg->post_nms_topk >= 1 && cfg->post_nms_topk <= 1024 && cfg->post_nms_topk <= cfg->pre_nms_topk, VK_EINVAL,
               "create: POST_NMS_TOPK_TEST must be in 1..min(1024, PRE_NMS_TOPK_TEST)");
    VK_REQUIRE(cfg->use_attr != 0, VK_EINVAL, "create: ROI_BOX_HEAD.ATTR=false is not supported");
    VK_REQUIRE(cfg->stem_out_channels == 64, VK_EINVAL, "create: STEM_OUT_CHANNELS must be 64");
    VK_CHECK_HIP(hipSetDevice(device));
    vk_handle *h = new vk_handle();
    h->cfg = *cfg;
    h->device = device;
    h->dt = (vk_dtype)cfg->precision;
    if (getenv("VK_PREDICTOR_FP16")) h->pdt = h->dt;            // A/B switch: round 1's 16-bit predictor
    const char *env = getenv("VK_HEAD_CHUNK");
    if (env && atoi(env) > 0) h->head_chunk = atoi(env);
    if (const char *bs = getenv("VK_BACKBONE_STREAMS"))
        if (bs[0] >= '1' && bs[0] <= '4') h->backbone_streams = bs[0] - '0';
    if (const char *hs = getenv("VK_HEAD_STREAMS"))
        if (hs[0] == '1' || hs[0] == '2') h->head_streams = hs[0] - '0';
    const int di = cfg->depth == 50 ? 0 : (cfg->depth == 101 ? 1 : 2);
    add_conv_names(h->names, "backbone.stem.conv1", true);
    h->stem = ConvLayer{"backbone.stem.conv1", 3, cfg->stem_out_channels, 7, 2, 3, 1, true, true};
    int cin = cfg->stem_out_channels, cout = cfg->res2_out_channels, cmid = cfg->num_groups * cfg->width_per_group;
    const char *sn[3] = {"res2", "res3", "res4"};
    for (int s = 0; s < 3; ++s) {
        for (int b = 0; b < kBlocks[di][s]; ++b) {
            const int stride = (b == 0 && s > 0) ? 2 : 1;   // frcnn.py:237
            h->stages[s].push_back(make_block(std::string("backbone.") + sn[s] + "." + std::to_string(b), cin, cmid, cout,
                                              stride, 1, cfg->stride_in_1x1 != 0, cfg->num_groups, h->names));
            cin = cout;
        }
        cout *= 2;
        cmid *= 2;
    }
    h->res4_c = cin;
    h->A = cfg->num_sizes * cfg->num_ratios;
    h->names.push_back("proposal_generator.anchor_generator.cell_anchors.0");
    h->hid = cfg->rpn_hidden_channels == -1 ? h->res4_c : cfg->rpn_hidden_channels;
    h->rpn_conv = ConvLayer{"proposal_generator.rpn_head.conv", h->res4_c, h->hid, 3, 1, 1, 1, false, true};
    add_conv_names(h->names, h->rpn_conv.prefix, false);
    add_conv_names(h->names, "proposal_generator.rpn_head.objectness_logits", false);
    add_conv_names(h->names, "proposal_generator.rpn_head.anchor_deltas", false);
    h->rpn_heads = ConvLayer{"proposal_generator.rpn_head.(objectness_logits|anchor_deltas)", h->hid, 5 * h->A, 1, 1, 0, 1, false, false};
    h->res5_c = cfg->res2_out_channels * 8;
    const int mid5 = cfg->num_groups * cfg->width_per_group * 8;
    cin = h->res4_c;
    for (int b = 0; b < 3; ++b) {
        // VG res5 (RES5HALVE=false): stride 1, conv2 dilation/padding 2 (frcnn.py:1345-1355);
        // RES5HALVE=true: the plain stage, first stride 2 (frcnn.py:1373-1383)
        const bool halve = cfg->res5_halve != 0;
        h->res5.push_back(make_block("roi_heads.res5." + std::to_string(b), cin, mid5, h->res5_c, (halve && b == 0) ? 2 : 1,
                                     halve ? 1 : 2, cfg->stride_in_1x1 != 0, cfg->num_groups, h->names));
        cin = h->res5_c;
    }
    const int C = cfg->num_classes, F = h->res5_c;
    h->emb_dim = F / 8;
    const std::string bp = "roi_heads.box_predictor.";
    h->cls_score = ConvLayer{bp + "cls_score", F, C + 1, 1, 1, 0, 1, false, false};
    h->fc_attr = ConvLayer{bp + "fc_attr", F + h->emb_dim, F / 4, 1, 1, 0, 1, false, true};
    h->attr_score = ConvLayer{bp + "attr_score", F / 4, cfg->num_attrs + 1, 1, 1, 0, 1, false, false};
    for (const char *n : {"cls_score", "bbox_pred"}) add_conv_names(h->names, bp + n, false);
    h->names.push_back(bp + "cls_embedding.weight");
    for (const char *n : {"fc_attr", "attr_score"}) add_conv_names(h->names, bp + n, false);
    *out = h;
    return VK_OK;
}

int vk_num_weights(vk_handle *h, int *count) {
    VK_REQUIRE(h && count, VK_EINVAL, "null argument");
    *count = (int)h->names.size();
    return VK_OK;
}

int vk_weight_name(vk_handle *h, int index, const char **name) {
    VK_REQUIRE(h && name && index >= 0 && index < (int)h->names.size(), VK_EINVAL, "weight index out of range");
    *name = h->names[index].c_str();
    return VK_OK;
}

int vk_load_weights(vk_handle *h, const char *name, const void *host_ptr, const int64_t *shape, int ndim, vk_dtype dtype) {
    VK_REQUIRE(h && name && host_ptr && (shape || ndim == 0), VK_EINVAL, "load_weights: null argument");
    VK_REQUIRE(!h->finalized, VK_EINVAL, "load_weights: model already finalized");
    std::string key(name);
    // old -> new naming, as the reference's loader does (frcnn.py:1862-1872)
    size_t pos;
    if ((pos = key.find("gamma")) != std::string::npos) key.replace(pos, 5, "weight");
    if ((pos = key.find("beta")) != std::string::npos) key.replace(pos, 4, "bias");
    if (key.size() > 19 && key.compare(key.size() - 19, 19, "num_batches_tracked") == 0) return VK_OK;   // unused in eval
    bool known = false;
    for (auto &n : h->names) known |= (n == key);
    VK_REQUIRE(known, VK_EWEIGHTS, "unexpected key '%s' in state_dict (strict load)", key.c_str());
    VK_REQUIRE(dtype == VK_F32, VK_EINVAL, "load_weights: '%s' must be float32", key.c_str());
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        t.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    t.data.assign((const float *)host_ptr, (const float *)host_ptr + n);
    t.loaded = true;
    h->host[key] = std::move(t);
    return VK_OK;
}

int vk_finalize(vk_handle *h) {
    VK_REQUIRE(h, VK_EINVAL, "finalize: null handle");
    VK_REQUIRE(!h->finalized, VK_EINVAL, "finalize: already finalized");
    VK_CHECK_HIP(hipSetDevice(h->device));
    for (auto &n : h->names) {
        auto it = h->host.find(n);
        VK_REQUIRE(it != h->host.end() && it->second.loaded, VK_EWEIGHTS, "missing key '%s' in state_dict (strict load)", n.c_str());
    }
    const vk_config &c = h->cfg;
    // stem
    {
        const HostTensor *w = get_t(h, "backbone.stem.conv1.weight", {c.stem_out_channels, 3, 7, 7});
        if (!w) return VK_EWEIGHTS;
        std::vector<float> bn(4 * (size_t)c.stem_out_channels);
        const char *parts[4] = {".norm.weight", ".norm.bias", ".norm.running_mean", ".norm.running_var"};
        for (int i = 0; i < 4; ++i) {
            const HostTensor *t = get_t(h, std::string("backbone.stem.conv1") + parts[i], {c.stem_out_channels});
            if (!t) return VK_EWEIGHTS;
            memcpy(bn.data() + (size_t)i * c.stem_out_channels, t->data.data(), sizeof(float) * c.stem_out_channels);
        }
        std::vector<char> packed(vk_packed_stem_bytes(c.stem_out_channels, h->dt));
        std::vector<float> pb(vk_packed_cout(c.stem_out_channels));
        VK_TRY(vk_pack_stem_weight(w->data.data(), bn.data(), c.stem_out_channels, h->dt, packed.data(), pb.data()));
        VK_TRY(upload(h, packed.data(), packed.size(), &h->stem.w));
        VK_TRY(upload(h, pb.data(), pb.size() * sizeof(float), (void **)&h->stem.b));
    }
    for (int s = 0; s < 3; ++s)
        for (auto &b : h->stages[s]) VK_TRY(finalize_block(h, b));
    for (auto &b : h->res5) VK_TRY(finalize_block(h, b));
    VK_TRY(finalize_conv(h, h->rpn_conv));
    {   // fuse the two 1x1 RPN heads into one GEMM: rows [0,A) objectness, [A,5A) anchor deltas
        const int A = h->A, hid = h->hid;
        const HostTensor *wo = get_t(h, "proposal_generator.rpn_head.objectness_logits.weight", {A, hid, 1, 1});
        const HostTensor *bo = get_t(h, "proposal_generator.rpn_head.objectness_logits.bias", {A});
        const HostTensor *wd = get_t(h, "proposal_generator.rpn_head.anchor_deltas.weight", {4 * A, hid, 1, 1});
        const HostTensor *bd = get_t(h, "proposal_generator.rpn_head.anchor_deltas.bias", {4 * A});
        if (!wo || !bo || !wd || !bd) return VK_EWEIGHTS;
        std::vector<float> w(wo->data), b(bo->data);
        w.insert(w.end(), wd->data.begin(), wd->data.end());
        b.insert(b.end(), bd->data.begin(), bd->data.end());
        std::vector<char> packed(vk_packed_weight_bytes(5 * A, hid, 1, 1, 1, h->dt));
        std::vector<float> pb(vk_packed_cout(5 * A));
        VK_TRY(vk_pack_conv_weight(w.data(), nullptr, b.data(), 5 * A, hid, 1, 1, 1, h->dt, packed.data(), pb.data()));
        VK_TRY(upload(h, packed.data(), packed.size(), &h->rpn_heads.w));
        VK_TRY(upload(h, pb.data(), pb.size() * sizeof(float), (void **)&h->rpn_heads.b));
        const HostTensor *ca = get_t(h, "proposal_generator.anchor_generator.cell_anchors.0", {A, 4});
        if (!ca) return VK_EWEIGHTS;
        VK_TRY(upload(h, ca->data.data(), ca->data.size() * sizeof(float), (void **)&h->cell_anchors));
    }
    // predictor: Linear weights [out,in] are 1x1 convs [out,in,1,1]
    const std::string bp = "roi_heads.box_predictor.";
    for (ConvLayer *L : {&h->cls_score, &h->fc_attr, &h->attr_score}) {
        auto it = h->host.find(L->prefix + ".weight");
        if (it != h->host.end() && it->second.shape.size() == 2) {
            it->second.shape.push_back(1);
            it->second.shape.push_back(1);
        }
        L->dt = h->pdt;
        VK_TRY(finalize_conv(h, *L));
    }
    {
        const int C = c.num_classes, F = h->res5_c;
        const int nb = c.cls_agnostic_bbox_reg ? 1 : C;
        const HostTensor *w = get_t(h, bp + "bbox_pred.weight", {4 * nb, F});
        const HostTensor *b = get_t(h, bp + "bbox_pred.bias", {4 * nb});
        const HostTensor *e = get_t(h, bp + "cls_embedding.weight", {C + 1, h->emb_dim});
        if (!w || !b || !e) return VK_EWEIGHTS;
        std::vector<char> tmp(w->data.size() * dtype_size(h->pdt));
        to_dt(w->data.data(), w->data.size(), h->pdt, tmp.data());
        VK_TRY(upload(h, tmp.data(), tmp.size(), &h->bbox_w));
        VK_TRY(upload(h, b->data.data(), b->data.size() * sizeof(float), (void **)&h->bbox_b));
        tmp.resize(e->data.size() * dtype_size(h->pdt));
        to_dt(e->data.data(), e->data.size(), h->pdt, tmp.data());
        VK_TRY(upload(h, tmp.data(), tmp.size(), &h->emb));
    }
    h->host.clear();
    h->finalized = true;
    return VK_OK;
}

int vk_set_option(vk_handle *h, const char *key, int value) {
    VK_REQUIRE(h && key, VK_EINVAL, "set_option: null argument");
    if (!strcmp(key, "head_chunk")) {
        VK_REQUIRE(value >= 0, VK_EINVAL, "head_chunk must be >= 0 (0 = all RoIs at once)");
        h->head_chunk = value;
        return VK_OK;
    }
    if (!strcmp(key, "backbone_streams")) {
        VK_REQUIRE(value >= 1 && value <= 4, VK_EINVAL, "backbone_streams must be 1..4");
        h->backbone_streams = value;
        return VK_OK;
    }
    if (!strcmp(key, "head_streams")) {
        VK_REQUIRE(value == 1 || value == 2, VK_EINVAL, "head_streams must be 1 or 2");
        h->head_streams = value;
        return VK_OK;
    }
    if (!strcmp(key, "head_split_min_rois")) {
        VK_REQUIRE(value >= 2, VK_EINVAL, "head_split_min_rois must be >= 2");
        h->head_split_min_rois = value;
        return VK_OK;
    }
    if (!strcmp(key, "backbone_split_min_batch")) {
        VK_REQUIRE(value >= 2, VK_EINVAL, "backbone_split_min_batch must be >= 2");
        h->backbone_split_min_batch = value;
        return VK_OK;
    }
    VK_REQUIRE(false, VK_EINVAL, "unknown option '%s'", key);
}

int vk_destroy(vk_handle *h) {
    if (!h) return VK_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (void *p : h->owned) (void)hipFree(p);
    if (h->arena) (void)hipFree(h->arena);
    for (auto &e : h->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : h->ev_done)
        if (e) (void)hipEventDestroy(e);
    if (h->flag_host) (void)hipHostFree(h->flag_host);
    if (h->meta_host) (void)hipHostFree(h->meta_host);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->side) (void)hipStreamDestroy(h->side);
    for (int i = 0; i < 2; ++i) {
        if (h->more_joins[i]) (void)hipEventDestroy(h->more_joins[i]);
        if (h->more_sides[i]) (void)hipStreamDestroy(h->more_sides[i]);
    }
    delete h->ktimer;
    delete h;
    return VK_OK;
}

int vk_enable_stage_timing(vk_handle *h, int enable) {
    VK_REQUIRE(h, VK_EINVAL, "null handle");
    h->timing = enable != 0;
    if (h->timing && !h->ev[0])
        for (auto &e : h->ev) VK_CHECK_HIP(hipEventCreate(&e));
    return VK_OK;
}

int vk_get_stage_timing(vk_handle *h, float *ms6) {
    VK_REQUIRE(h && ms6, VK_EINVAL, "null argument");
    VK_REQUIRE(h->timing && h->ev_valid, VK_EINVAL, "stage timing was not recorded");
    VK_CHECK_HIP(hipEventSynchronize(h->ev[5]));
    for (int i = 0; i < 5; ++i) VK_CHECK_HIP(hipEventElapsedTime(&ms6[i], h->ev[i], h->ev[i + 1]));
    VK_CHECK_HIP(hipEventElapsedTime(&ms6[5], h->ev[0], h->ev[5]));
    return VK_OK;
}

int vk_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes, void *stream) {
    VK_REQUIRE(dst_dev && src_dev, VK_EINVAL, "memcpy_d2d: null pointer");
    VK_CHECK_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return VK_OK;
}

int vk_get_stage(vk_handle *h, const char *name, const void **dev_ptr, vk_dtype *dtype, int64_t *shape, int *ndim) {
    VK_REQUIRE(h && name && dev_ptr && dtype && shape && ndim, VK_EINVAL, "null argument");
    auto it = h->stages_out.find(name);
    VK_REQUIRE(it != h->stages_out.end(), VK_EINVAL, "unknown stage '%s' (or no forward has run)", name);
    *dev_ptr = it->second.ptr;
    *dtype = it->second.dt;
    *ndim = it->second.ndim;
    for (int i = 0; i < it->second.ndim; ++i) shape[i] = it->second.shape[i];
    return VK_OK;
}

int vk_forward(vk_handle *h, const float *images_dev, int N, int H, int W, const int32_t *image_hw,
               const float *scales_yx, const vk_roi_params *rp, const vk_outputs *out, void *stream) {
    int64_t ticket = -1;
    VK_TRY(vk_forward_begin(h, images_dev, N, H, W, image_hw, scales_yx, rp, out, stream, &ticket));
    return vk_forward_end(h, ticket);
}

int vk_forward_begin(vk_handle *h, const float *images_dev, int N, int H, int W, const int32_t *image_hw,
                     const float *scales_yx, const vk_roi_params *rp, const vk_outputs *out, void *stream, int64_t *ticket) {
    VK_REQUIRE(h && images_dev && image_hw && rp && out && ticket, VK_EINVAL, "forward: null argument");
    VK_REQUIRE(h->next_ticket - h->oldest_open < vk_handle::VK_MAX_INFLIGHT, VK_EINVAL,
               "forward_begin: %d forwards are already in flight; end the oldest first", vk_handle::VK_MAX_INFLIGHT);
    VK_REQUIRE(h->finalized, VK_EINVAL, "forward: vk_finalize has not been called");
    VK_REQUIRE(N >= 1 && H >= 32 && W >= 32, VK_EINVAL, "forward: bad input size N=%d H=%d W=%d", N, H, W);
    VK_REQUIRE(rp->num_nms_thresh >= 1 && rp->num_nms_thresh <= VK_MAX_NMS_THRESH, VK_EINVAL, "forward: 1..%d nms thresholds", VK_MAX_NMS_THRESH);
    VK_REQUIRE(rp->max_detections >= 1 && rp->max_detections <= h->cfg.post_nms_topk, VK_EINVAL,
               "forward: max_detections=%d must be in 1..POST_NMS_TOPK_TEST", rp->max_detections);
    // image_shapes only bound the box clipping (frcnn.py:147-153); the reference does not check them
    // against the tensor size (its own adapter passes PIL (w,h) order, adapters/frcnn.py:50-52)
    for (int n = 0; n < N; ++n)
        VK_REQUIRE(image_hw[2 * n] >= 1 && image_hw[2 * n + 1] >= 1, VK_EINVAL, "forward: image_shapes[%d]=(%d,%d) must be positive",
                   n, image_hw[2 * n], image_hw[2 * n + 1]);
    VK_CHECK_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const vk_config &c = h->cfg;
    const int D = rp->max_detections;

    Plan need = make_plan(h, nullptr, N, H, W, D);
    if (need.total > h->arena_bytes) {
        if (h->arena) {
            VK_CHECK_HIP(hipDeviceSynchronize());
            VK_CHECK_HIP(hipFree(h->arena));
            h->arena = nullptr;
            h->arena_bytes = 0;
        }
        VK_CHECK_HIP(hipMalloc((void **)&h->arena, need.total));
        h->arena_bytes = need.total;
    }
    Plan p = make_plan(h, h->arena, N, H, W, D);
    h->stages_out.clear();
    struct TimerScope {   // per-launch events only inside this forward
        explicit TimerScope(KernelTimer *t) { g_timer = t; }
        ~TimerScope() { g_timer = nullptr; }
    } timer_scope(h->ktimer);
    const bool tm = h->timing;
    if (tm) VK_CHECK_HIP(hipEventRecord(h->ev[0], s));

    // the caller's host arrays are copied into the ticket's pinned slot: consumed before this call returns, and the
    // host-to-device copies are truly asynchronous
    const size_t meta_need = (sizeof(int32_t) + sizeof(float)) * 2 * (size_t)N;
    if (meta_need > h->meta_cap) {
        VK_CHECK_HIP(hipDeviceSynchronize());
        if (h->meta_host) VK_CHECK_HIP(hipHostFree(h->meta_host));
        h->meta_host = nullptr;
        h->meta_cap = align_up(meta_need, 4096);
        VK_CHECK_HIP(hipHostMalloc((void **)&h->meta_host, h->meta_cap * vk_handle::VK_MAX_INFLIGHT, hipHostMallocDefault));
    }
    char *meta = h->meta_host + (size_t)(h->next_ticket % vk_handle::VK_MAX_INFLIGHT) * h->meta_cap;
    memcpy(meta, image_hw, sizeof(int32_t) * 2 * N);
    VK_CHECK_HIP(hipMemcpyAsync(p.image_hw, meta, sizeof(int32_t) * 2 * N, hipMemcpyHostToDevice, s));
    if (scales_yx) {
        memcpy(meta + sizeof(int32_t) * 2 * N, scales_yx, sizeof(float) * 2 * N);
        VK_CHECK_HIP(hipMemcpyAsync(p.scales, meta + sizeof(int32_t) * 2 * N, sizeof(float) * 2 * N, hipMemcpyHostToDevice, s));
    }
    VK_CHECK_HIP(hipMemsetAsync(p.nonfinite, 0, sizeof(int32_t), s));

    // ---- backbone (ResNet.forward frcnn.py:1076-1090) ----
    VK_TRY(stem_impl(images_dev, N, H, W, h->stem.w, h->stem.b, c.stem_out_channels, c.caffe_maxpool, p.bufA, h->dt,
                     p.img_pad, p.stem_out, s, p.nonfinite));
    void *cur = p.bufA, *nxt = p.bufB;
    int ch = p.Hs[0], cw = p.Ws[0];
    // res4 at batch 32 is 2.05 rounds of tiles on 256 CUs: every N = 256 layer pays 3 rounds.  Its two half-batches run on
    // two streams, so the tail of one half's layer k overlaps the other half's layer k (images are independent; the
    // halves touch disjoint parts of every buffer).  Option "backbone_streams" = 1 / VK_BACKBONE_STREAMS=1 disables it.
    for (int st = 0; st < 3; ++st) {
        const bool split = h->backbone_streams >= 2 && st >= 1 && N >= h->backbone_streams && N >= h->backbone_split_min_batch && h->dt == VK_F16;
        const int ns = split ? h->backbone_streams : 1;      // image groups, one stream each
        hipStream_t gs_[4] = {s, nullptr, nullptr, nullptr};
        hipEvent_t gj_[4] = {nullptr, nullptr, nullptr, nullptr};
        if (split) {
            if (!h->side) {
                VK_CHECK_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
                VK_CHECK_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
                VK_CHECK_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
            }
            for (int i = 0; i < ns - 2; ++i)
                if (!h->more_sides[i]) {
                    VK_CHECK_HIP(hipStreamCreateWithFlags(&h->more_sides[i], hipStreamNonBlocking));
                    VK_CHECK_HIP(hipEventCreateWithFlags(&h->more_joins[i], hipEventDisableTiming));
                }
            gs_[1] = h->side;
            gj_[1] = h->ev_join;
            for (int i = 2; i < ns; ++i) {
                gs_[i] = h->more_sides[i - 2];
                gj_[i] = h->more_joins[i - 2];
            }
            VK_CHECK_HIP(hipEventRecord(h->ev_fork, s));
            for (int i = 1; i < ns; ++i) VK_CHECK_HIP(hipStreamWaitEvent(gs_[i], h->ev_fork, 0));
        }
        for (auto &b : h->stages[st]) {
            int ho, wo;
            if (split) {
                for (int i = 0; i < ns; ++i) {
                    const int n0 = (int)((long)N * i / ns), n1 = (int)((long)N * (i + 1) / ns);
                    VK_TRY(run_block(h, b, cur, N, ch, cw, p.bufT1, p.bufT2, p.bufSC, nxt, gs_[i], &ho, &wo, nullptr, n0, n1 - n0));
                }
            } else {
                VK_TRY(run_block(h, b, cur, N, ch, cw, p.bufT1, p.bufT2, p.bufSC, nxt, s, &ho, &wo));
            }
            std::swap(cur, nxt);
            ch = ho;
            cw = wo;
        }
        if (split) {
            for (int i = 1; i < ns; ++i) {
                VK_CHECK_HIP(hipEventRecord(gj_[i], gs_[i]));
                VK_CHECK_HIP(hipStreamWaitEvent(s, gj_[i], 0));
            }
        }
    }
    VK_REQUIRE(ch == p.Hf && cw == p.Wf, VK_EINVAL, "internal: res4 geometry mismatch (%dx%d vs %dx%d)", ch, cw, p.Hf, p.Wf);
    const void *res4 = cur;
    set_stage(h, "res4", res4, h->dt, {N, p.Hf, p.Wf, h->res4_c});
    if (tm) VK_CHECK_HIP(hipEventRecord(h->ev[1], s));

    // ---- RPN head (RPNHead.forward frcnn.py:1561-1572) ----
    const int ld_rpn = (5 * h->A + 7) / 8 * 8;
    VK_TRY(run_conv(h, h->rpn_conv, res4, N, p.Hf, p.Wf, nullptr, p.rpn_hid, true, h->dt, 0, s));
    VK_TRY(run_conv(h, h->rpn_heads, p.rpn_hid, N, p.Hf, p.Wf, nullptr, p.rpn_out, false, VK_F32, ld_rpn, s));
    set_stage(h, "rpn_out", p.rpn_out, VK_F32, {N, p.Hf, p.Wf, ld_rpn});
    if (tm) VK_CHECK_HIP(hipEventRecord(h->ev[2], s));

    // ---- proposals (RPN.inference frcnn.py:1615-1638) ----
    VK_TRY(vk_rpn_proposals(p.rpn_out, ld_rpn, p.rpn_out + h->A, ld_rpn, N, p.Hf, p.Wf, h->A, h->cell_anchors, 16,
                            c.anchor_offset, p.image_hw, c.rpn_bbox_weights, c.rpn_min_size, c.rpn_nms_thresh,
                            c.pre_nms_topk, c.post_nms_topk, p.prop_boxes, p.prop_logits, p.prop_counts, p.nonfinite,
                            p.rpn_ws, p.rpn_ws_bytes, s));
    VK_TRY(launch_make_rois(p.prop_boxes, N, p.R, p.rois, s));
    set_stage(h, "proposal_boxes", p.prop_boxes, VK_F32, {N, p.R, 4});
    set_stage(h, "proposal_logits", p.prop_logits, VK_F32, {N, p.R});
    set_stage(h, "proposal_counts", p.prop_counts, VK_I32, {N});
    if (tm) VK_CHECK_HIP(hipEventRecord(h->ev[3], s));

    // ---- RoI heads (Res5ROIHeads.forward frcnn.py:1391-1403), chunked over RoIs ----
    const int P = p.P;
    auto ensure_side = [&]() -> int {
        if (!h->side) {
            VK_CHECK_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
            VK_CHECK_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            VK_CHECK_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        }
        return VK_OK;
    };
    const size_t es5 = dtype_size(h->dt);
    for (int k0 = 0; k0 < p.K; k0 += p.chunk) {
        const int kc = std::min(p.chunk, p.K - k0);
        // option "head_streams" = 2: the chunk's two halves run on two streams (RoIs are independent; the halves use disjoint
        // rows of every head buffer).  Measured +0.9 % end to end at 9600 RoIs (tails of 58-round launches overlap); starting
        // the second half one or two layers late, so that a 3x3 MFMA loop runs beside a memory-bound 1x1 epilogue, is 1.5 %
        // SLOWER than one stream.  Off by default: it buys little and makes per-kernel durations overlap.
        const bool split = h->head_streams == 2 && kc >= 2 && kc >= h->head_split_min_rois && h->dt == VK_F16;
        const int ka = split ? kc / 2 : kc;
        if (split) {
            VK_TRY(ensure_side());
            VK_CHECK_HIP(hipEventRecord(h->ev_fork, s));
            VK_CHECK_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
        }
        const void *x = p.pooled;
        int hh = P, ww = P;
        for (int half = 0; half < (split ? 2 : 1); ++half) {
            const int n0 = half ? ka : 0, nb = half ? kc - ka : ka;
            hipStream_t hs = half ? h->side : s;
            VK_TRY(vk_roi_pool(res4, N, p.Hf, p.Wf, h->res4_c, p.rois + 5 * (size_t)(k0 + n0), nb, 1.0f / 16.0f, P,
                               (char *)p.pooled + (size_t)n0 * P * P * h->res4_c * es5, h->dt, hs));
            // the fused mean's per-tile partials: the second half gets its own region (tiles are counted per launch)
            float *pp = p.pool_part ? (float *)((char *)p.pool_part + (half ? conv_duo_pool_part_bytes((long)ka * P * P, h->res5_c) : 0)) : nullptr;
            void *a = p.h_a, *b2 = p.h_b;
            x = p.pooled;
            hh = P;
            ww = P;
            for (size_t bi = 0; bi < h->res5.size(); ++bi) {
                int ho, wo;
                const bool last = bi + 1 == h->res5.size();
                VK_TRY(run_block(h, h->res5[bi], x, kc, hh, ww, p.h_t1, p.h_t2, p.h_sc, a, hs, &ho, &wo, last ? pp : nullptr,
                                 split ? n0 : 0, split ? nb : -1));
                hh = ho;
                ww = wo;
                x = a;
                std::swap(a, b2);
            }
            if (p.pool_part)
                VK_TRY(launch_pool_finish(pp, nb, hh * ww, h->res5.back().conv3.cin, h->res5_c, h->res5.back().fused_shortcut,
                                          p.feat + (size_t)(k0 + n0) * h->res5_c, hs));
        }
        if (split) {
            VK_CHECK_HIP(hipEventRecord(h->ev_join, h->side));
            VK_CHECK_HIP(hipStreamWaitEvent(s, h->ev_join, 0));
        }
        if (!p.pool_part) VK_TRY(vk_mean_pool(x, kc, hh * ww, h->res5_c, p.feat + (size_t)k0 * h->res5_c, h->dt, s));
    }
    if (p.chunk >= p.K) set_stage(h, "pooled", p.pooled, h->dt, {p.K, P, P, h->res4_c});
    set_stage(h, "feature_pooled", p.feat, VK_F32, {p.K, h->res5_c});

    // ---- box predictor (FastRCNNOutputLayers.forward frcnn.py:1726-1740) ----
    const int C = c.num_classes, F = h->res5_c, E = h->emb_dim, AT = c.num_attrs;
    const int ld_cls = (C + 1 + 7) / 8 * 8, ld_attr = (AT + 1 + 7) / 8 * 8;
    VK_TRY(launch_concat_embed(p.feat, nullptr, nullptr, F, 0, p.K, p.featT, h->pdt, s));
    VK_TRY(run_conv(h, h->cls_score, p.featT, p.K, 1, 1, nullptr, p.cls_logits, false, VK_F32, ld_cls, s));
    VK_TRY(launch_softmax_argmax(p.cls_logits, ld_cls, p.K, C + 1, C, p.obj_prob, p.obj_cls, p.max_class, s));
    VK_TRY(launch_concat_embed(p.feat, h->emb, p.max_class, F, E, p.K, p.concat, h->pdt, s));
    VK_TRY(run_conv(h, h->fc_attr, p.concat, p.K, 1, 1, nullptr, p.attr_hid, true, h->pdt, 0, s));
    VK_TRY(run_conv(h, h->attr_score, p.attr_hid, p.K, 1, 1, nullptr, p.attr_logits, false, VK_F32, ld_attr, s));
    VK_TRY(launch_softmax_argmax(p.attr_logits, ld_attr, p.K, AT, AT, p.attr_prob, p.attr_cls, nullptr, s));
    VK_TRY(launch_chosen_deltas(p.featT, F, h->bbox_w, h->bbox_b, p.obj_cls, c.cls_agnostic_bbox_reg, F, p.K, p.chosen, h->pdt, s));
    set_stage(h, "obj_logits", p.cls_logits, VK_F32, {p.K, ld_cls});
    set_stage(h, "attr_logits", p.attr_logits, VK_F32, {p.K, ld_attr});
    set_stage(h, "chosen_deltas", p.chosen, VK_F32, {p.K, 4});
    if (tm) VK_CHECK_HIP(hipEventRecord(h->ev[4], s));

    // ---- outputs (ROIOutputs.inference frcnn.py:1262-1294) ----
    RoiFinalArgs a;
    memset(&a, 0, sizeof(a));
    a.obj_prob = p.obj_prob;
    a.obj_cls = p.obj_cls;
    a.attr_prob = p.attr_prob;
    a.attr_cls = p.attr_cls;
    a.box_deltas = p.chosen;
    a.ld_box = 4;
    a.delta_mode = 1;
    a.proposals = p.prop_boxes;
    a.counts = p.prop_counts;
    a.features = p.feat;
    a.F = F;
    a.R = p.R;
    a.D = D;
    a.image_hw = p.image_hw;
    a.scales_yx = scales_yx ? p.scales : nullptr;
    a.wx = c.roi_bbox_weights[0];
    a.wy = c.roi_bbox_weights[1];
    a.ww = c.roi_bbox_weights[2];
    a.wh = c.roi_bbox_weights[3];
    a.clampv = (float)std::log(1000.0 / 16.0);
    a.n_thresh = rp->num_nms_thresh;
    for (int i = 0; i < rp->num_nms_thresh; ++i) a.thresh[i] = rp->nms_thresh[i];
    a.mind = rp->min_detections;
    a.maxd = rp->max_detections;
    a.out = *out;
    a.keep_ids = p.keep_ids;
    a.nonfinite = p.nonfinite;
    VK_TRY(launch_roi_final(a, N, s));
    set_stage(h, "keep_ids", p.keep_ids, VK_I64, {N, D});
    if (tm) {
        VK_CHECK_HIP(hipEventRecord(h->ev[5], s));
        h->ev_valid = true;
    }

    // the reference asserts finite boxes on the host (frcnn.py:148): one 4-byte read-back into the ticket's pinned slot
    if (!h->flag_host) {
        VK_CHECK_HIP(hipHostMalloc((void **)&h->flag_host, sizeof(int32_t) * vk_handle::VK_MAX_INFLIGHT, hipHostMallocDefault));
        for (auto &e : h->ev_done) VK_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int slot = (int)(h->next_ticket % vk_handle::VK_MAX_INFLIGHT);
    VK_CHECK_HIP(hipMemcpyAsync(&h->flag_host[slot], p.nonfinite, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    VK_CHECK_HIP(hipEventRecord(h->ev_done[slot], s));
    *ticket = h->next_ticket++;
    return VK_OK;
}

int vk_forward_end(vk_handle *h, int64_t ticket) {
    VK_REQUIRE(h, VK_EINVAL, "forward_end: null handle");
    VK_REQUIRE(ticket == h->oldest_open && ticket < h->next_ticket, VK_EINVAL,
               "forward_end: ticket %lld is not the oldest forward in flight (%lld)", (long long)ticket, (long long)h->oldest_open);
    const int slot = (int)(ticket % vk_handle::VK_MAX_INFLIGHT);
    h->oldest_open++;
    VK_CHECK_HIP(hipSetDevice(h->device));
    VK_CHECK_HIP(hipEventSynchronize(h->ev_done[slot]));
    if (h->ktimer) h->ktimer->collect();
    VK_REQUIRE(h->flag_host[slot] == 0, VK_ENONFINITE, "Box tensor contains infinite or NaN!");
    return VK_OK;
}

int vk_enable_kernel_timing(vk_handle *h, int enable) {
    VK_REQUIRE(h, VK_EINVAL, "null handle");
    if (enable && !h->ktimer) h->ktimer = new KernelTimer();
    if (!enable && h->ktimer) {
        delete h->ktimer;
        h->ktimer = nullptr;
    }
    return VK_OK;
}

int vk_get_kernel_timing(vk_handle *h, int64_t *launches, double *ms, double *flops, double *bytes, int reset) {
    VK_REQUIRE(h && launches && ms && flops && bytes, VK_EINVAL, "null argument");
    VK_REQUIRE(h->ktimer, VK_EINVAL, "kernel timing is not enabled");
    for (int i = 0; i < VK_NUM_KERNEL_BUCKETS; ++i) {
        launches[i] = h->ktimer->launches[i];
        ms[i] = h->ktimer->ms[i];
        flops[i] = h->ktimer->flops[i];
        bytes[i] = h->ktimer->bytes[i];
        if (reset) {
            h->ktimer->bytes[i] = 0;
            h->ktimer->launches[i] = 0;
            h->ktimer->ms[i] = 0;
            h->ktimer->flops[i] = 0;
        }
    }
    return VK_OK;
}

// ---- pieces of FastRCNNOutputLayers.forward (frcnn.py:1726-1740) / ROIPooler input format (:426-441) for callers that
// compose a box head themselves (the FPN detector, vltk_amd/frcnn_fpn.py) ----
int vk_make_rois(const float *boxes, int N, int R, float *rois, void *stream) {
    VK_REQUIRE(boxes && rois && N > 0 && R > 0, VK_EINVAL, "make_rois: bad arguments");
    return launch_make_rois(boxes, N, R, rois, (hipStream_t)stream);
}

int vk_softmax_argmax(const float *logits, int ld, int K, int n_softmax, int n_argmax, float *prob_out, int32_t *cls_out,
                      int32_t *raw_argmax_out, void *stream) {
    VK_REQUIRE(logits && prob_out && cls_out && K >= 0 && n_softmax >= 1 && n_argmax >= 1 && n_argmax <= n_softmax && ld >= n_softmax,
               VK_EINVAL, "softmax_argmax: bad arguments");
    return launch_softmax_argmax(logits, ld, K, n_softmax, n_argmax, prob_out, cls_out, raw_argmax_out, (hipStream_t)stream);
}

int vk_concat_embed(const float *features, int F, const void *emb, int E, const int32_t *cls, int K, void *out, vk_dtype dt,
                    void *stream) {
    VK_REQUIRE(features && out && F > 0 && E >= 0 && K >= 0 && (E == 0 || (emb && cls)), VK_EINVAL, "concat_embed: bad arguments");
    VK_REQUIRE(dt == VK_F16 || dt == VK_F32, VK_EINVAL, "concat_embed: dtype must be f16 or f32");
    return launch_concat_embed(features, emb, cls, F, E, K, out, dt, (hipStream_t)stream);
}

int vk_chosen_deltas(const void *x, int ldx, const void *w_rows, const float *bias, const int32_t *cls, int cls_agnostic, int F,
                     int K, float *out, vk_dtype dt, void *stream) {
    VK_REQUIRE(x && w_rows && bias && out && (cls || cls_agnostic) && F > 0 && ldx >= F && K >= 0, VK_EINVAL, "chosen_deltas: bad arguments");
    VK_REQUIRE(dt == VK_F16 || dt == VK_F32, VK_EINVAL, "chosen_deltas: dtype must be f16 or f32");
    return launch_chosen_deltas(x, ldx, w_rows, bias, cls, cls_agnostic, F, K, out, dt, (hipStream_t)stream);
}

int vk_roi_outputs(const float *obj_logits, int ld_obj, const float *attr_logits, int ld_attr, const float *box_deltas,
                   int ld_box, int chosen_only, const float *proposals, const int32_t *counts, const float *features, int F,
                   int N, int R, int C, int A, const int32_t *image_hw, const float *scales_yx_dev,
                   const float *weights4_host, const vk_roi_params *rp, const vk_outputs *out, int64_t *keep_ids_out,
                   int32_t *nonfinite_flag, void *stream) {
    VK_REQUIRE(obj_logits && box_deltas && proposals && counts && features && image_hw && rp && out && nonfinite_flag, VK_EINVAL,
               "roi_outputs: null argument");
    VK_REQUIRE(rp->num_nms_thresh >= 1 && rp->num_nms_thresh <= VK_MAX_NMS_THRESH, VK_EINVAL, "roi_outputs: 1..%d nms thresholds", VK_MAX_NMS_THRESH);
    hipStream_t s = (hipStream_t)stream;
    const int K = N * R;
    // scratch for the per-RoI scores (freed after the stream drains; stage-level entry point only)
    char *scratch = nullptr;
    const size_t per = align_up((size_t)K * 4, 256);
    VK_CHECK_HIP(hipMalloc((void **)&scratch, per * 4));
    float *obj_prob = (float *)scratch, *attr_prob = (float *)(scratch + per);
    int32_t *obj_cls = (int32_t *)(scratch + 2 * per), *attr_cls = (int32_t *)(scratch + 3 * per);
    int st = launch_softmax_argmax(obj_logits, ld_obj, K, C + 1, C, obj_prob, obj_cls, nullptr, s);
    if (st == VK_OK && attr_logits) st = launch_softmax_argmax(attr_logits, ld_attr, K, A, A, attr_prob, attr_cls, nullptr, s);
    if (st == VK_OK) {
        RoiFinalArgs a;
        memset(&a, 0, sizeof(a));
        a.obj_prob = obj_prob;
        a.obj_cls = obj_cls;
        a.attr_prob = attr_logits ? attr_prob : nullptr;
        a.attr_cls = attr_logits ? attr_cls : nullptr;
        a.box_deltas = box_deltas;
        a.ld_box = ld_box;
        a.delta_mode = chosen_only ? 1 : 0;
        a.proposals = proposals;
        a.counts = counts;
        a.features = features;
        a.F = F;
        a.R = R;
        a.D = rp->max_detections;
        a.image_hw = image_hw;
        a.scales_yx = scales_yx_dev;
        a.wx = weights4_host[0];
        a.wy = weights4_host[1];
        a.ww = weights4_host[2];
        a.wh = weights4_host[3];
        a.clampv = (float)std::log(1000.0 / 16.0);
        a.n_thresh = rp->num_nms_thresh;
        for (int i = 0; i < rp->num_nms_thresh; ++i) a.thresh[i] = rp->nms_thresh[i];
        a.mind = rp->min_detections;
        a.maxd = rp->max_detections;
        a.out = *out;
        a.keep_ids = keep_ids_out;
        a.nonfinite = nonfinite_flag;
        st = launch_roi_final(a, N, s);
    }
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    if (st != VK_OK) return st;
    VK_CHECK_HIP(e);
    return VK_OK;
}

}  // extern "C"
