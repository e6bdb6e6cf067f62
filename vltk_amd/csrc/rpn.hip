// RPN proposal generation on gfx950: top-k selection, anchor decode, clip / size filter and
// greedy NMS, one image per workgroup, fixed-capacity buffers and device-side counts (no host
// synchronisation, unlike the reference's per-image Python loop with `.item()`).
//
// Replaces (reference vltk/modeling/frcnn.py):
//   RPNOutputs.predict_proposals / predict_objectness_logits      :748-781
//   AnchorGenerator.grid_anchors + _create_grid_offsets           :1463-1477, :176-197
//   Box2BoxTransform.apply_deltas                                 :548-584
//   find_top_rpn_proposals (sort, top-k, clip, filter, NMS, top)  :264-390
//   torchvision.ops.boxes.batched_nms / nms (third party; single level => plain nms)
//
// Arithmetic contract: all box math is fp32 with the reference's operation order and NO fused
// multiply-add (this file is compiled with -ffp-contract=off); the IoU test is
// (double)(inter / (a_i + a_j - inter)) > thr as in torchvision's CPU kernel; ranking is
// "score descending, ties -> lower flat index" (stable), the order this build defines where the
// reference's un-stable torch.sort leaves it open.
//
// Kernels
//   rpn_select_decode  1 WG (1024 thr) / image: 4-pass 8-bit radix select of the pre_topk-th
//                      best logit, index-ordered compaction (wave ballots), bitonic sort of
//                      <= 8192 (key,index) pairs in LDS, then decode+clip+filter of the sorted
//                      candidates.
//   nms_mask           64x64 IoU blocks -> 64-bit suppression masks (upper triangle only).
//   nms_scan           1 WG / image: greedy sweep over 64-box chunks; in-chunk resolution with
//                      v_readlane on the diagonal words, cross-chunk propagation by one lane per
//                      mask word; stops as soon as post_topk boxes are kept.
#include <cfloat>

#include "vk_common.h"

namespace vk {

constexpr int RPN_MAX_PRE = 8192;
constexpr int RPN_THREADS = 1024;

__device__ __forceinline__ uint32_t desc_key(float v) {
    uint32_t u = __float_as_uint(v);
    if (u == 0x80000000u) u = 0u;                       // -0.0 ranks equal to +0.0
    uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~asc;                                        // ascending key order == descending score
}

struct DecodeCfg {
    float wx, wy, ww, wh;
    float scale_clamp;
    float min_size;
    int stride;
    float offset;
    int stop;                // tools build only (VK_RPN_STOP): leave after the radix select (1), the compaction (2), the sort (3) -- timing only
};

// Box2BoxTransform.apply_deltas for one box (frcnn.py:559-583), fp32, reference op order.
__device__ __forceinline__ void apply_deltas1(const float a[4], const float d[4], float wx, float wy, float ww,
                                              float wh, float clampv, float o[4]) {
    float widths = a[2] - a[0];
    float heights = a[3] - a[1];
    float ctr_x = a[0] + 0.5f * widths;
    float ctr_y = a[1] + 0.5f * heights;
    float dx = d[0] / wx;
    float dy = d[1] / wy;
    float dw = d[2] / ww;
    float dh = d[3] / wh;
    dw = dw > clampv ? clampv : dw;     // torch.clamp(max=)
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

__device__ __forceinline__ bool finite4(const float b[4]) {
    return isfinite(b[0]) && isfinite(b[1]) && isfinite(b[2]) && isfinite(b[3]);
}

// _clip_box frcnn.py:147-153
__device__ __forceinline__ void clip4(float b[4], float h, float w) {
    b[0] = fminf(fmaxf(b[0], 0.f), w);
    b[1] = fminf(fmaxf(b[1], 0.f), h);
    b[2] = fminf(fmaxf(b[2], 0.f), w);
    b[3] = fminf(fmaxf(b[3], 0.f), h);
}

// ---------------------------------------------------------------------------
// workspace per image: cand_boxes [pre][4] f32, cand_logit [pre] f32, cand_valid [pre] u8 (as i32 words), cand_count [1]
__device__ __forceinline__ void rpn_select_decode_body(
    const float *__restrict__ logits, int ld_logits, const float *__restrict__ deltas, int ld_deltas, int Hf, int Wf,
    int A, const float *__restrict__ cell_anchors, const int32_t *__restrict__ image_hw, DecodeCfg cfg, int pre_topk,
    int sortn /* pow2 >= min(pre_topk, HWA) */, float *__restrict__ cand_boxes, float *__restrict__ cand_logit,
    int32_t *__restrict__ cand_valid, int32_t *__restrict__ cand_count, int32_t *__restrict__ nonfinite) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // all LDS in the one dynamic array (keeps its base 16-byte aligned): keys | hist | wave counts | scalars
    unsigned long long *skey = reinterpret_cast<unsigned long long *>(smem_raw);   // [sortn]
    uint32_t *hist = reinterpret_cast<uint32_t *>(skey + sortn);                   // [256]
    uint32_t *wave_lt = hist + 256, *wave_eq = wave_lt + 16;                       // [16] each
    uint32_t &sh_prefix = wave_eq[16], &sh_need = wave_eq[17], &sh_base_lt = wave_eq[18], &sh_base_eq = wave_eq[19];

    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int HW = Hf * Wf;
    const int HWA = HW * A;
    const int K = min(pre_topk, HWA);
    const float *lg = logits + (long)n * HW * ld_logits;
    const float *dl = deltas + (long)n * HW * ld_deltas;

    auto key_at = [&](int i) -> uint32_t {
        int pix = i / A, a = i - pix * A;
        return desc_key(lg[(long)pix * ld_logits + a]);
    };

    // ---- radix select: the K-th smallest key (== K-th best logit) ----
    uint32_t prefix = 0, need = (uint32_t)K;   // keys with (key >> shift_done) < prefix... tracked below
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint32_t himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        // eight loads in flight per thread (the sweep was one dependent load per iteration: latency-bound on one CU)
        int i = tid;
        for (; i + 7 * RPN_THREADS < HWA; i += 8 * RPN_THREADS) {
            uint32_t kk[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) kk[q] = key_at(i + q * RPN_THREADS);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if ((kk[q] & himask) == (prefix & himask)) atomicAdd(&hist[(kk[q] >> shift) & 255u], 1u);
        }
        for (; i < HWA; i += RPN_THREADS) {
            uint32_t k = key_at(i);
            if ((k & himask) == (prefix & himask)) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t cum = 0, b = 0;
            for (; b < 256; ++b) {
                if (cum + hist[b] >= need) break;
                cum += hist[b];
            }
            sh_prefix = prefix | (b << shift);
            sh_need = need - cum;
        }
        __syncthreads();
        prefix = sh_prefix;
        need = sh_need;
        __syncthreads();
    }
    const uint32_t T = prefix;   // threshold key; `need` of the keys equal to T are taken, lowest index first
#ifdef VK_ABLATION
    if (cfg.stop == 1) return;
#endif

    // ---- compaction in flat-index order ----
    for (int i = tid; i < sortn; i += RPN_THREADS) skey[i] = ~0ull;
    if (tid == 0) {
        sh_base_lt = 0;
        sh_base_eq = 0;
    }
    __syncthreads();
    for (int base = 0; base < HWA; base += RPN_THREADS) {
        const int i = base + tid;
        uint32_t k = 0xFFFFFFFFu;
        bool lt = false, eq = false;
        if (i < HWA) {
            k = key_at(i);
            lt = k < T;
            eq = k == T;
        }
        const unsigned long long blt = __ballot(lt), beq = __ballot(eq);
        if (lane == 0) {
            wave_lt[wave] = (uint32_t)__popcll(blt);
            wave_eq[wave] = (uint32_t)__popcll(beq);
        }
        __syncthreads();
        uint32_t lt_before = sh_base_lt, eq_before = sh_base_eq;
        for (int w = 0; w < wave; ++w) {
            lt_before += wave_lt[w];
            eq_before += wave_eq[w];
        }
        const unsigned long long lower = (1ull << lane) - 1ull;
        lt_before += (uint32_t)__popcll(blt & lower);
        eq_before += (uint32_t)__popcll(beq & lower);
        if (lt || (eq && eq_before < need)) {
            uint32_t pos = lt_before + min(eq_before, need);
            skey[pos] = ((unsigned long long)k << 32) | (uint32_t)i;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t a = 0, b = 0;
            for (int w = 0; w < 16; ++w) {
                a += wave_lt[w];
                b += wave_eq[w];
            }
            sh_base_lt += a;
            sh_base_eq += b;
        }
        __syncthreads();
    }

#ifdef VK_ABLATION
    if (cfg.stop == 2) return;
#endif
    // ---- bitonic sort (ascending composite key == score desc, index asc) ----
    for (int k2 = 2; k2 <= sortn; k2 <<= 1) {
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (int i = tid; i < sortn; i += RPN_THREADS) {
                int ixj = i ^ j2;
                if (ixj > i) {
                    unsigned long long a = skey[i], b = skey[ixj];
                    bool up = (i & k2) == 0;
                    if ((a > b) == up) {
                        skey[i] = b;
                        skey[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }

#ifdef VK_ABLATION
    if (cfg.stop == 3) return;
#endif
    // ---- decode + clip + size filter of the K sorted candidates ----
    const float img_h = (float)image_hw[2 * n], img_w = (float)image_hw[2 * n + 1];
    bool bad = false;
    for (int e = tid; e < K; e += RPN_THREADS) {
        const uint32_t idx = (uint32_t)(skey[e] & 0xFFFFFFFFull);
        const int pix = idx / A, a = idx - pix * A;
        const int y = pix / Wf, x = pix - y * Wf;
        const float sx = (float)((double)cfg.offset * cfg.stride + (double)x * cfg.stride);
        const float sy = (float)((double)cfg.offset * cfg.stride + (double)y * cfg.stride);
        const float *ca = cell_anchors + 4 * a;
        float anc[4] = {sx + ca[0], sy + ca[1], sx + ca[2], sy + ca[3]};
        const float *dp = dl + (long)pix * ld_deltas + 4 * a;
        float d[4] = {dp[0], dp[1], dp[2], dp[3]};
        float b[4];
        apply_deltas1(anc, d, cfg.wx, cfg.wy, cfg.ww, cfg.wh, cfg.scale_clamp, b);
        if (!finite4(b)) bad = true;
        clip4(b, img_h, img_w);
        const bool valid = ((b[2] - b[0]) > cfg.min_size) && ((b[3] - b[1]) > cfg.min_size);
        float *ob = cand_boxes + ((long)n * pre_topk + e) * 4;
        ob[0] = b[0];
        ob[1] = b[1];
        ob[2] = b[2];
        ob[3] = b[3];
        cand_logit[(long)n * pre_topk + e] = lg[(long)pix * ld_logits + a];
        cand_valid[(long)n * pre_topk + e] = valid ? 1 : 0;
    }
    if (bad) atomicOr(nonfinite, 1);
    if (tid == 0) cand_count[n] = K;
}

__global__ __launch_bounds__(RPN_THREADS) void rpn_select_decode_kernel(
    const float *__restrict__ logits, int ld_logits, const float *__restrict__ deltas, int ld_deltas, int Hf, int Wf,
    int A, const float *__restrict__ cell_anchors, const int32_t *__restrict__ image_hw, DecodeCfg cfg, int pre_topk,
    int sortn, float *__restrict__ cand_boxes, float *__restrict__ cand_logit, int32_t *__restrict__ cand_valid,
    int32_t *__restrict__ cand_count, int32_t *__restrict__ nonfinite) {
    rpn_select_decode_body(logits, ld_logits, deltas, ld_deltas, Hf, Wf, A, cell_anchors, image_hw, cfg, pre_topk, sortn, cand_boxes, cand_logit,
                           cand_valid, cand_count, nonfinite);
}

// Every level of a pyramid in ONE launch (grid = images x levels): a level's selection is one workgroup per image, so five launches of
// N workgroups one after the other left most of the chip idle (FPN detector at 32 images: 5 x 232 us; round 3).
constexpr int RPN_SEL_LEVELS = 6;
struct SelectLevels {
    const float *logits[RPN_SEL_LEVELS], *deltas[RPN_SEL_LEVELS], *cell_anchors[RPN_SEL_LEVELS];
    int ld_logits[RPN_SEL_LEVELS], ld_deltas[RPN_SEL_LEVELS], Hf[RPN_SEL_LEVELS], Wf[RPN_SEL_LEVELS], stride[RPN_SEL_LEVELS], sortn[RPN_SEL_LEVELS];
    float *cand_boxes[RPN_SEL_LEVELS], *cand_logit[RPN_SEL_LEVELS];
    int32_t *cand_valid[RPN_SEL_LEVELS], *cand_count[RPN_SEL_LEVELS];
};
__global__ __launch_bounds__(RPN_THREADS) void rpn_select_decode_levels_kernel(SelectLevels sl, int A, const int32_t *__restrict__ image_hw,
                                                                               DecodeCfg cfg, int pre_topk, int32_t *__restrict__ nonfinite) {
    const int l = blockIdx.y;
    cfg.stride = sl.stride[l];
    rpn_select_decode_body(sl.logits[l], sl.ld_logits[l], sl.deltas[l], sl.ld_deltas[l], sl.Hf[l], sl.Wf[l], A, sl.cell_anchors[l], image_hw, cfg,
                           pre_topk, sl.sortn[l], sl.cand_boxes[l], sl.cand_logit[l], sl.cand_valid[l], sl.cand_count[l], nonfinite);
}

// ---------------------------------------------------------------------------
// mask[n][i][cb] bit b set <=> j = cb*64+b > i and IoU(box_i, box_j) > thr.  grid (cb, rb, n), cb >= rb only.
// Two-phase use (launch_nms below): phase A computes only the leading `lead` x `lead` blocks -- the greedy sweep stops at the
// post-NMS cap, and the boxes behind the one that reaches it are never looked at; phase B (skip_lead = lead, done = the sweep's
// per-image flag) fills in the rest for the images whose sweep ran out of phase A's rows, and returns at once for the others.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float *__restrict__ boxes, const int32_t *__restrict__ counts,
                                                      int cap, int nwords, double thr,
                                                      unsigned long long *__restrict__ mask, int skip_lead,
                                                      const int32_t *__restrict__ done) {
    const int cb = blockIdx.x, rb = blockIdx.y, n = blockIdx.z;
    if (cb < rb) return;
    if (done && done[n]) return;
    if (cb < skip_lead) return;                 // (rb <= cb: the block lies in the part phase A computed)
    const int cnt = counts[n];
    if (rb * 64 >= cnt || cb * 64 >= cnt) return;
    __shared__ float cbx[64][4];
    const int lane = threadIdx.x;
    const float *bb = boxes + (long)n * cap * 4;
    const int j0 = cb * 64;
    if (j0 + lane < cnt) {
        const float *s = bb + (long)(j0 + lane) * 4;
        cbx[lane][0] = s[0];
        cbx[lane][1] = s[1];
        cbx[lane][2] = s[2];
        cbx[lane][3] = s[3];
    }
    __syncthreads();
    const int i = rb * 64 + lane;
    if (i >= cnt) return;
    const float *s = bb + (long)i * 4;
    const float ix1 = s[0], iy1 = s[1], ix2 = s[2], iy2 = s[3];
    const float ia = (ix2 - ix1) * (iy2 - iy1);
    unsigned long long bits = 0;
    const int jn = min(64, cnt - j0);
    for (int b = 0; b < jn; ++b) {
        const int jdx = j0 + b;
        if (jdx <= i) continue;
        const float xx1 = fmaxf(ix1, cbx[b][0]), yy1 = fmaxf(iy1, cbx[b][1]);
        const float xx2 = fminf(ix2, cbx[b][2]), yy2 = fminf(iy2, cbx[b][3]);
        const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
        const float inter = w * h;
        const float ja = (cbx[b][2] - cbx[b][0]) * (cbx[b][3] - cbx[b][1]);
        const float ovr = inter / (ia + ja - inter);
        if ((double)ovr > thr) bits |= 1ull << b;
    }
    mask[((long)n * cap + i) * nwords + cb] = bits;
}

// ---------------------------------------------------------------------------
// Greedy sweep.  128 threads: thread t owns removed-word t (nwords <= 128).  keep_idx [n][max_keep] i32.
__global__ __launch_bounds__(128) void nms_scan_kernel(const unsigned long long *__restrict__ mask,
                                                       const int32_t *__restrict__ valid /* [n][cap] or null */,
                                                       const int32_t *__restrict__ counts, int cap, int nwords,
                                                       int max_keep, int32_t *__restrict__ keep_idx,
                                                       int32_t *__restrict__ keep_count, int max_chunks /* <= 0: all */,
                                                       int32_t *__restrict__ done /* phase A writes, phase B reads; or null */) {
    const int n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int cnt = counts[n];
    const int nw_all = (cnt + 63) >> 6;
    if (max_chunks <= 0 && done && done[n]) return;          // phase B: phase A's sweep was complete for this image
    const int nw = max_chunks > 0 && max_chunks < nw_all ? max_chunks : nw_all;     // words (64-box chunks) this sweep may touch
    const unsigned long long *mk = mask + (long)n * cap * nwords;
    __shared__ unsigned long long sh_cur, sh_kept;
    __shared__ int sh_total;

    // removed-word initialised with the invalid (size-filtered / out-of-range) boxes
    unsigned long long remv = 0;
    if (tid < nw) {
        for (int b = 0; b < 64; ++b) {
            int i = tid * 64 + b;
            bool ok = i < cnt && (valid == nullptr || valid[(long)n * cap + i] != 0);
            if (!ok) remv |= 1ull << b;
        }
    }
    int total = 0;
    for (int c = 0; c < nw; ++c) {
        if (tid == c) sh_cur = remv;
        __syncthreads();
        if (tid < 64) {
            unsigned long long cur = sh_cur;
            const int i = c * 64 + lane;
            unsigned long long diag = (i < cnt) ? mk[(long)i * nwords + c] : 0ull;
            const uint32_t dlo = (uint32_t)diag, dhi = (uint32_t)(diag >> 32);
            unsigned long long kept = 0;
#pragma unroll
            for (int b = 0; b < 64; ++b) {
                if (!((cur >> b) & 1ull)) {
                    kept |= 1ull << b;
                    // readlane returns a signed int: go through uint32_t or the low word sign-extends
                    unsigned long long db = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(dhi, b) << 32) |
                                            (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(dlo, b);
                    cur |= db;
                }
            }
            // honour the post-NMS cap: only the first (max_keep - total) kept boxes count
            int room = max_keep - total;
            int pc = __popcll(kept);
            if (pc > room) {
                // drop the highest kept bits beyond `room`
                unsigned long long k2 = kept;
                int seen = 0;
                unsigned long long out = 0;
                while (k2 && seen < room) {
                    int b = __builtin_ctzll(k2);
                    out |= 1ull << b;
                    k2 &= k2 - 1;
                    ++seen;
                }
                kept = out;
                pc = room;
            }
            if ((kept >> lane) & 1ull) {
                int pos = total + __popcll(kept & ((1ull << lane) - 1ull));
                keep_idx[(long)n * max_keep + pos] = i;
            }
            if (lane == 0) {
                sh_kept = kept;
                sh_total = total + pc;
            }
        }
        __syncthreads();
        total = sh_total;
        if (total >= max_keep) break;
        unsigned long long kept = sh_kept;
        if (tid > c && tid < nw) {
            while (kept) {
                int b = __builtin_ctzll(kept);
                kept &= kept - 1;
                remv |= mk[(long)(c * 64 + b) * nwords + tid];
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        keep_count[n] = total;
        // phase A: complete if the cap was reached or every box was swept (then phase B has nothing to add)
        if (max_chunks > 0 && done) done[n] = (total >= max_keep || nw >= nw_all) ? 1 : 0;
    }
}

// proposals out: boxes [n][R][4], logits [n][R], rois [n*R][5] (batch index + box; zero box beyond count)
__global__ void rpn_gather_kernel(const float *__restrict__ cand_boxes, const float *__restrict__ cand_logit, int cap,
                                  const int32_t *__restrict__ keep_idx, const int32_t *__restrict__ keep_count, int R,
                                  float *__restrict__ out_boxes, float *__restrict__ out_logits,
                                  int32_t *__restrict__ out_counts) {
    const int n = blockIdx.x;
    const int cnt = min(keep_count[n], R);
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, lgt = 0.f;
        if (r < cnt) {
            const int i = keep_idx[(long)n * R + r];
            const float *s = cand_boxes + ((long)n * cap + i) * 4;
            b0 = s[0];
            b1 = s[1];
            b2 = s[2];
            b3 = s[3];
            lgt = cand_logit[(long)n * cap + i];
        }
        float *o = out_boxes + ((long)n * R + r) * 4;
        o[0] = b0;
        o[1] = b1;
        o[2] = b2;
        o[3] = b3;
        out_logits[(long)n * R + r] = lgt;
    }
    if (threadIdx.x == 0) out_counts[n] = cnt;
}

// standalone nms (vk_nms): composite-key bitonic sort of up to 8192 scores, then gather boxes in order
__global__ __launch_bounds__(RPN_THREADS) void sort_scores_kernel(const float *__restrict__ scores, int n, int sortn,
                                                                  const float *__restrict__ boxes,
                                                                  float *__restrict__ sorted_boxes,
                                                                  int32_t *__restrict__ order, int32_t *__restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long *skey = reinterpret_cast<unsigned long long *>(smem_raw);
    const int tid = threadIdx.x;
    for (int i = tid; i < sortn; i += RPN_THREADS)
        skey[i] = i < n ? (((unsigned long long)desc_key(scores[i]) << 32) | (uint32_t)i) : ~0ull;
    __syncthreads();
    for (int k2 = 2; k2 <= sortn; k2 <<= 1)
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (int i = tid; i < sortn; i += RPN_THREADS) {
                int ixj = i ^ j2;
                if (ixj > i) {
                    unsigned long long a = skey[i], b = skey[ixj];
                    bool up = (i & k2) == 0;
                    if ((a > b) == up) {
                        skey[i] = b;
                        skey[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    for (int e = tid; e < n; e += RPN_THREADS) {
        int i = (int)(skey[e] & 0xFFFFFFFFull);
        order[e] = i;
        sorted_boxes[4 * e + 0] = boxes[4 * i + 0];
        sorted_boxes[4 * e + 1] = boxes[4 * i + 1];
        sorted_boxes[4 * e + 2] = boxes[4 * i + 2];
        sorted_boxes[4 * e + 3] = boxes[4 * i + 3];
    }
    if (tid == 0) count[0] = n;
}

__global__ void nms_emit_kernel(const int32_t *__restrict__ order, const int32_t *__restrict__ keep_idx,
                                const int32_t *__restrict__ keep_count, int64_t *__restrict__ keep_out,
                                int32_t *__restrict__ count_out) {
    const int cnt = keep_count[0];
    for (int r = threadIdx.x; r < cnt; r += blockDim.x) keep_out[r] = (int64_t)order[keep_idx[r]];
    if (threadIdx.x == 0) count_out[0] = cnt;
}

__global__ void box_decode_kernel(const float *__restrict__ deltas, const float *__restrict__ boxes, long total, int k,
                                  float wx, float wy, float ww, float wh, float clampv, float *__restrict__ out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    long m = i / k;
    float a[4] = {boxes[4 * m], boxes[4 * m + 1], boxes[4 * m + 2], boxes[4 * m + 3]};
    float d[4] = {deltas[4 * i], deltas[4 * i + 1], deltas[4 * i + 2], deltas[4 * i + 3]};
    float o[4];
    apply_deltas1(a, d, wx, wy, ww, wh, clampv, o);
    out[4 * i] = o[0];
    out[4 * i + 1] = o[1];
    out[4 * i + 2] = o[2];
    out[4 * i + 3] = o[3];
}

static int next_pow2(int v) {
    int p = 2;
    while (p < v) p <<= 1;
    return p;
}

// ---------------------------------------------------------------------------
// Several levels (find_top_rpn_proposals frcnn.py:264-390 with a list of levels; N4): the per-level candidates (each
// level's top-k, decoded / clipped / size-flagged by rpn_select_decode_kernel) are merged per image in the concat order
// (level-major, rank-minor), the size-filtered ones dropped, and sorted by logit (ties -> lower concat index).  NMS
// runs on boxes shifted by level * (max coordinate + 1) exactly like torchvision's batched_nms (frcnn.py:383), so boxes
// of different levels never overlap; the un-shifted boxes are what comes out.
constexpr int RPN_MAX_LEVELS = 6;
struct LevelCands {
    const float *boxes[RPN_MAX_LEVELS];
    const float *logit[RPN_MAX_LEVELS];
    const int32_t *valid[RPN_MAX_LEVELS];
    const int32_t *count[RPN_MAX_LEVELS];
    int levels, pre;
};

__global__ __launch_bounds__(RPN_THREADS) void rpn_merge_levels_kernel(LevelCands lc, int cap, int sortn, float *__restrict__ m_boxes,
                                                                       float *__restrict__ m_shift, float *__restrict__ m_logit,
                                                                       int32_t *__restrict__ m_count) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long *skey = reinterpret_cast<unsigned long long *>(smem_raw);   // [sortn]
    float *red = reinterpret_cast<float *>(skey + sortn);                          // [RPN_THREADS / 64]
    __shared__ int sh_cnt;
    const int n = blockIdx.x, tid = threadIdx.x, pre = lc.pre;
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    float mx = -INFINITY;
    int local = 0;
    for (int i = tid; i < sortn; i += RPN_THREADS) {
        unsigned long long key = ~0ull;
        if (i < lc.levels * pre) {
            const int l = i / pre, r = i - l * pre;
            if (r < lc.count[l][n] && lc.valid[l][(long)n * pre + r]) {
                key = ((unsigned long long)desc_key(lc.logit[l][(long)n * pre + r]) << 32) | (uint32_t)i;
                const float *b = lc.boxes[l] + ((long)n * pre + r) * 4;
                mx = fmaxf(mx, fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])));
                ++local;
            }
        }
        skey[i] = key;
    }
    atomicAdd(&sh_cnt, local);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < RPN_THREADS / 64; ++w) mx = fmaxf(mx, red[w]);
    const int cnt = sh_cnt;
    for (int k2 = 2; k2 <= sortn; k2 <<= 1)
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (int i = tid; i < sortn; i += RPN_THREADS) {
                int ixj = i ^ j2;
                if (ixj > i) {
                    unsigned long long a = skey[i], b = skey[ixj];
                    bool up = (i & k2) == 0;
                    if ((a > b) == up) {
                        skey[i] = b;
                        skey[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    const float step = mx + 1.0f;                       // offsets = idxs.to(boxes) * (max_coordinate + 1)
    for (int e = tid; e < cnt; e += RPN_THREADS) {
        const uint32_t slot = (uint32_t)(skey[e] & 0xFFFFFFFFull);
        const int l = slot / pre, r = slot - l * pre;
        const float *b = lc.boxes[l] + ((long)n * pre + r) * 4;
        const float off = (float)l * step;
        float *ob = m_boxes + ((long)n * cap + e) * 4, *os = m_shift + ((long)n * cap + e) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ob[c] = b[c];
            os[c] = b[c] + off;
        }
        m_logit[(long)n * cap + e] = lc.logit[l][(long)n * pre + r];
    }
    if (tid == 0) m_count[n] = cnt;
}

struct RpnWs {
    float *cand_boxes;
    float *cand_logit;
    int32_t *cand_valid;
    int32_t *cand_count;
    unsigned long long *mask;
    int32_t *keep_idx;
    int32_t *keep_count;
    size_t total;
};

static RpnWs carve_rpn_ws(void *base, int N, int pre, int post) {
    RpnWs w;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *r = p ? p + off : nullptr;
        off += align_up(bytes, 256);
        return r;
    };
    const int nwords = ceil_div(pre, 64);
    w.cand_boxes = (float *)take((size_t)N * pre * 4 * sizeof(float));
    w.cand_logit = (float *)take((size_t)N * pre * sizeof(float));
    w.cand_valid = (int32_t *)take((size_t)N * pre * sizeof(int32_t));
    w.cand_count = (int32_t *)take((size_t)N * sizeof(int32_t));
    w.mask = (unsigned long long *)take((size_t)N * pre * nwords * 8);
    w.keep_idx = (int32_t *)take((size_t)N * (post > pre ? post : pre) * sizeof(int32_t));
    w.keep_count = (int32_t *)take((size_t)2 * N * sizeof(int32_t));          // [N] counts, [N] phase-A-complete flags (launch_nms)
    w.total = off;
    return w;
}

// Greedy NMS of N images' sorted candidates (boxes [N][cap][4], counts [N]): suppression masks + sweep, in two phases.  The sweep stops
// at `max_keep` kept boxes, so only the boxes ahead of the one that reaches the cap matter: on the bench's input the 300th kept
// proposal is the ~400th of 6000 candidates, and the full 6000 x 6000 mask was 474 us of IoU tests per batch for a sweep that read 0.5 %
// of it.  Phase A: the leading `lead` boxes (2 x max_keep, at least 1024) -- mask blocks and sweep; phase B, for the images whose sweep
// ran out of rows before reaching the cap: the rest of the mask and the sweep again from the start (greedy NMS is causal, so phase A's
// result is exact whenever it completes).  No host round trip: phase B's workgroups look at a per-image flag and leave.
static int launch_nms(const float *boxes, const int32_t *valid, const int32_t *counts, int N, int cap, int nb /* 64-box blocks in use */,
                      double thr, int max_keep, unsigned long long *mask, int32_t *keep_idx, int32_t *keep_count, int32_t *done,
                      hipStream_t s) {
    const int nwords = ceil_div(cap, 64);
    const char *e = getenv("VK_NMS_LEAD");                    // boxes of phase A; "0": one phase (A/B switch and tests; re-read per call)
    int lead = e ? atoi(e) : (2 * max_keep > 1024 ? 2 * max_keep : 1024);
    const int na = ceil_div(lead, 64);
    if ((e && lead <= 0) || na >= nb || !done) {
        hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb, N), dim3(64), 0, s, boxes, counts, cap, nwords, thr, mask, 0, (const int32_t *)nullptr);
        VK_CHECK_HIP(hipGetLastError());
        hipLaunchKernelGGL(nms_scan_kernel, dim3(N), dim3(128), 0, s, mask, valid, counts, cap, nwords, max_keep, keep_idx, keep_count, 0,
                           (int32_t *)nullptr);
        VK_CHECK_HIP(hipGetLastError());
        return VK_OK;
    }
    hipLaunchKernelGGL(nms_mask_kernel, dim3(na, na, N), dim3(64), 0, s, boxes, counts, cap, nwords, thr, mask, 0, (const int32_t *)nullptr);
    VK_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(nms_scan_kernel, dim3(N), dim3(128), 0, s, mask, valid, counts, cap, nwords, max_keep, keep_idx, keep_count, na, done);
    VK_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb, N), dim3(64), 0, s, boxes, counts, cap, nwords, thr, mask, na, (const int32_t *)done);
    VK_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(nms_scan_kernel, dim3(N), dim3(128), 0, s, mask, valid, counts, cap, nwords, max_keep, keep_idx, keep_count, 0, done);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

}  // namespace vk

using namespace vk;

extern "C" {

size_t vk_rpn_workspace_bytes(int N, int HWA, int pre_topk) {
    int pre = pre_topk < HWA ? pre_topk : HWA;
    (void)pre;
    return carve_rpn_ws(nullptr, N, pre_topk, pre_topk).total;
}

int vk_rpn_proposals(const float *logits, int ld_logits, const float *deltas, int ld_deltas, int N, int Hf, int Wf, int A,
                     const float *cell_anchors, int stride, float offset, const int32_t *image_hw,
                     const float *bbox_weights4_host, float min_size, double nms_thresh, int pre_topk, int post_topk,
                     float *out_boxes, float *out_logits, int32_t *out_counts, int32_t *nonfinite_flag, void *workspace,
                     size_t workspace_bytes, void *stream) {
    VK_REQUIRE(N > 0 && Hf > 0 && Wf > 0 && A > 0, VK_EINVAL, "rpn: empty problem");
    VK_REQUIRE(pre_topk > 0 && pre_topk <= RPN_MAX_PRE, VK_EINVAL, "rpn: pre_nms_topk=%d must be in 1..%d", pre_topk, RPN_MAX_PRE);
    VK_REQUIRE(post_topk > 0 && post_topk <= pre_topk, VK_EINVAL, "rpn: post_nms_topk=%d must be in 1..pre_nms_topk", post_topk);
    const long HWA = (long)Hf * Wf * A;
    VK_REQUIRE(HWA < (1L << 31), VK_EINVAL, "rpn: too many anchors");
    RpnWs w = carve_rpn_ws(workspace, N, pre_topk, post_topk);
    VK_REQUIRE(workspace && workspace_bytes >= w.total, VK_EINVAL, "rpn: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    const int K = (int)(pre_topk < HWA ? pre_topk : HWA);
    const int sortn = next_pow2(K);
    DecodeCfg cfg;
    cfg.wx = bbox_weights4_host[0];
    cfg.wy = bbox_weights4_host[1];
    cfg.ww = bbox_weights4_host[2];
    cfg.wh = bbox_weights4_host[3];
    cfg.scale_clamp = (float)log(1000.0 / 16.0);   // frcnn.py:510
    cfg.min_size = min_size;
    cfg.stride = stride;
    cfg.offset = offset;
    cfg.stop = 0;
#ifdef VK_ABLATION
    if (const char *e = getenv("VK_RPN_STOP")) cfg.stop = atoi(e);
#endif
    const size_t smem = (size_t)sortn * 8 + 2048;
    static bool attr_set = false;
    if (!attr_set) {
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rpn_select_decode_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, RPN_MAX_PRE * 8 + 2048));
        attr_set = true;
    }
    hipLaunchKernelGGL(rpn_select_decode_kernel, dim3(N), dim3(RPN_THREADS), smem, s, logits, ld_logits, deltas, ld_deltas,
                       Hf, Wf, A, cell_anchors, image_hw, cfg, pre_topk, sortn, w.cand_boxes, w.cand_logit, w.cand_valid,
                       w.cand_count, nonfinite_flag);
    VK_CHECK_HIP(hipGetLastError());
    const int nb = ceil_div(K, 64);
    VK_TRY(launch_nms(w.cand_boxes, w.cand_valid, w.cand_count, N, pre_topk, nb, nms_thresh, post_topk, w.mask, w.keep_idx, w.keep_count,
                      w.keep_count + N, s));
    hipLaunchKernelGGL(rpn_gather_kernel, dim3(N), dim3(256), 0, s, w.cand_boxes, w.cand_logit, pre_topk, w.keep_idx,
                       w.keep_count, post_topk, out_boxes, out_logits, out_counts);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

struct MlWs {
    RpnWs lv[RPN_MAX_LEVELS];
    float *m_boxes, *m_shift, *m_logit;
    int32_t *m_count;
    unsigned long long *mask;
    int32_t *keep_idx, *keep_count;
    size_t total;
};

static MlWs carve_ml_ws(void *base, int N, int levels, int pre, int post) {
    MlWs w;
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *r = p ? p + off : nullptr;
        off += align_up(bytes, 256);
        return r;
    };
    for (int l = 0; l < levels; ++l) {
        w.lv[l].cand_boxes = (float *)take((size_t)N * pre * 4 * sizeof(float));
        w.lv[l].cand_logit = (float *)take((size_t)N * pre * sizeof(float));
        w.lv[l].cand_valid = (int32_t *)take((size_t)N * pre * sizeof(int32_t));
        w.lv[l].cand_count = (int32_t *)take((size_t)N * sizeof(int32_t));
    }
    const int cap = levels * pre, nwords = ceil_div(cap, 64);
    w.m_boxes = (float *)take((size_t)N * cap * 4 * sizeof(float));
    w.m_shift = (float *)take((size_t)N * cap * 4 * sizeof(float));
    w.m_logit = (float *)take((size_t)N * cap * sizeof(float));
    w.m_count = (int32_t *)take((size_t)N * sizeof(int32_t));
    w.mask = (unsigned long long *)take((size_t)N * cap * nwords * 8);
    w.keep_idx = (int32_t *)take((size_t)N * post * sizeof(int32_t));
    w.keep_count = (int32_t *)take((size_t)2 * N * sizeof(int32_t));          // [N] counts, [N] phase-A-complete flags (launch_nms)
    w.total = off;
    return w;
}

size_t vk_rpn_multilevel_workspace_bytes(int N, int levels, int pre_topk, int post_topk) {
    if (levels < 1 || levels > RPN_MAX_LEVELS) return 0;
    return carve_ml_ws(nullptr, N, levels, pre_topk, post_topk).total;
}

int vk_rpn_proposals_multilevel(const float *const *logits, const int32_t *ld_logits, const float *const *deltas, const int32_t *ld_deltas,
                                int levels, int N, const int32_t *Hs, const int32_t *Ws, int A, const float *const *cell_anchors,
                                const int32_t *strides, float offset, const int32_t *image_hw, const float *bbox_weights4_host,
                                float min_size, double nms_thresh, int pre_topk, int post_topk, float *out_boxes, float *out_logits,
                                int32_t *out_counts, int32_t *nonfinite_flag, void *workspace, size_t workspace_bytes, void *stream) {
    VK_REQUIRE(logits && deltas && Hs && Ws && cell_anchors && strides && image_hw && bbox_weights4_host, VK_EINVAL, "rpn_ml: null argument");
    VK_REQUIRE(levels >= 1 && levels <= RPN_MAX_LEVELS && N > 0 && A > 0, VK_EINVAL, "rpn_ml: 1..%d levels", RPN_MAX_LEVELS);
    VK_REQUIRE(pre_topk > 0 && (long)levels * pre_topk <= RPN_MAX_PRE, VK_EINVAL,
               "rpn_ml: levels * pre_nms_topk = %ld must be in 1..%d", (long)levels * pre_topk, RPN_MAX_PRE);
    VK_REQUIRE(post_topk > 0 && post_topk <= levels * pre_topk, VK_EINVAL, "rpn_ml: post_nms_topk=%d out of range", post_topk);
    MlWs w = carve_ml_ws(workspace, N, levels, pre_topk, post_topk);
    VK_REQUIRE(workspace && workspace_bytes >= w.total, VK_EINVAL, "rpn_ml: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rpn_select_decode_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, RPN_MAX_PRE * 8 + 2048));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rpn_select_decode_levels_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, RPN_MAX_PRE * 8 + 2048));
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rpn_merge_levels_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, RPN_MAX_PRE * 8 + 256));
        attr_set = true;
    }
    LevelCands lc;
    memset(&lc, 0, sizeof(lc));
    lc.levels = levels;
    lc.pre = pre_topk;
    static_assert(RPN_SEL_LEVELS >= RPN_MAX_LEVELS, "one slot per level");
    SelectLevels sl;
    memset(&sl, 0, sizeof(sl));
    int max_sortn = 2;
    for (int l = 0; l < levels; ++l) {
        const long HWA = (long)Hs[l] * Ws[l] * A;
        VK_REQUIRE(Hs[l] > 0 && Ws[l] > 0 && HWA < (1L << 31), VK_EINVAL, "rpn_ml: level %d has a bad size", l);
        const int K = (int)(pre_topk < HWA ? pre_topk : HWA);
        sl.logits[l] = logits[l];
        sl.deltas[l] = deltas[l];
        sl.cell_anchors[l] = cell_anchors[l];
        sl.ld_logits[l] = ld_logits[l];
        sl.ld_deltas[l] = ld_deltas[l];
        sl.Hf[l] = Hs[l];
        sl.Wf[l] = Ws[l];
        sl.stride[l] = strides[l];
        sl.sortn[l] = next_pow2(K);
        max_sortn = sl.sortn[l] > max_sortn ? sl.sortn[l] : max_sortn;
        sl.cand_boxes[l] = w.lv[l].cand_boxes;
        sl.cand_logit[l] = w.lv[l].cand_logit;
        sl.cand_valid[l] = w.lv[l].cand_valid;
        sl.cand_count[l] = w.lv[l].cand_count;
        lc.boxes[l] = w.lv[l].cand_boxes;
        lc.logit[l] = w.lv[l].cand_logit;
        lc.valid[l] = w.lv[l].cand_valid;
        lc.count[l] = w.lv[l].cand_count;
    }
    DecodeCfg cfg;
    cfg.wx = bbox_weights4_host[0];
    cfg.wy = bbox_weights4_host[1];
    cfg.ww = bbox_weights4_host[2];
    cfg.wh = bbox_weights4_host[3];
    cfg.scale_clamp = (float)log(1000.0 / 16.0);
    cfg.min_size = min_size;
    cfg.stride = 0;                                  // per level, from sl.stride
    cfg.offset = offset;
    cfg.stop = 0;
    hipLaunchKernelGGL(rpn_select_decode_levels_kernel, dim3(N, levels), dim3(RPN_THREADS), (size_t)max_sortn * 8 + 2048, s, sl, A, image_hw, cfg,
                       pre_topk, nonfinite_flag);
    VK_CHECK_HIP(hipGetLastError());
    const int cap = levels * pre_topk, sortn = next_pow2(cap), nwords = ceil_div(cap, 64);
    hipLaunchKernelGGL(rpn_merge_levels_kernel, dim3(N), dim3(RPN_THREADS), (size_t)sortn * 8 + 256, s, lc, cap, sortn, w.m_boxes, w.m_shift,
                       w.m_logit, w.m_count);
    VK_CHECK_HIP(hipGetLastError());
    VK_TRY(launch_nms(w.m_shift, nullptr, w.m_count, N, cap, nwords, nms_thresh, post_topk, w.mask, w.keep_idx, w.keep_count, w.keep_count + N, s));
    hipLaunchKernelGGL(rpn_gather_kernel, dim3(N), dim3(256), 0, s, w.m_boxes, w.m_logit, cap, w.keep_idx, w.keep_count, post_topk, out_boxes,
                       out_logits, out_counts);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

size_t vk_nms_workspace_bytes(int n) {
    int cap = n < 1 ? 1 : n;
    // sorted boxes, order, count + the RPN-style carve (capacity n, one image)
    return align_up((size_t)cap * 16, 256) + align_up((size_t)cap * 4, 256) + 256 + carve_rpn_ws(nullptr, 1, cap, cap).total;
}

int vk_nms(const float *boxes, const float *scores, int n, double thresh, int64_t *keep_out, int32_t *count_out,
           void *workspace, size_t workspace_bytes, void *stream) {
    VK_REQUIRE(n >= 0 && n <= RPN_MAX_PRE, VK_EINVAL, "nms: n=%d must be in 0..%d", n, RPN_MAX_PRE);
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        VK_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int32_t), s));
        return VK_OK;
    }
    VK_REQUIRE(workspace && workspace_bytes >= vk_nms_workspace_bytes(n), VK_EINVAL, "nms: workspace too small");
    char *p = (char *)workspace;
    float *sorted_boxes = (float *)p;
    p += align_up((size_t)n * 16, 256);
    int32_t *order = (int32_t *)p;
    p += align_up((size_t)n * 4, 256);
    int32_t *cnt = (int32_t *)p;
    p += 256;
    RpnWs w = carve_rpn_ws(p, 1, n, n);
    const int sortn = next_pow2(n);
    static bool attr_set = false;
    if (!attr_set) {
        VK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sort_scores_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, RPN_MAX_PRE * 8));
        attr_set = true;
    }
    hipLaunchKernelGGL(sort_scores_kernel, dim3(1), dim3(RPN_THREADS), (size_t)sortn * 8, s, scores, n, sortn, boxes,
                       sorted_boxes, order, cnt);
    VK_CHECK_HIP(hipGetLastError());
    const int nwords = ceil_div(n, 64);
    // (max_keep = n: one phase unless VK_NMS_LEAD forces a short phase A -- which is how the tests reach phase B)
    VK_TRY(launch_nms(sorted_boxes, nullptr, cnt, 1, n, nwords, thresh, n, w.mask, w.keep_idx, w.keep_count, w.keep_count + 1, s));
    hipLaunchKernelGGL(nms_emit_kernel, dim3(1), dim3(256), 0, s, order, w.keep_idx, w.keep_count, keep_out, count_out);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

int vk_box_decode(const float *deltas, const float *boxes, int M, int k, const float *weights4_host, float *out,
                  void *stream) {
    VK_REQUIRE(M >= 0 && k > 0, VK_EINVAL, "box_decode: bad sizes");
    if (M == 0) return VK_OK;
    long total = (long)M * k;
    hipLaunchKernelGGL(box_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, deltas,
                       boxes, total, k, weights4_host[0], weights4_host[1], weights4_host[2], weights4_host[3],
                       (float)log(1000.0 / 16.0), out);
    VK_CHECK_HIP(hipGetLastError());
    return VK_OK;
}

}  // extern "C"
