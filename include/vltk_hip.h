/*
 * vltk_hip.h -- C ABI of libvltk_hip.so: the MI355X (gfx950) implementation of
 * vltk's Faster R-CNN visual-feature extraction forward pass.
 *
 * This is the drop-in boundary.  The reference has no native layer (it is
 * pure Python over torch/torchvision), so every entry point below replaces a
 * Python-level interface of /root/reference/vltk/modeling/frcnn.py; the
 * file:line each one stands in for is given per function.  A maintainer binds
 * these with ctypes; vltk_amd/_lib.py is that binding.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - Every function returns an int status: VK_OK (0) or a VK_E* code;
 *     vk_last_error() returns a thread-local message for the last failure.
 *   - "dev" pointers are device (HBM) pointers owned by the caller; "host"
 *     pointers are ordinary host memory.  `stream` is a hipStream_t passed as
 *     void* (NULL = the default stream).
 *   - Activations are NHWC ("pixel-major": a feature map is a row-major
 *     [N*H*W, C] matrix), dtype per vk_dtype; boxes are (x1,y1,x2,y2) f32.
 */
#ifndef VLTK_HIP_H
#define VLTK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VK_OK 0
#define VK_EINVAL 1      /* bad argument / unsupported configuration  -> ValueError      */
#define VK_ENOTIMPL 2    /* reference raises NotImplementedError (frcnn.py:1930)         */
#define VK_ENONFINITE 3  /* non-finite box: reference's assert in _clip_box (frcnn.py:148) -> AssertionError */
#define VK_EWEIGHTS 4    /* missing / mis-shaped weight (strict load, frcnn.py:1881)     -> OSError */
#define VK_EHIP 5        /* HIP runtime failure                                          -> RuntimeError */
#define VK_ENOMEM 6

typedef enum { VK_F32 = 0, VK_F16 = 1, VK_I64 = 2, VK_I32 = 3, VK_BF16 = 4 } vk_dtype;
/* epilogue activation of vk_conv2d / vk_linear (the `relu` argument): */
typedef enum { VK_ACT_NONE = 0, VK_ACT_RELU = 1, VK_ACT_GELU = 2 /* erf form */, VK_ACT_TANH = 3 } vk_act;

#define VK_MAX_ANCHOR_DIM 8
#define VK_MAX_NMS_THRESH 8

/* The config keys the reference model reads (frcnn.py:200-223,
 * 1230-1237, 1312-1336, 1414-1417, 1537, 1583-1607). */
typedef struct vk_config {
    int32_t depth;                 /* RESNETS.DEPTH 50|101|152 */
    int32_t num_groups;            /* RESNETS.NUM_GROUPS (frcnn.py:217; > 1: ResNeXt, width_per_group a power of two) */
    int32_t width_per_group;       /* RESNETS.WIDTH_PER_GROUP */
    int32_t stem_out_channels;     /* RESNETS.STEM_OUT_CHANNELS */
    int32_t res2_out_channels;     /* RESNETS.RES2_OUT_CHANNELS */
    int32_t stride_in_1x1;         /* RESNETS.STRIDE_IN_1X1 */
    int32_t caffe_maxpool;         /* MODEL.MAX_POOL */
    int32_t num_sizes;             /* ANCHOR_GENERATOR.SIZES[0] */
    float   sizes[VK_MAX_ANCHOR_DIM];
    int32_t num_ratios;            /* ANCHOR_GENERATOR.ASPECT_RATIOS[0] */
    float   ratios[VK_MAX_ANCHOR_DIM];
    float   anchor_offset;         /* ANCHOR_GENERATOR.OFFSET */
    int32_t rpn_hidden_channels;   /* PROPOSAL_GENERATOR.HIDDEN_CHANNELS (-1 = same as res4) */
    float   rpn_min_size;          /* PROPOSAL_GENERATOR.MIN_SIZE */
    double  rpn_nms_thresh;        /* RPN.NMS_THRESH */
    int32_t pre_nms_topk;          /* RPN.PRE_NMS_TOPK_TEST  (<= 8192) */
    int32_t post_nms_topk;         /* RPN.POST_NMS_TOPK_TEST (<= 1024) */
    float   rpn_bbox_weights[4];   /* RPN.BBOX_REG_WEIGHTS */
    int32_t num_classes;           /* ROI_HEADS.NUM_CLASSES */
    int32_t num_attrs;             /* ROI_BOX_HEAD.NUM_ATTRS */
    int32_t use_attr;              /* ROI_BOX_HEAD.ATTR */
    int32_t pooler_resolution;     /* ROI_BOX_HEAD.POOLER_RESOLUTION */
    int32_t res5_halve;            /* ROI_BOX_HEAD.RES5HALVE (only 0 supported this round) */
    int32_t cls_agnostic_bbox_reg; /* ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG */
    float   roi_bbox_weights[4];   /* ROI_BOX_HEAD.BBOX_REG_WEIGHTS */
    int32_t precision;             /* VK_F16: fp16 storage / fp32 accumulate (fast);
                                      VK_F32: fp32 storage, exact-f32 MFMA (strict parity mode) */
} vk_config;

/* Per-call knobs = the mutable attributes of the reference's model.roi_outputs
 * (frcnn.py:1229-1240; set by callers, tests/frcnn_test.py:16-19). */
typedef struct vk_roi_params {
    int32_t num_nms_thresh;
    double  nms_thresh[VK_MAX_NMS_THRESH];
    int32_t min_detections;
    int32_t max_detections;
} vk_roi_params;

/* Device output block of one forward (caller-allocated, fixed capacity D =
 * max_detections per image; rows >= preds_per_image[n] are zero).  Mirrors the
 * OrderedDict returned by FRCNN.inference (frcnn.py:1996-2004). */
typedef struct vk_outputs {
    int64_t *obj_ids;          /* [N, D]        */
    float   *obj_probs;        /* [N, D]        */
    int64_t *attr_ids;         /* [N, D]        */
    float   *attr_probs;       /* [N, D]        */
    float   *boxes;            /* [N, D, 4]     */
    int64_t *preds_per_image;  /* [N]           */
    float   *roi_features;     /* [N, D, 2048]  */
} vk_outputs;

typedef struct vk_handle vk_handle;

const char *vk_last_error(void);
int vk_version(void);

/* ---- model lifetime ------------------------------------------------------
 * vk_create        <- FRCNN.__init__            frcnn.py:1744-1755 (build_backbone :200-261,
 *                                               RPN :1580-1610, Res5ROIHeads :1312-1363)
 * vk_load_weights  <- load_state_dict, one call per state-dict tensor, reference key names
 *                                               frcnn.py:1862-1881
 * vk_finalize      <- model.eval() + BN folding / NHWC repack (frcnn.py:1920; Conv2d :794-822)
 * vk_destroy       <- Python GC
 */
int vk_create(const vk_config *cfg, int device, vk_handle **out);
int vk_load_weights(vk_handle *h, const char *name, const void *host_ptr,
                    const int64_t *shape, int ndim, vk_dtype dtype);
int vk_finalize(vk_handle *h);
int vk_destroy(vk_handle *h);

/* Tunables.  "head_chunk": RoIs per Res5-head chunk (0 = all RoIs in one pass; default 9600 or
 * the VK_HEAD_CHUNK environment variable).  Results do not depend on it. */
int vk_set_option(vk_handle *h, const char *key, int value);

/* Number of weight tensors the model expects and their names (strict load). */
int vk_num_weights(vk_handle *h, int *count);
int vk_weight_name(vk_handle *h, int index, const char **name);

/* ---- the forward pass ----------------------------------------------------
 * vk_forward <- FRCNN.forward / inference   frcnn.py:1924-2004
 *   images_dev : [N,3,H,W] f32 NCHW, already resized / mean-subtracted / zero-padded
 *   image_hw   : host [N,2] int32 (h, w) of the un-padded content   (image_shapes)
 *   scales_yx  : host [N,2] f32 or NULL                              (frcnn.py:1280-1283)
 * Asynchronous on `stream` except for one small device->host read of the
 * non-finite flag at the end (the reference asserts on host, frcnn.py:148).
 */
int vk_forward(vk_handle *h, const float *images_dev, int N, int H, int W,
               const int32_t *image_hw, const float *scales_yx,
               const vk_roi_params *rp, const vk_outputs *out_dev, void *stream);

/* The same forward in two halves, so that a caller can enqueue the next batch before the previous one has finished
 * (the reference's loop is strictly serial, abc/extraction.py:189-213; on the GPU that leaves the device idle while the
 * host formats one batch and launches the next).  vk_forward_begin enqueues everything on `stream` and returns a
 * ticket; vk_forward_end(ticket) waits for that forward only, and raises the non-finite assertion (frcnn.py:148).
 * Tickets must be ended in order; at most 4 may be open.  All forwards of a handle share its workspace: they must be
 * enqueued on the same stream (they are stream-ordered, not concurrent); `image_hw` / `scales_yx` are consumed before
 * _begin returns; the output buffers of different tickets must be distinct.  vk_forward == begin + end. */
int vk_forward_begin(vk_handle *h, const float *images_dev, int N, int H, int W,
                     const int32_t *image_hw, const float *scales_yx,
                     const vk_roi_params *rp, const vk_outputs *out_dev, void *stream, int64_t *ticket);
int vk_forward_end(vk_handle *h, int64_t ticket);

/* Intermediate tensors of the last forward, for stage-level parity tests.
 * name in {"res4","rpn_out","proposal_boxes","proposal_logits",
 * "proposal_counts","pooled","feature_pooled","obj_logits","attr_logits","chosen_deltas","keep_ids"};
 * "rpn_out" is the fused RPN head output [N,Hf,Wf,ld]: columns [0,A) objectness, [A,5A) deltas.
 * Returns a device pointer owned by the handle (valid until the next forward),
 * its dtype and its shape (up to 4 dims). */
int vk_get_stage(vk_handle *h, const char *name, const void **dev_ptr,
                 vk_dtype *dtype, int64_t *shape, int *ndim);

/* device->device copy on `stream` (lets a binding copy a stage tensor into memory it owns). */
int vk_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes, void *stream);

/* timing of the last forward's stages (HIP events on the launch stream), ms:
 * [0] backbone [1] rpn head [2] proposals [3] roi pool+res5 head [4] predictor+outputs [5] total.
 * Only recorded when enabled (costs event records, no syncs in the timed path). */
int vk_enable_stage_timing(vk_handle *h, int enable);
int vk_get_stage_timing(vk_handle *h, float *ms6);

/* Per-kernel timing of the forward's convolution launches (HIP events on the launch stream around
 * every launch; read after the forward's own end-of-call synchronisation, accumulated until reset).
 * bucket 0: conv_mfma256_kernel (256x256 LDS-ring tile)          1: conv_mfma_kernel f16->f16 (128x{64,128} tile)
 * bucket 2: conv_mfma_kernel f16->f32 out (RPN heads, predictor)  3: f32 strict-mode convs / stem
 * bucket 4: conv3x3_panel_kernel (3x3, LDS-resident input panel)  5: conv_duo_kernel (1x1, 128x256 tile, two per CU)
 * bucket 6: any conv kernel launched in the two-stream section of the backbone (res3 / res4 half-batches): these launches
 *           overlap each other in time, so their summed durations exceed the wall time they took -- kept apart so that the
 *           buckets above hold only launches that had the GPU to themselves
 * bucket 7: conv3x3_blk_kernel (3x3 over narrow channel blocks: ResNeXt grouped conv2, dense 64 -> 64)
 * bucket 8: conv_ws_kernel (1x1, K <= 512, weight-stationary: a workgroup keeps its 256 x K weights in registers)
 * bucket 9: conv_mfma256_kernel<0, true> (the ring kernel's two-input build: conv3 + projection shortcut as one GEMM; its own
 *           symbol in a rocprofv3 trace)
 * bucket 10: conv_gemm4_kernel (1x1 with K >= 1024, one or two inputs: 256x256 tile, four waves of 128x128)
 * bucket 11: bneck64_kernel (a whole res2 BottleneckBlock as one kernel: conv1 -> 3x3 -> conv3 + shortcut, 64 bottleneck channels)
 * launches[12], ms[12], flops[12] (algorithmic 2*M*Cout*K of the launches), bytes[12] (algorithmic HBM bytes:
 * input + output (+ residual) + weights, each once). */
#define VK_NUM_KERNEL_BUCKETS 12
int vk_enable_kernel_timing(vk_handle *h, int enable);
int vk_get_kernel_timing(vk_handle *h, int64_t *launches, double *ms, double *flops, double *bytes, int reset);

/* ---- stage-level entry points (tests, micro-benchmarks) -------------------
 * All pointers are device pointers unless marked host.                      */

/* Packed-weight helpers (host side).  K is ordered (kh, kw, cin) and padded to
 * whole 128-byte K-tiles; rows are padded to a multiple of 128 output channels. */
size_t vk_packed_weight_bytes(int cout, int cin, int kh, int kw, int groups, vk_dtype dt);
int vk_packed_cout(int cout);
/* Grouped convolutions (ResNeXt, BottleneckBlock conv2 `groups=num_groups` frcnn.py:942-952) run as dense GEMMs over an
 * input-channel SLICE per 64-output-channel tile: slice = min(max(cin/groups, 64), cin) channels, weights
 * zero outside a channel's own group.  Needs cin == cout and cin/groups a power of two.  groups == 1: dense. */
int vk_conv_slice_channels(int cin, int groups);
/* w_oihw [cout,cin/groups,kh,kw] f32; bn = {gamma,beta,mean,var} each [cout] or NULL;
 * bias [cout] or NULL; outputs: packed weights (dtype dt) and f32 bias [vk_packed_cout]. */
int vk_pack_conv_weight(const float *w_oihw_host, const float *bn_host, const float *bias_host,
                        int cout, int cin, int kh, int kw, int groups, vk_dtype dt,
                        void *w_packed_host, float *bias_packed_host);

/* conv + folded-BN bias (+ residual) (+ ReLU)  <- Conv2d.forward frcnn.py:794-822,
 * BottleneckBlock.forward :963-979.  x [N,H,W,cin] (dt), residual/y [N*Ho*Wo, ldy]. */
int vk_conv2d(const void *x, int N, int H, int W, int cin,
              const void *w_packed, const float *bias_packed, const void *residual,
              void *y, int cout, int ldy, int kh, int kw, int stride, int pad, int dil, int groups,
              int relu, vk_dtype dt, vk_dtype out_dt, void *stream);

/* conv3 + projection shortcut of a stride-1 BottleneckBlock as ONE f16 GEMM (`out = conv3(t) ; out += shortcut(x)`,
 * frcnn.py:970-977): y[M,cout] = relu?([x1 | x2] . W^T + bias (+ residual)), W rows = [conv3 row (cin1) | shortcut row
 * (cin2)] as packed by vk_pack_conv_weight and concatenated per output channel, bias = the two folded biases
 * summed.  cout % 256 == 0, cin1 and cin2 multiples of 64 (whole 128-byte K-tiles of the packer). */
int vk_conv1x1_dual(const void *x1, int cin1, const void *x2, int cin2, long M,
                    const void *w_packed, const float *bias_packed, const void *residual,
                    void *y, int cout, int relu, void *stream);

/* A whole stride-1 BottleneckBlock with 64 bottleneck channels and 256 outputs (res2) as ONE f16 kernel
 * (BottleneckBlock.forward frcnn.py:963-979): y = relu(conv3(relu(conv2(relu(conv1 x)))) + shortcut(x)); the two 64-channel
 * intermediates stay in LDS (rounded to f16 there, as the layer-by-layer path rounds them in HBM), x is read once.
 * proj == 0: identity shortcut, cin == 256, w3 = conv3's packed rows [256][64].  proj != 0: projection shortcut of block 0,
 * cin == 64, w3 = [conv3 row | shortcut row] per output channel and b3 = the two folded biases summed (vk_conv1x1_dual's
 * layout).  w1 [>=64][cin], w2 [>=64][9*64] as vk_pack_conv_weight writes them.  x [N,H,W,cin], y [N,H,W,256]; N*H*W*512 < 2^31. */
int vk_bottleneck64(const void *x, int N, int H, int W, int cin, int proj,
                    const void *w1_packed, const float *b1_packed, const void *w2_packed, const float *b2_packed,
                    const void *w3_packed, const float *b3_packed, void *y, void *stream);

/* Last Res5 conv3 with the RoI's spatial mean folded into its epilogue (`res5(x).mean(dim=[2,3])`, frcnn.py:1401):
 * out_mean[n][c] = mean over the HW rows of image n of relu?(x . W^T + bias + residual), summed EXACTLY (integer
 * accumulation of the f16-rounded values) and rounded once, so it does not depend on how rows fall into tiles;
 * a non-finite value gives NaN.  The [N*HW, cout] tensor itself is never written.  f16, 128 <= HW <= 255,
 * cout % 256 == 0. */
size_t vk_conv1x1_meanpool_workspace_bytes(int N, int HW, int cout);
int vk_conv1x1_meanpool(const void *x, int N, int HW, int cin, const void *w_packed, const float *bias_packed,
                        const void *residual, int cout, int relu, float *out_mean,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---- N3: LXMERT-style cross-modality encoder ops (transformers LxmertModel, the consumer the reference feeds:
 * vltk/legacy/legacy_train.py:30-39; restated from transformers/models/lxmert/modeling_lxmert.py v5.15) ---- */

/* nn.Linear (+ residual) (+ activation): y[M, ldy] = act(x[M,K] . W^T + bias + residual).  W packed by
 * vk_pack_conv_weight(w [N,K,1,1], NULL, bias, ...); K a whole number of 128-byte K-tiles; dt f32 | f16 | bf16. */
int vk_linear(const void *x, long M, int K, const void *w_packed, const float *bias_packed, const void *residual,
              void *y, int N, int ldy, int act, vk_dtype dt, vk_dtype out_dt, void *stream);
/* y = scale * LayerNorm(x) (+ y when accumulate): LxmertAttentionOutput / LxmertOutput :269-342 (eps 1e-12),
 * LxmertVisualFeatureEncoder `(LN(a) + LN(b)) / 2` :468-476 as two calls with scale 0.5. */
int vk_layernorm(const void *x, int ldx, const float *gamma, const float *beta, void *y, int ldy, int M, int C,
                 float eps, float scale, int accumulate, vk_dtype dt, void *stream);
/* LxmertEmbeddings.forward :191-214: LayerNorm(word[ids] + position[arange(L)] + token_type[tt]); tables in dt. */
int vk_embed_layernorm(const int64_t *input_ids, const int64_t *token_type_ids, int B, int L, const void *word,
                       const void *position, const void *token_type, const float *gamma, const float *beta,
                       void *y, int C, float eps, vk_dtype dt, void *stream);
/* LxmertAttention.forward :238-266 after the three projections: out[b, i, h*d:(h+1)*d] =
 * softmax_j(q_i . k_j / sqrt(d) + mask[b, j]) . v_j; q [B*Lq, ldq], k / v [B*Lk, ld*], heads side by side in a row. */
int vk_attention(const void *q, int ldq, const void *k, int ldk, const void *v, int ldv, const float *mask,
                 void *out, int ldo, int B, int heads, int Lq, int Lk, int d, vk_dtype dt, void *stream);

/* ---- N4: FPN-side ops (what north_star names; the reference holds only fragments, see oracle/fpn_oracle.py) ---- */

/* RoIAlign over a feature pyramid (torchvision roi_align semantics, `aligned` as detectron2's ROIAlignV2; the level loop of
 * ROIPooler.forward frcnn.py:1214-1222): maps[l] NHWC [N,Hs[l],Ws[l],C], rois [K,5] (batch,x1,y1,x2,y2), roi_levels [K]
 * (NULL when levels == 1), out [K,P,P,C].  PARITY UNPINNED (no RoIAlign in the reference). */
int vk_roi_align(const void *const *maps, const int32_t *Hs, const int32_t *Ws, const float *scales, int levels,
                 int N, int C, const float *rois, const int32_t *roi_levels, int K, int P, int sampling_ratio,
                 int aligned, void *out, vk_dtype dt, void *stream);
/* assign_boxes_to_levels frcnn.py:444-460: floor(canonical_level + log2(sqrt(area)/canonical_box_size + 1e-8)),
 * clamped to [min_level, max_level], minus min_level.  boxes [K, ld] (x1,y1,x2,y2 first). */
int vk_assign_levels(const float *boxes, int ld, int K, int min_level, int max_level, float canonical_box_size,
                     int canonical_level, int32_t *levels_out, void *stream);
/* FPN top-down step: y = lateral + nearest-2x(top) (detectron2 FPN; absent from the reference: unpinned). NHWC. */
int vk_upsample2x_add(const void *lateral, const void *top, void *y, int N, int H, int W, int Ht, int Wt, int C,
                      vk_dtype dt, void *stream);
/* LastLevelMaxPool frcnn.py:835-836: max_pool2d(kernel 1, stride 2) = every second pixel.  y [N,(H-1)/2+1,(W-1)/2+1,C] */
int vk_subsample2(const void *x, void *y, int N, int H, int W, int C, vk_dtype dt, void *stream);
/* the ReLU between LastLevelP6P7's convolutions frcnn.py:852-853 (p6 itself stays un-rectified) */
int vk_relu_copy(const void *x, void *y, long n, vk_dtype dt, void *stream);

/* find_top_rpn_proposals frcnn.py:264-390 over SEVERAL levels (the reference's RPN class itself cannot run them: its
 * AnchorGenerator stacks per-level anchors of unequal length; the function and the per-level pieces can, which is how the
 * golden vectors are made): per level top pre_nms_topk of [N,Hl,Wl,A] logits (row stride ld), decode with that level's cell
 * anchors / stride, clip, size filter; concat level-major; batched NMS (torchvision form: boxes shifted by
 * level * (max coordinate + 1)); first post_nms_topk.  levels * pre_nms_topk <= 8192.  Outputs as vk_rpn_proposals. */
size_t vk_rpn_multilevel_workspace_bytes(int N, int levels, int pre_topk, int post_topk);
int vk_rpn_proposals_multilevel(const float *const *logits, const int32_t *ld_logits, const float *const *deltas,
                                const int32_t *ld_deltas, int levels, int N, const int32_t *Hs, const int32_t *Ws, int A,
                                const float *const *cell_anchors, const int32_t *strides, float offset,
                                const int32_t *image_hw, const float *bbox_weights4_host, float min_size,
                                double nms_thresh, int pre_topk, int post_topk, float *out_boxes, float *out_logits,
                                int32_t *out_counts, int32_t *nonfinite_flag, void *workspace, size_t workspace_bytes,
                                void *stream);

/* NCHW f32 -> NHWC (dt) and back (layout plumbing for tests). */
int vk_nchw_to_nhwc(const float *x, int N, int C, int H, int W, void *y, vk_dtype dt, void *stream);
int vk_nhwc_to_nchw(const void *x, int N, int C, int H, int W, float *y, vk_dtype dt, void *stream);

/* stem: 7x7 s2 p3 conv + BN + ReLU + max-pool  <- BasicStem.forward frcnn.py:872-879.
 * x NCHW f32 [N,3,H,W]; w_packed from vk_pack_stem_weight; y NHWC [N,Hp,Wp,cout]. */
size_t vk_packed_stem_bytes(int cout, vk_dtype dt);
int vk_pack_stem_weight(const float *w_oihw_host, const float *bn_host, int cout, vk_dtype dt,
                        void *w_packed_host, float *bias_packed_host);
int vk_stem(const float *x, int N, int H, int W, const void *w_packed, const float *bias_packed,
            int cout, int caffe_maxpool, void *y, vk_dtype dt, void *workspace, size_t workspace_bytes,
            void *stream);
size_t vk_stem_workspace_bytes(int N, int H, int W, int cout, vk_dtype dt);
void vk_stem_out_hw(int H, int W, int caffe_maxpool, int *Ho, int *Wo);

/* Image pre-processing  <- ResizeShortestEdge + Preprocess, vltk/legacy/processing.py:29-150:
 * per image bilinear resize (align_corners=False) of a float HWC (BGR 0-255) device image to new_hw, (x-mean)/std,
 * pad with pad_value to [N,3,Hmax,Wmax] f32 NCHW.  raw_dev_ptrs_host: host array of N device pointers;
 * raw_hw_host / new_hw_host: host [N,2] (h, w); the resize size rule itself is host logic (vltk_amd/preprocess.py). */
int vk_preprocess(const float *const *raw_dev_ptrs_host, const int32_t *raw_hw_host, const int32_t *new_hw_host,
                  int N, int Hmax, int Wmax, const float *mean3_host, const float *std3_host, float pad_value,
                  float *out_nchw_dev, void *stream);

/* max-pool 3x3 s2 (ceil_mode pad 0, or pad 1)  <- frcnn.py:875-878 */
int vk_maxpool3x3s2(const void *x, int N, int H, int W, int C, int caffe, void *y, vk_dtype dt, void *stream);

/* RPN proposals  <- RPNOutputs.predict_* frcnn.py:748-781, find_top_rpn_proposals :264-390,
 * AnchorGenerator :1463-1510, Box2BoxTransform.apply_deltas :548-584, batched_nms (torchvision).
 *   logits [N,Hf,Wf,A] f32, deltas [N,Hf,Wf,4A] f32 (row stride ld_* elements),
 *   cell_anchors [A,4] f32, image_hw dev [N,2] i32.
 *   out_boxes [N,post,4], out_logits [N,post], out_counts [N] i32, nonfinite_flag [1] i32. */
size_t vk_rpn_workspace_bytes(int N, int HWA, int pre_topk);
int vk_rpn_proposals(const float *logits, int ld_logits, const float *deltas, int ld_deltas,
                     int N, int Hf, int Wf, int A, const float *cell_anchors, int stride, float offset,
                     const int32_t *image_hw, const float *bbox_weights4_host, float min_size,
                     double nms_thresh, int pre_topk, int post_topk,
                     float *out_boxes, float *out_logits, int32_t *out_counts, int32_t *nonfinite_flag,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Greedy NMS (torchvision.ops.nms semantics; frcnn.py:132): boxes [n,4], scores [n];
 * keep_out [n] i64 in score order, count_out [1] i32. */
int vk_nms(const float *boxes, const float *scores, int n, double thresh,
           int64_t *keep_out, int32_t *count_out, void *workspace, size_t workspace_bytes, void *stream);
size_t vk_nms_workspace_bytes(int n);

/* RoIPool (torchvision.ops.RoIPool; frcnn.py:1179,1198): feat [N,H,W,C] NHWC, rois [K,5] f32
 * -> out [K,P,P,C] NHWC. */
int vk_roi_pool(const void *feat, int N, int H, int W, int C, const float *rois, int K,
                float spatial_scale, int P, void *out, vk_dtype dt, void *stream);

/* mean over the P*P positions of each RoI  <- frcnn.py:1401: x [K,S,C] (dt) -> out [K,C] f32 */
int vk_mean_pool(const void *x, int K, int S, int C, float *out, vk_dtype dt, void *stream);

/* Box2BoxTransform.apply_deltas (frcnn.py:548-584): deltas [M,4k], boxes [M,4] -> out [M,4k] */
int vk_box_decode(const float *deltas, const float *boxes, int M, int k, const float *weights4_host,
                  float *out, void *stream);

/* Pieces of the box head for callers that compose one themselves (the FPN detector, vltk_amd/frcnn_fpn.py):
 * vk_make_rois       <- convert_boxes_to_pooler_format frcnn.py:426-441: boxes [N,R,4] -> rois [N*R,5] (batch,x1,y1,x2,y2)
 * vk_softmax_argmax  <- ROIOutputs._predict_objs / _predict_attrs :1252-1260: per row soft-max over the first n_softmax
 *                       logits, max / arg-max of the probabilities over the first n_argmax; raw_argmax_out (optional) =
 *                       arg-max of the raw logits over n_softmax (`scores.max(-1)` incl. background, :1732)
 * vk_concat_embed    <- `cat([roi_features, cls_embedding(max_class)], -1)` :1733-1734: out[k] = [ (dt)features[k][0:F] |
 *                       emb[cls[k]][0:E] ]; E == 0: a plain f32 -> dt conversion of the features
 * vk_chosen_deltas   <- bbox_pred :1730 for the arg-max class only: out[k][j] = bias[r] + <x[k], w_rows[r]>,
 *                       r = 4 * cls[k] + j (class-specific) or j (agnostic); w_rows [4C or 4, F] in dt, unpadded rows */
int vk_make_rois(const float *boxes, int N, int R, float *rois, void *stream);
int vk_softmax_argmax(const float *logits, int ld, int K, int n_softmax, int n_argmax, float *prob_out,
                      int32_t *cls_out, int32_t *raw_argmax_out, void *stream);
int vk_concat_embed(const float *features, int F, const void *emb, int E, const int32_t *cls, int K, void *out,
                    vk_dtype dt, void *stream);
int vk_chosen_deltas(const void *x, int ldx, const void *w_rows, const float *bias, const int32_t *cls,
                     int cls_agnostic, int F, int K, float *out, vk_dtype dt, void *stream);

/* ROIOutputs.inference (frcnn.py:1262-1294) + do_nms (:116-143), per image:
 *   obj_logits [K,C+1] f32 (row stride ld_obj), attr_logits [K,A+1] f32 (ld_attr),
 *   box_deltas: either full [K,4C] (ld_box=4C, chosen_only=0) or only the arg-max class's
 *   4 deltas [K,4] (chosen_only=1), proposals [N,R,4] + counts [N], features [K,F] f32. */
int vk_roi_outputs(const float *obj_logits, int ld_obj, const float *attr_logits, int ld_attr,
                   const float *box_deltas, int ld_box, int chosen_only,
                   const float *proposals, const int32_t *counts, const float *features, int F,
                   int N, int R, int C, int A, const int32_t *image_hw, const float *scales_yx_dev,
                   const float *weights4_host, const vk_roi_params *rp, const vk_outputs *out,
                   int64_t *keep_ids_out, int32_t *nonfinite_flag, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VLTK_HIP_H */
