"""The FPN detector end to end: ResNet bottom-up (res2..res5) -> FPN neck -> one RPN head over P2..P6 -> multi-level
proposals -> RoIAlign 7x7 by level -> 2-FC box head -> the reference's box predictor / ROIOutputs.

What BASELINE.json's configs and north_star literally name ("ResNet-101-FPN", "RoIAlign").  The reference itself has
no FPN model: only `LastLevelMaxPool` :825-836, `assign_boxes_to_levels` :444-460, the level loop of
`ROIPooler.forward` :1200-1224 and `find_top_rpn_proposals` :264-390 exist, plus the level-agnostic `BottleneckBlock`,
`RPNHead`, `FastRCNNOutputLayers` (sized from `input_size`, :1699-1719) and `ROIOutputs`.  Those pieces are used with the
reference's semantics; the composition is detectron2's standard ResNet-FPN Faster R-CNN and is a build extension --
PARITY UNPINNED vs the reference end to end (oracle: oracle/fpn_oracle.py FPNDetectorOracle).

`vltk_amd.FRCNN(cfg)` returns this class when `cfg.RPN.IN_FEATURES` names several levels (`vltk_amd.config.fpn_config`).
Same call surface and outputs as the C4 model (roi_features are `ROI_BOX_HEAD.FC_DIM` wide).  The composition is host
logic (Python, as in the reference); every arithmetic step is a C-ABI call into libvltk_hip.so -- no CPU path.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L
from .config import CONFIG_NAME, WEIGHTS_NAME, Config          # noqa: F401
from .frcnn import FRCNN, ROIOutputs, _TORCH_DT
from .parallel import OutputBlock, output_spec
from .weights import BLOCKS_PER_STAGE, fpn_layer_spec

_DT = {"fp32": (L.VK_F32, torch.float32), "fp16": (L.VK_F16, torch.float16)}


def _np(a):
    return np.ascontiguousarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a, dtype=np.float32)


def _pack(w, bn, bias, dt, groups=1):
    """-> (packed weight bytes, f32 bias [packed cout]) as numpy, BN folded by the library (vk_pack_conv_weight)."""
    w = _np(w)
    if w.ndim == 2:
        w = w.reshape(w.shape[0], w.shape[1], 1, 1)
    cout, cin, kh, kw = w.shape
    cin *= groups
    lib = L.load()
    wp = np.zeros(lib.vk_packed_weight_bytes(cout, cin, kh, kw, groups, dt), np.uint8)
    bp = np.zeros(lib.vk_packed_cout(cout), np.float32)
    bnp = np.ascontiguousarray(np.concatenate([_np(v).reshape(-1) for v in bn])) if bn is not None else None
    bi = _np(bias) if bias is not None else None
    L.call("vk_pack_conv_weight", w.ctypes.data_as(C.c_void_p), bnp.ctypes.data_as(C.c_void_p) if bnp is not None else None,
           bi.ctypes.data_as(C.c_void_p) if bi is not None else None, cout, cin, kh, kw, groups, dt,
           wp.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p))
    return wp, bp


class _Layer:
    """conv (+ folded BN / bias) (+ residual) (+ ReLU) on NHWC device tensors: Conv2d.forward frcnn.py:794-822."""

    def __init__(self, model, w, bn=None, bias=None, stride=1, pad=0, dil=1, groups=1):
        w = _np(w)
        if w.ndim == 2:
            w = w.reshape(w.shape[0], w.shape[1], 1, 1)
        self.m = model
        self.cout, self.cin, self.k = w.shape[0], w.shape[1] * groups, w.shape[2]
        self.stride, self.pad, self.dil, self.groups = stride, pad, dil, groups
        self.host = _pack(w, bn, bias, model.dt, groups)
        self.w = torch.from_numpy(self.host[0]).to(model.device)
        self.b = torch.from_numpy(self.host[1]).to(model.device)

    def out_hw(self, H, W):
        e = self.dil * (self.k - 1) + 1
        return (H + 2 * self.pad - e) // self.stride + 1, (W + 2 * self.pad - e) // self.stride + 1

    def __call__(self, x, relu=False, residual=None, out_f32=False):
        m = self.m
        N, H, W, cin = x.shape
        assert cin == self.cin, (cin, self.cin)
        Ho, Wo = self.out_hw(H, W)
        ldy = (self.cout + 7) // 8 * 8
        y = torch.empty((N, Ho, Wo, ldy), dtype=torch.float32 if out_f32 else m.tdt, device=m.device)
        L.call("vk_conv2d", x.data_ptr(), N, H, W, cin, self.w.data_ptr(), self.b.data_ptr(),
               residual.data_ptr() if residual is not None else None, y.data_ptr(), self.cout, ldy, self.k, self.k,
               self.stride, self.pad, self.dil, self.groups, int(relu), m.dt, L.VK_F32 if out_f32 else m.dt, m._stream())
        return y


class _Linear:
    """nn.Linear (+ ReLU) through the MFMA GEMMs (vk_linear); `fp32`: exact-f32 MFMA whatever the model's mode."""

    def __init__(self, model, w, bias, fp32=False):
        self.m = model
        w = _np(w)
        self.nout, self.k = w.shape
        self.dt, self.tdt = (L.VK_F32, torch.float32) if fp32 else (model.dt, model.tdt)
        wp, bp = _pack(w, None, bias, self.dt)
        self.w, self.b = torch.from_numpy(wp).to(model.device), torch.from_numpy(bp).to(model.device)

    def __call__(self, x, relu=False, out_f32=False):
        m = self.m
        M = x.shape[0]
        assert x.shape[1] == self.k and x.is_contiguous() and x.dtype == self.tdt
        ldy = (self.nout + 7) // 8 * 8
        y = torch.empty((M, ldy), dtype=torch.float32 if out_f32 else self.tdt, device=m.device)
        if M:
            L.call("vk_linear", x.data_ptr(), M, self.k, self.w.data_ptr(), self.b.data_ptr(), None, y.data_ptr(), self.nout, ldy,
                   L.VK_ACT_RELU if relu else L.VK_ACT_NONE, self.dt, L.VK_F32 if out_f32 else self.dt, m._stream())
        return y


class _Bottleneck:
    """BottleneckBlock frcnn.py:903-979.  A stride-1 projection shortcut is part of conv3's GEMM in the fast mode
    (`out = conv3(t); out += shortcut(x)` as one dual-source GEMM, vk_conv1x1_dual), as in the C4 model."""

    def __init__(self, model, sd, p, stride, groups, stride_in_1x1, dil=1):
        def bn(q):
            return [sd[f"{q}.norm.weight"], sd[f"{q}.norm.bias"], sd[f"{q}.norm.running_mean"], sd[f"{q}.norm.running_var"]]
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)             # frcnn.py:932
        self.m = model
        self.conv1 = _Layer(model, sd[p + ".conv1.weight"], bn(p + ".conv1"), stride=s1)
        self.conv2 = _Layer(model, sd[p + ".conv2.weight"], bn(p + ".conv2"), stride=s3, pad=dil, dil=dil, groups=groups)
        self.conv3 = _Layer(model, sd[p + ".conv3.weight"], bn(p + ".conv3"))
        self.shortcut, self.fused = None, None
        if (p + ".shortcut.weight") in sd:
            self.shortcut = _Layer(model, sd[p + ".shortcut.weight"], bn(p + ".shortcut"), stride=stride)
            c3, sc = self.conv3, self.shortcut
            if model.dt == L.VK_F16 and stride == 1 and c3.cout % 256 == 0 and c3.cin % 64 == 0 and sc.cin % 64 == 0:
                rows = c3.host[1].shape[0]
                cat = np.concatenate([c3.host[0].reshape(rows, -1), sc.host[0].reshape(rows, -1)], axis=1)
                self.fused = (torch.from_numpy(np.ascontiguousarray(cat).reshape(-1)).to(model.device),
                              torch.from_numpy(c3.host[1] + sc.host[1]).to(model.device))

    def _whole_block_kernel(self, x):
        """res2's blocks (64 bottleneck channels, 256 out, stride 1, f16) run as ONE kernel: csrc/bneck_fused.hip, as in the C4
        model (bit-identical to the layer-by-layer kernels; VK_BNECK_FUSED=0 switches it off)."""
        m, c1, c2, c3 = self.m, self.conv1, self.conv2, self.conv3
        if m.dt != L.VK_F16 or os.environ.get("VK_BNECK_FUSED") == "0":
            return False
        if not (c1.cout == 64 and c3.cout == 256 and c1.stride == 1 and c2.stride == 1 and c2.dil == 1 and c2.groups == 1):
            return False
        proj = self.shortcut is not None
        if (proj and (self.fused is None or c1.cin != 64)) or (not proj and c1.cin != 256):
            return False
        N, H, W, _ = x.shape
        return N * H * W * 512 < (1 << 31)

    def __call__(self, x):
        m = self.m
        if self._whole_block_kernel(x):
            N, H, W, cin = x.shape
            proj = self.shortcut is not None
            w3, b3 = (self.fused if proj else (self.conv3.w, self.conv3.b))
            y = torch.empty((N, H, W, 256), dtype=m.tdt, device=m.device)
            L.call("vk_bottleneck64", x.data_ptr(), N, H, W, cin, int(proj), self.conv1.w.data_ptr(), self.conv1.b.data_ptr(),
                   self.conv2.w.data_ptr(), self.conv2.b.data_ptr(), w3.data_ptr(), b3.data_ptr(), y.data_ptr(), m._stream())
            return y
        t = self.conv2(self.conv1(x, relu=True), relu=True)
        if self.fused is not None:
            N, H, W, c1 = t.shape
            y = torch.empty((N, H, W, self.conv3.cout), dtype=m.tdt, device=m.device)
            L.call("vk_conv1x1_dual", t.data_ptr(), c1, x.data_ptr(), x.shape[3], N * H * W, self.fused[0].data_ptr(),
                   self.fused[1].data_ptr(), None, y.data_ptr(), self.conv3.cout, 1, m._stream())
            return y
        res = self.shortcut(x) if self.shortcut is not None else x
        return self.conv3(t, relu=True, residual=res)


class _Done:
    """A forward that has already finished (this model runs its forward synchronously)."""

    def __init__(self, model, block, hw):
        self.model, self.block, self.hw = model, block, hw

    def wait_raw(self):
        return self.block

    def wait(self, **kwargs):
        return FRCNN._format(self.block, self.hw, **kwargs)


class FRCNNFPN(FRCNN):
    STAGES = ("res2", "res3", "res4", "res5")

    def __init__(self, cfg, precision=None, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("vltk_amd.FRCNN needs an AMD GPU (HIP device); there is no CPU fallback")
        L.load()
        self.config = cfg
        self.min_detections, self.max_detections = cfg.min_detections, cfg.max_detections
        dev = torch.device(device if device is not None else cfg.MODEL.DEVICE)
        if dev.type != "cuda" or dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.precision = precision or os.environ.get("VLTK_AMD_PRECISION", "fp16")
        self.dt, self.tdt = _DT[self.precision]
        self.roi_outputs = ROIOutputs(cfg)
        self.training = False
        self._h = None
        self._open = []
        self._finalized = False
        self._stages = {}
        self._timing = None
        self.visual_dim = int(cfg.ROI_BOX_HEAD.FC_DIM)
        levels = len(cfg.RPN.IN_FEATURES)
        if levels * int(cfg.RPN.PRE_NMS_TOPK_TEST) > 8192 or int(cfg.RPN.POST_NMS_TOPK_TEST) > 1024:
            raise ValueError("levels * PRE_NMS_TOPK_TEST must be <= 8192 and POST_NMS_TOPK_TEST <= 1024")
        if len(cfg.ROI_HEADS.IN_FEATURES) != 4 or levels not in (4, 5):
            raise ValueError("the FPN detector pools from p2..p5 and proposes from p2..p5(+p6)")

    def __del__(self):
        pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def expected_keys(self):
        keys = []
        for prefix, shape, kind in fpn_layer_spec(self.config):
            keys.append(prefix + ".weight")
            if kind.startswith("conv_bn"):
                keys += [prefix + ".norm." + s for s in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
            elif kind != "embedding":
                keys.append(prefix + ".bias")
        keys += [f"proposal_generator.anchor_generator.cell_anchors.{i}" for i in range(len(self.config.ANCHOR_GENERATOR.SIZES))]
        return keys

    def load_state_dict(self, state_dict, strict=True):
        """Strict load (frcnn.py:1862-1881) in detectron2's FPN key layout (weights.fpn_layer_spec), gamma/beta renamed."""
        if self._finalized:
            raise RuntimeError("weights were already loaded into this model")
        sd = {}
        for k, v in state_dict.items():
            k = k.replace("norm.gamma", "norm.weight").replace("norm.beta", "norm.bias")
            sd[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        want = self.expected_keys()
        missing = [k for k in want if k not in sd and not k.endswith("num_batches_tracked")]
        extra = [k for k in sd if k not in set(want)]
        if missing:
            raise OSError(f"missing key(s) in state_dict: {missing[:5]}{' ...' if len(missing) > 5 else ''}")
        if extra:
            raise OSError(f"unexpected key(s) in state_dict: {extra[:5]}{' ...' if len(extra) > 5 else ''}")
        cfg, r = self.config, self.config.RESNETS
        # stem (BasicStem frcnn.py:857-888)
        p = "backbone.bottom_up.stem.conv1"
        w = _np(sd[p + ".weight"])
        self.stem_c = w.shape[0]
        lib = L.load()
        wp = np.zeros(lib.vk_packed_stem_bytes(self.stem_c, self.dt), np.uint8)
        bp = np.zeros(lib.vk_packed_cout(self.stem_c), np.float32)
        bn = np.ascontiguousarray(np.concatenate([_np(sd[f"{p}.norm.{s}"]) for s in ("weight", "bias", "running_mean", "running_var")]))
        L.call("vk_pack_stem_weight", w.ctypes.data_as(C.c_void_p), bn.ctypes.data_as(C.c_void_p), self.stem_c, self.dt,
               wp.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p))
        self.stem_w, self.stem_b = torch.from_numpy(wp).to(self.device), torch.from_numpy(bp).to(self.device)
        # bottom-up stages (build_backbone frcnn.py:200-261 with res5 as a backbone stage)
        self.stages = []
        for si, name in enumerate(self.STAGES):
            blocks = []
            for b in range(BLOCKS_PER_STAGE[r.DEPTH][si]):
                stride = (1 if si == 0 else 2) if b == 0 else 1
                blocks.append(_Bottleneck(self, sd, f"backbone.bottom_up.{name}.{b}", stride, r.NUM_GROUPS, bool(r.STRIDE_IN_1X1)))
            self.stages.append(blocks)
        # neck
        self.lateral = [_Layer(self, sd[f"backbone.fpn_lateral{l}.weight"], None, sd[f"backbone.fpn_lateral{l}.bias"]) for l in (2, 3, 4, 5)]
        self.output = [_Layer(self, sd[f"backbone.fpn_output{l}.weight"], None, sd[f"backbone.fpn_output{l}.bias"], pad=1) for l in (2, 3, 4, 5)]
        # RPN head (RPNHead frcnn.py:1513-1572): 3x3 + ReLU, then [objectness | deltas] as one 1x1 GEMM with f32 output
        q = "proposal_generator.rpn_head."
        self.rpn_conv = _Layer(self, sd[q + "conv.weight"], None, sd[q + "conv.bias"], pad=1)
        self.A = int(sd[q + "objectness_logits.weight"].shape[0])
        self.rpn_out = _Layer(self, np.concatenate([_np(sd[q + "objectness_logits.weight"]), _np(sd[q + "anchor_deltas.weight"])], 0), None,
                              np.concatenate([_np(sd[q + "objectness_logits.bias"]), _np(sd[q + "anchor_deltas.bias"])], 0))
        self.cells = [torch.from_numpy(_np(sd[f"proposal_generator.anchor_generator.cell_anchors.{i}"])).to(self.device)
                      for i in range(len(cfg.RPN.IN_FEATURES))]
        # box head: fc1 expects detectron2's (c, y, x) flatten; the pooled tensor here is [K, y, x, c]
        P, fc = int(cfg.ROI_BOX_HEAD.POOLER_RESOLUTION), int(cfg.FPN.OUT_CHANNELS)
        self.fcs = []
        for i in range(int(cfg.ROI_BOX_HEAD.NUM_FC)):
            w = _np(sd[f"roi_heads.box_head.fc{i + 1}.weight"])
            if i == 0:
                w = np.ascontiguousarray(w.reshape(w.shape[0], fc, P * P).transpose(0, 2, 1).reshape(w.shape[0], -1))
            self.fcs.append(_Linear(self, w, sd[f"roi_heads.box_head.fc{i + 1}.bias"]))
        # predictor (FastRCNNOutputLayers frcnn.py:1676-1740): fp32 in both modes, as in the C4 model (csrc/model.hip pdt)
        bp_ = "roi_heads.box_predictor."
        self.cls_score = _Linear(self, sd[bp_ + "cls_score.weight"], sd[bp_ + "cls_score.bias"], fp32=True)
        self.bbox_w = torch.from_numpy(_np(sd[bp_ + "bbox_pred.weight"])).to(self.device).contiguous()
        self.bbox_b = torch.from_numpy(_np(sd[bp_ + "bbox_pred.bias"])).to(self.device)
        self.use_attr = bool(cfg.ROI_BOX_HEAD.ATTR)
        if not self.use_attr:
            raise NotImplementedError("ROI_BOX_HEAD.ATTR=false: the reference's Res5ROIHeads unpacks three outputs (frcnn.py:1400)")
        self.emb = torch.from_numpy(_np(sd[bp_ + "cls_embedding.weight"])).to(self.device).contiguous()
        self.fc_attr = _Linear(self, sd[bp_ + "fc_attr.weight"], sd[bp_ + "fc_attr.bias"], fp32=True)
        self.attr_score = _Linear(self, sd[bp_ + "attr_score.weight"], sd[bp_ + "attr_score.bias"], fp32=True)
        self._finalized = True
        return self

    # ---- options / timing (the C4 model keeps these in the library; here they are host-side) ----
    def set_option(self, key, value):
        pass

    def enable_stage_timing(self, on=True):
        self._timing = {} if on else None

    def stage_timing_ms(self):
        t = self._timing or {}
        ev = t.get("ev")
        if not ev:
            return {}
        torch.cuda.synchronize(self.device)
        names = ("backbone", "neck", "rpn_head", "proposals", "box_head", "predictor_outputs")
        out = {n: ev[i].elapsed_time(ev[i + 1]) for i, n in enumerate(names)}
        out["total"] = ev[0].elapsed_time(ev[-1])
        return out

    def enable_kernel_timing(self, on=True):
        pass

    def kernel_timing(self, reset=False):
        return {}

    def get_stage(self, name):
        return self._stages[name]

    def _mark(self, evs):
        if evs is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(self.device))
            evs.append(e)

    # ---- forward ----
    def forward_async(self, images, image_shapes, gt_boxes=None, proposals=None, scales_yx=None, ignorey=None):
        if self.training:
            raise NotImplementedError()
        if proposals is not None or ignorey is not None:
            raise NotImplementedError("precomputed proposals / ignorey are not supported")
        if not self._finalized:
            raise RuntimeError("no weights loaded: call load_state_dict / from_pretrained first")
        images = torch.as_tensor(images)
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f"images must be [N,3,H,W], got {tuple(images.shape)}")
        images = images.to(device=self.device, dtype=torch.float32).contiguous()
        N, _, H, W = images.shape
        hw = np.ascontiguousarray(np.asarray(torch.as_tensor(image_shapes).cpu()).reshape(N, 2), dtype=np.int32)
        if (hw < 1).any():
            raise ValueError("image_shapes must be positive")
        cfg, dev, s = self.config, self.device, self._stream()
        evs = [] if self._timing is not None else None
        st = self._stages = {}
        self._mark(evs)
        # ---- bottom-up ----
        ho, wo = C.c_int(), C.c_int()
        L.load().vk_stem_out_hw(H, W, int(bool(cfg.MODEL.MAX_POOL)), C.byref(ho), C.byref(wo))
        x = torch.empty((N, ho.value, wo.value, self.stem_c), dtype=self.tdt, device=dev)
        nb = L.load().vk_stem_workspace_bytes(N, H, W, self.stem_c, self.dt)
        ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
        L.call("vk_stem", images.data_ptr(), N, H, W, self.stem_w.data_ptr(), self.stem_b.data_ptr(), self.stem_c,
               int(bool(cfg.MODEL.MAX_POOL)), x.data_ptr(), self.dt, ws.data_ptr(), nb, s)
        feats = []
        for name, blocks in zip(self.STAGES, self.stages):
            for blk in blocks:
                x = blk(x)
            feats.append(x)
            st[name] = x
        self._mark(evs)
        # ---- neck (detectron2 FPN; top block LastLevelMaxPool frcnn.py:825-836) ----
        prev = self.lateral[3](feats[3])
        pyr = [self.output[3](prev)]
        for i in (2, 1, 0):
            lat = self.lateral[i](feats[i])
            n_, h_, w_, c_ = lat.shape
            y = torch.empty_like(lat)
            L.call("vk_upsample2x_add", lat.data_ptr(), prev.data_ptr(), y.data_ptr(), n_, h_, w_, prev.shape[1], prev.shape[2], c_, self.dt, s)
            prev = y
            pyr.insert(0, self.output[i](prev))
        if len(cfg.RPN.IN_FEATURES) == 5:
            p5 = pyr[-1]
            n_, h_, w_, c_ = p5.shape
            p6 = torch.empty((n_, (h_ - 1) // 2 + 1, (w_ - 1) // 2 + 1, c_), dtype=self.tdt, device=dev)
            L.call("vk_subsample2", p5.data_ptr(), p6.data_ptr(), n_, h_, w_, c_, self.dt, s)
            pyr.append(p6)
        for i, p_ in enumerate(pyr):
            st[f"p{i + 2}"] = p_
        self._mark(evs)
        # ---- RPN head over every level ----
        A, nl = self.A, len(pyr)
        heads = [self.rpn_out(self.rpn_conv(p_, relu=True), out_f32=True) for p_ in pyr]          # [N,h,w,ld] f32: [0,A) obj, [A,5A) deltas
        for i, h_ in enumerate(heads):
            st[f"rpn_out{i + 2}"] = h_
        self._mark(evs)
        # ---- proposals (find_top_rpn_proposals frcnn.py:264-390 over the levels) ----
        R, pre = int(cfg.RPN.POST_NMS_TOPK_TEST), int(cfg.RPN.PRE_NMS_TOPK_TEST)
        ld = heads[0].shape[3]
        P_ = lambda ptrs: (C.c_void_p * nl)(*ptrs)                                             # noqa: E731
        I_ = lambda vs: (C.c_int32 * nl)(*[int(v) for v in vs])                                # noqa: E731
        hw_dev = torch.from_numpy(hw).to(dev)
        pb = torch.zeros((N, R, 4), dtype=torch.float32, device=dev)
        pl = torch.zeros((N, R), dtype=torch.float32, device=dev)
        pc = torch.zeros(N, dtype=torch.int32, device=dev)
        flag = torch.zeros(2, dtype=torch.int32, device=dev)
        nbw = L.load().vk_rpn_multilevel_workspace_bytes(N, nl, pre, R)
        wsp = torch.empty(nbw, dtype=torch.uint8, device=dev)
        wts = (C.c_float * 4)(*[float(v) for v in cfg.RPN.BBOX_REG_WEIGHTS])
        L.call("vk_rpn_proposals_multilevel", P_([h_.data_ptr() for h_ in heads]), I_([ld] * nl),
               P_([h_.data_ptr() + 4 * A for h_ in heads]), I_([ld] * nl), nl, N, I_([h_.shape[1] for h_ in heads]),
               I_([h_.shape[2] for h_ in heads]), A, P_([c_.data_ptr() for c_ in self.cells]), I_([4 * 2 ** i for i in range(nl)]),
               float(cfg.ANCHOR_GENERATOR.OFFSET), hw_dev.data_ptr(), wts, float(cfg.PROPOSAL_GENERATOR.MIN_SIZE),
               float(cfg.RPN.NMS_THRESH), pre, R, pb.data_ptr(), pl.data_ptr(), pc.data_ptr(), flag.data_ptr(), wsp.data_ptr(), nbw, s)
        st["proposal_boxes"], st["proposal_logits"], st["proposal_counts"] = pb, pl, pc
        self._mark(evs)
        # ---- box head: RoIAlign by level (ROIPooler.forward's level loop :1200-1224, level rule :444-460) + FCs ----
        K = N * R
        rois = torch.empty((K, 5), dtype=torch.float32, device=dev)
        L.call("vk_make_rois", pb.data_ptr(), N, R, rois.data_ptr(), s)
        lv = torch.zeros(K, dtype=torch.int32, device=dev)
        L.call("vk_assign_levels", rois.data_ptr() + 4, 5, K, 2, 5, 224.0, 4, lv.data_ptr(), s)
        P, fc = int(cfg.ROI_BOX_HEAD.POOLER_RESOLUTION), pyr[0].shape[3]
        pooled = torch.empty((K, P, P, fc), dtype=self.tdt, device=dev)
        maps = (C.c_void_p * 4)(*[p_.data_ptr() for p_ in pyr[:4]])
        L.call("vk_roi_align", maps, (C.c_int32 * 4)(*[p_.shape[1] for p_ in pyr[:4]]), (C.c_int32 * 4)(*[p_.shape[2] for p_ in pyr[:4]]),
               (C.c_float * 4)(1 / 4, 1 / 8, 1 / 16, 1 / 32), 4, N, fc, rois.data_ptr(), lv.data_ptr(), K, P,
               int(cfg.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO), 1, pooled.data_ptr(), self.dt, s)
        st["pooled"], st["levels"] = pooled, lv
        x = pooled.view(K, P * P * fc)
        for i, fcl in enumerate(self.fcs):
            x = fcl(x, relu=True, out_f32=(i + 1 == len(self.fcs)))
        feat = x                                                                              # [K, FC_DIM] f32 = roi_features
        st["box_features"] = feat
        self._mark(evs)
        # ---- predictor (FastRCNNOutputLayers.forward :1726-1740) ----
        F_, Cn, At, E = feat.shape[1], int(cfg.ROI_HEADS.NUM_CLASSES), int(cfg.ROI_BOX_HEAD.NUM_ATTRS), self.emb.shape[1]
        cls_logits = self.cls_score(feat, out_f32=True)
        obj_prob = torch.empty(K, dtype=torch.float32, device=dev)
        obj_cls = torch.empty(K, dtype=torch.int32, device=dev)
        max_class = torch.empty(K, dtype=torch.int32, device=dev)
        L.call("vk_softmax_argmax", cls_logits.data_ptr(), cls_logits.shape[1], K, Cn + 1, Cn, obj_prob.data_ptr(), obj_cls.data_ptr(),
               max_class.data_ptr(), s)
        cat = torch.empty((K, F_ + E), dtype=torch.float32, device=dev)
        L.call("vk_concat_embed", feat.data_ptr(), F_, self.emb.data_ptr(), E, max_class.data_ptr(), K, cat.data_ptr(), L.VK_F32, s)
        attr_logits = self.attr_score(self.fc_attr(cat, relu=True), out_f32=True)
        chosen = torch.empty((K, 4), dtype=torch.float32, device=dev)
        L.call("vk_chosen_deltas", feat.data_ptr(), F_, self.bbox_w.data_ptr(), self.bbox_b.data_ptr(), obj_cls.data_ptr(),
               int(bool(cfg.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG)), F_, K, chosen.data_ptr(), L.VK_F32, s)
        st["obj_logits"], st["attr_logits"], st["chosen_deltas"] = cls_logits, attr_logits, chosen
        # ---- outputs (ROIOutputs.inference :1262-1294) ----
        ro = self.roi_outputs
        D = int(ro.max_detections)
        rp = L.vk_roi_params()
        thr = list(ro.nms_thresh)
        if len(thr) > L.VK_MAX_NMS_THRESH:
            raise ValueError(f"at most {L.VK_MAX_NMS_THRESH} NMS thresholds")
        rp.num_nms_thresh = len(thr)
        for i, t in enumerate(thr):
            rp.nms_thresh[i] = float(t)
        rp.min_detections, rp.max_detections = int(ro.min_detections), D
        bufs = OutputBlock(output_spec(N, D, F_), device=dev)
        bufs.flat.zero_()
        out = L.vk_outputs(*[bufs[k].data_ptr() for k in bufs])
        sc_dev = None
        if scales_yx is not None:
            sc_dev = torch.from_numpy(np.ascontiguousarray(np.asarray(torch.as_tensor(scales_yx).cpu(), dtype=np.float32).reshape(N, 2))).to(dev)
        keep = torch.zeros((N, D), dtype=torch.int64, device=dev)
        wts2 = (C.c_float * 4)(*[float(v) for v in cfg.ROI_BOX_HEAD.BBOX_REG_WEIGHTS])
        L.call("vk_roi_outputs", cls_logits.data_ptr(), cls_logits.shape[1], attr_logits.data_ptr(), attr_logits.shape[1],
               chosen.data_ptr(), 4, 1, pb.data_ptr(), pc.data_ptr(), feat.data_ptr(), F_, N, R, Cn, At, hw_dev.data_ptr(),
               sc_dev.data_ptr() if sc_dev is not None else None, wts2, C.byref(rp), C.byref(out), keep.data_ptr(),
               flag.data_ptr() + 4, s)
        st["keep_ids"] = keep
        self._mark(evs)
        if evs is not None:
            self._timing["ev"] = evs
        if int(flag.cpu().sum()) != 0:
            raise AssertionError("Box tensor contains infinite or NaN!")          # frcnn.py:148
        self._last_padded = bufs
        return _Done(self, bufs, hw)
