"""Host-side mirror of the reference's `FRCNN` module over the HIP C ABI.

Same call surface as `vltk.modeling.frcnn.FRCNN` (reference
vltk/modeling/frcnn.py:1743-2004): `FRCNN(cfg)`, `FRCNN.from_pretrained(path,
config=...)` (local paths only), `model(images, image_shapes, scales_yx=...,
**kwargs)` returning the 7-key OrderedDict, and the mutable
`model.roi_outputs.{nms_thresh, score_thresh, min_detections, max_detections}`
attributes callers set (tests/frcnn_test.py:16-19).

All arithmetic runs in libvltk_hip.so (hand-written gfx950 kernels); torch is
used only to own device memory and to hand out result tensors.  There is no
CPU path: constructing the model without the library or without a GPU raises.
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib as L
from .config import CONFIG_NAME, WEIGHTS_NAME, Config
from .parallel import OutputBlock, output_spec


class ROIOutputs:
    """The mutable knobs of the reference's ROIOutputs (frcnn.py:1229-1240)."""

    def __init__(self, cfg):
        self.score_thresh = cfg.ROI_HEADS.SCORE_THRESH_TEST     # accepted and unused, as upstream (frcnn.py:116)
        self.min_detections = cfg.MIN_DETECTIONS
        self.max_detections = cfg.MAX_DETECTIONS
        nms_thresh = cfg.ROI_HEADS.NMS_THRESH_TEST
        self.nms_thresh = list(nms_thresh) if isinstance(nms_thresh, (list, tuple)) else [nms_thresh]


def _c_config(cfg, precision):
    c = L.vk_config()
    r = cfg.RESNETS
    c.depth, c.num_groups, c.width_per_group = r.DEPTH, r.NUM_GROUPS, r.WIDTH_PER_GROUP
    c.stem_out_channels, c.res2_out_channels = r.STEM_OUT_CHANNELS, r.RES2_OUT_CHANNELS
    c.stride_in_1x1 = int(bool(r.STRIDE_IN_1X1))
    c.caffe_maxpool = int(bool(cfg.MODEL.MAX_POOL))
    sizes, ratios = cfg.ANCHOR_GENERATOR.SIZES, cfg.ANCHOR_GENERATOR.ASPECT_RATIOS
    if len(sizes) != 1 or len(ratios) != 1:
        raise ValueError("only the single-level (C4) anchor generator is supported")
    c.num_sizes, c.num_ratios = len(sizes[0]), len(ratios[0])
    for i, v in enumerate(sizes[0]):
        c.sizes[i] = v
    for i, v in enumerate(ratios[0]):
        c.ratios[i] = v
    c.anchor_offset = cfg.ANCHOR_GENERATOR.OFFSET
    c.rpn_hidden_channels = cfg.PROPOSAL_GENERATOR.HIDDEN_CHANNELS
    c.rpn_min_size = cfg.PROPOSAL_GENERATOR.MIN_SIZE
    c.rpn_nms_thresh = cfg.RPN.NMS_THRESH
    c.pre_nms_topk, c.post_nms_topk = cfg.RPN.PRE_NMS_TOPK_TEST, cfg.RPN.POST_NMS_TOPK_TEST
    for i, v in enumerate(cfg.RPN.BBOX_REG_WEIGHTS):
        c.rpn_bbox_weights[i] = v
    c.num_classes, c.num_attrs = cfg.ROI_HEADS.NUM_CLASSES, cfg.ROI_BOX_HEAD.NUM_ATTRS
    c.use_attr = int(bool(cfg.ROI_BOX_HEAD.ATTR))
    c.pooler_resolution = cfg.ROI_BOX_HEAD.POOLER_RESOLUTION
    c.res5_halve = int(bool(cfg.ROI_BOX_HEAD.RES5HALVE))
    c.cls_agnostic_bbox_reg = int(bool(cfg.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG))
    for i, v in enumerate(cfg.ROI_BOX_HEAD.BBOX_REG_WEIGHTS):
        c.roi_bbox_weights[i] = v
    c.precision = {"fp16": L.VK_F16, "fp32": L.VK_F32}[precision]
    return c


_TORCH_DT = {L.VK_F32: torch.float32, L.VK_F16: torch.float16, L.VK_I64: torch.int64, L.VK_I32: torch.int32}


class _Ticket:
    """State of one forward in flight; the model keeps these in issue order (`FRCNN._open`)."""

    __slots__ = ("ticket", "block", "images", "done", "error")

    def __init__(self, ticket, block, images):
        self.ticket, self.block = ticket, block
        self.images = images           # keeps the input alive until the kernels have read it
        self.done, self.error = False, None


class PendingForward:
    """A forward in flight (FRCNN.forward_async).  Handles are waited for in issue order; a handle that is DROPPED
    (garbage-collected) without a wait closes its own ticket and every older one that is still open, in order, so an
    out-of-order drop cannot leave tickets open (the results of those older forwards stay available to their handles)."""

    def __init__(self, model, state, hw):
        self.model, self._state, self.hw = model, state, hw

    @property
    def ticket(self):
        return self._state.ticket

    @property
    def block(self):
        return self._state.block

    def wait_raw(self):
        """Finish the forward; returns the fixed-capacity OutputBlock ([N, D, ...] device tensors)."""
        st = self._state
        if not st.done:
            self.model._end(st)         # ValueError if an older forward is still open (it stays open)
        if st.error is not None:
            raise st.error              # the forward finished and failed (non-finite boxes, frcnn.py:148)
        self.model._last_padded = st.block
        return st.block

    def wait(self, **kwargs):
        return FRCNN._format(self.wait_raw(), self.hw, **kwargs)

    def __del__(self):
        try:
            if not self._state.done and self.model._h:
                self.model._close_through(self._state)
        except Exception:
            pass


class FRCNN:
    def __new__(cls, cfg=None, *a, **k):
        # several RPN input levels = the FPN detector (frcnn_fpn.py, a build extension); one = the reference's C4 model
        if cls is FRCNN and cfg is not None and len(cfg.RPN.IN_FEATURES) > 1:
            from .frcnn_fpn import FRCNNFPN
            return object.__new__(FRCNNFPN)
        return object.__new__(cls)

    def __init__(self, cfg, precision=None, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("vltk_amd.FRCNN needs an AMD GPU (HIP device); there is no CPU fallback")
        L.load()
        self.config = cfg
        self.min_detections = cfg.min_detections
        self.max_detections = cfg.max_detections
        dev = torch.device(device if device is not None else cfg.MODEL.DEVICE)
        if dev.type != "cuda":
            dev = torch.device("cuda", torch.cuda.current_device())
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.precision = precision or os.environ.get("VLTK_AMD_PRECISION", "fp16")
        self.roi_outputs = ROIOutputs(cfg)
        self.training = False
        self._h = C.c_void_p()
        self._open = []                # _Ticket of every forward in flight, oldest first
        L.call("vk_create", C.byref(_c_config(cfg, self.precision)), dev.index, C.byref(self._h))
        self._finalized = False

    # ---- nn.Module-like surface -----------------------------------------
    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def to(self, *a, **k):
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                L.load().vk_destroy(h)
            except Exception:
                pass
            self._h = None

    def expected_keys(self):
        n = C.c_int()
        L.call("vk_num_weights", self._h, C.byref(n))
        out = []
        for i in range(n.value):
            s = C.c_char_p()
            L.call("vk_weight_name", self._h, i, C.byref(s))
            out.append(s.value.decode())
        return out

    def load_state_dict(self, state_dict, strict=True):
        """Strict load in the reference's key layout (frcnn.py:1862-1881), then fold BN / repack on the device."""
        if self._finalized:
            raise RuntimeError("weights were already loaded into this model")
        for k, v in state_dict.items():
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if a.dtype == np.int64:
                dt = L.VK_I64
            else:
                a = np.ascontiguousarray(a, dtype=np.float32)
                dt = L.VK_F32
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            L.call("vk_load_weights", self._h, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim, dt)
        L.call("vk_finalize", self._h)
        self._finalized = True
        return self

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        """Local branch of the reference loader (frcnn.py:1757-1922): a directory holding
        `pytorch_model.bin` (+ `config.yaml`) or a direct file path.  Fetch-by-name needs the network."""
        config = kwargs.pop("config", None)
        state_dict = kwargs.pop("state_dict", None)
        precision = kwargs.pop("precision", None)
        if not isinstance(config, Config):
            config = Config.from_pretrained(config if config is not None else pretrained_model_name_or_path)
        if state_dict is None:
            path = pretrained_model_name_or_path
            if os.path.isdir(path):
                path = os.path.join(path, WEIGHTS_NAME)
                if not os.path.isfile(path):
                    raise EnvironmentError(
                        "Error no file named {} found in directory {} ".format(WEIGHTS_NAME, pretrained_model_name_or_path))
            elif not os.path.isfile(path):
                raise EnvironmentError(f"Can't load weights for '{pretrained_model_name_or_path}'.")
            try:
                state_dict = torch.load(path, map_location="cpu", weights_only=True)
            except Exception:
                raise OSError("Unable to load weights from pytorch checkpoint file. ")
        model = cls(config, precision=precision)
        model.load_state_dict(state_dict)
        return model.eval()

    def set_option(self, key, value):
        L.call("vk_set_option", self._h, key.encode(), int(value))

    def enable_stage_timing(self, on=True):
        L.call("vk_enable_stage_timing", self._h, int(on))

    def stage_timing_ms(self):
        ms = (C.c_float * 6)()
        L.call("vk_get_stage_timing", self._h, ms)
        return dict(zip(("backbone", "rpn_head", "proposals", "roi_heads", "predictor_outputs", "total"), list(ms)))

    def enable_kernel_timing(self, on=True):
        L.call("vk_enable_kernel_timing", self._h, int(on))

    def kernel_timing(self, reset=False):
        """Per-bucket (launches, ms, algorithmic flops) of the conv launches since the last reset."""
        n = (C.c_int64 * 12)()
        ms = (C.c_double * 12)()
        fl = (C.c_double * 12)()
        by = (C.c_double * 12)()
        L.call("vk_get_kernel_timing", self._h, n, ms, fl, by, int(reset))
        names = ("conv_mfma256", "conv_mfma_f16", "conv_mfma_f16_f32out", "other", "conv3x3_panel", "conv_duo", "two_stream_backbone", "conv3x3_blk", "conv_ws",
                 "conv_mfma256_dual", "conv_gemm4", "bneck64")
        return {k: {"launches": int(n[i]), "ms": float(ms[i]), "flops": float(fl[i]), "bytes": float(by[i])}
                for i, k in enumerate(names)}

    def get_stage(self, name):
        """Intermediate tensor of the last forward as a torch tensor (a copy)."""
        ptr, dt, nd = C.c_void_p(), C.c_int(), C.c_int()
        shape = (C.c_int64 * 4)()
        L.call("vk_get_stage", self._h, name.encode(), C.byref(ptr), C.byref(dt), shape, C.byref(nd))
        shp = [int(shape[i]) for i in range(nd.value)]
        tdt = _TORCH_DT[dt.value]
        out = torch.empty(shp, dtype=tdt, device=self.device)
        nbytes = out.numel() * out.element_size()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        L.call("vk_memcpy_d2d", out.data_ptr(), ptr.value, nbytes, C.c_void_p(stream))
        torch.cuda.synchronize(self.device)
        return out

    # ---- forward ----------------------------------------------------------
    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def forward(self, images, image_shapes, gt_boxes=None, proposals=None, scales_yx=None, ignorey=None, **kwargs):
        """kwargs (v1.0.0 semantics): max_detections, return_tensors {"np","pt",None},
        padding {None,"max_detections","max_batch"}, pad_value, location {"cuda","cpu"}."""
        return self.forward_async(images, image_shapes, gt_boxes, proposals, scales_yx, ignorey).wait(**kwargs)

    def forward_async(self, images, image_shapes, gt_boxes=None, proposals=None, scales_yx=None, ignorey=None):
        """Enqueue a forward and return at once (vk_forward_begin); `.wait(**kwargs)` on the returned handle finishes it
        (vk_forward_end) and formats the outputs like forward().  Up to four may be in flight; they must be waited for
        in order, on the same stream.  The caller must not modify `images` before wait() returns."""
        if self.training:
            raise NotImplementedError()            # frcnn.py:1930-1931
        if proposals is not None:
            raise NotImplementedError("precomputed proposals: the reference path is broken (frcnn.py:1957-1963)")
        if ignorey is not None:
            raise NotImplementedError("ignorey is not supported")
        if not self._finalized:
            raise RuntimeError("no weights loaded: call load_state_dict / from_pretrained first")
        images = torch.as_tensor(images)
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f"images must be [N,3,H,W], got {tuple(images.shape)}")
        images = images.to(device=self.device, dtype=torch.float32).contiguous()
        N, _, H, W = images.shape
        hw = np.ascontiguousarray(np.asarray(torch.as_tensor(image_shapes).cpu()).reshape(N, 2), dtype=np.int32)
        sc = None
        if scales_yx is not None:
            sc = np.ascontiguousarray(np.asarray(torch.as_tensor(scales_yx).cpu(), dtype=np.float32).reshape(N, 2))
        ro = self.roi_outputs
        D = int(ro.max_detections)
        rp = L.vk_roi_params()
        thr = list(ro.nms_thresh)
        rp.num_nms_thresh = len(thr)
        if len(thr) > L.VK_MAX_NMS_THRESH:
            raise ValueError(f"at most {L.VK_MAX_NMS_THRESH} NMS thresholds")
        for i, t in enumerate(thr):
            rp.nms_thresh[i] = float(t)
        rp.min_detections, rp.max_detections = int(ro.min_detections), D
        F = self.config.RESNETS.RES2_OUT_CHANNELS * 8
        dev = self.device
        # one flat block, the seven arrays are views (so the multi-GPU exchange is a single all-gather: parallel.py)
        bufs = OutputBlock(output_spec(N, D, F), device=dev)
        out = L.vk_outputs(*[bufs[k].data_ptr() for k in bufs])
        stream = torch.cuda.current_stream(dev).cuda_stream
        ticket = C.c_int64(-1)
        L.call("vk_forward_begin", self._h, images.data_ptr(), N, H, W, hw.ctypes.data_as(C.c_void_p),
               sc.ctypes.data_as(C.c_void_p) if sc is not None else None, C.byref(rp), C.byref(out),
               C.c_void_p(stream), C.byref(ticket))
        st = _Ticket(ticket.value, bufs, images)
        self._open.append(st)
        return PendingForward(self, st, hw)

    def _end(self, st):
        """vk_forward_end for one ticket.  Out of order -> ValueError from the library, nothing changes."""
        try:
            L.call("vk_forward_end", self._h, C.c_int64(st.ticket))
        except ValueError:
            raise
        except Exception as e:          # the forward itself ended; its assertion failed
            st.error = e
        st.done, st.images = True, None
        if st in self._open:
            self._open.remove(st)

    def _close_through(self, st):
        """End every open forward up to and including `st`, oldest first."""
        while self._open:
            head = self._open[0]
            self._end(head)
            if head is st:
                break

    inference = forward

    def forward_padded(self):
        """Fixed-capacity [N, D, ...] device tensors of the last forward (rows >= preds_per_image are zero)."""
        return self._last_padded

    @staticmethod
    def _format(bufs, hw, padding=None, max_detections=None, return_tensors=None, pad_value=0, location=None, **_):
        assert return_tensors in {"pt", "np", None}
        assert padding in {"max_detections", "max_batch", None}
        ppi = bufs["preds_per_image"].cpu()
        counts = ppi.tolist()
        N = len(counts)
        keys = ("obj_ids", "obj_probs", "attr_ids", "attr_probs", "boxes", "roi_features")
        if padding is None and return_tensors is None:
            # live-tree behaviour (frcnn.py:1996-2004): lists of per-image tensors
            out = OrderedDict((k, [bufs[k][i, :counts[i]] for i in range(N)]) for k in keys[:5])
            out["preds_per_image"] = ppi
            out["roi_features"] = [bufs["roi_features"][i, :counts[i]] for i in range(N)]
            return OrderedDict((k, out[k]) for k in
                               ("obj_ids", "obj_probs", "attr_ids", "attr_probs", "boxes", "preds_per_image", "roi_features"))
        D = bufs["obj_ids"].shape[1]
        if padding == "max_detections":
            width = int(max_detections) if max_detections is not None else D
        elif padding == "max_batch":
            width = max(counts) if counts else 0
        else:
            width = None

        def fix(t):
            if width is None:
                if len(set(counts)) > 1:
                    raise ValueError("return_tensors without padding needs equal detections per image")
                t = t[:, :counts[0]] if counts else t
            elif width <= t.shape[1]:
                t = t[:, :width].clone()
            else:
                pad = torch.zeros((t.shape[0], width - t.shape[1]) + tuple(t.shape[2:]), dtype=t.dtype, device=t.device)
                t = torch.cat([t, pad], dim=1)
            if pad_value != 0 and width is not None:
                for i, c in enumerate(counts):
                    t[i, c:] = pad_value
            if location == "cpu" or return_tensors == "np":
                t = t.cpu()
            return t.numpy() if return_tensors == "np" else t

        out = OrderedDict((k, fix(bufs[k])) for k in keys)
        sizes = torch.as_tensor(hw.astype(np.int64))
        out["preds_per_image"] = ppi.numpy() if return_tensors == "np" else ppi
        out["sizes"] = sizes.numpy() if return_tensors == "np" else sizes
        nb = out["boxes"].copy() if return_tensors == "np" else out["boxes"].clone()
        hwf = hw.astype(np.float32)
        if return_tensors == "np":
            nb[:, :, 0::2] /= hwf[:, 1].reshape(-1, 1, 1)
            nb[:, :, 1::2] /= hwf[:, 0].reshape(-1, 1, 1)
        else:
            s = torch.as_tensor(hwf, device=nb.device)
            nb[:, :, 0::2] /= s[:, 1].view(-1, 1, 1)
            nb[:, :, 1::2] /= s[:, 0].view(-1, 1, 1)
        out["normalized_boxes"] = nb
        return OrderedDict((k, out[k]) for k in ("obj_ids", "obj_probs", "attr_ids", "attr_probs", "boxes", "sizes",
                                                 "preds_per_image", "roi_features", "normalized_boxes"))
