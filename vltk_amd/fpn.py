"""N4: host-side mirrors of the FPN-side pieces, all arithmetic in libvltk_hip.so.

* `FPNNeck`            detectron2-style FPN (lateral 1x1 + nearest-2x top-down + 3x3 output convs, P6 by LastLevelMaxPool
                       frcnn.py:825-836) over NHWC maps [C2..C5] -- the reference has no neck class: parity unpinned.
* `LastLevelP6P7`      frcnn.py:839-854.
* `MultiLevelRoIAlign` ROIPooler.forward's level loop (frcnn.py:1200-1224) with RoIAlign instead of RoIPool and the level rule
                       of assign_boxes_to_levels (frcnn.py:444-460).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

_DT = {"fp32": (L.VK_F32, torch.float32), "fp16": (L.VK_F16, torch.float16), "bf16": (L.VK_BF16, torch.bfloat16)}


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _Conv:
    """conv + bias on the HIP path; weights [cout, cin, k, k] f32."""

    def __init__(self, w, b, dt, dev, stride=1):
        w = np.ascontiguousarray(w, np.float32)
        self.cout, self.cin, self.k, _ = w.shape
        self.dt, self.dev, self.stride = dt, dev, stride
        lib = L.load()
        nb = lib.vk_packed_weight_bytes(self.cout, self.cin, self.k, self.k, 1, dt)
        wp = np.zeros(nb, np.uint8)
        bp = np.zeros(lib.vk_packed_cout(self.cout), np.float32)
        bb = np.ascontiguousarray(b, np.float32)
        L.call("vk_pack_conv_weight", w.ctypes.data_as(C.c_void_p), None, bb.ctypes.data_as(C.c_void_p), self.cout, self.cin, self.k, self.k,
               1, dt, wp.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p))
        self.w, self.b = torch.from_numpy(wp).to(dev), torch.from_numpy(bp).to(dev)

    def __call__(self, x):                      # x NHWC
        N, H, W, _ = x.shape
        pad = self.k // 2
        Ho, Wo = (H + 2 * pad - self.k) // self.stride + 1, (W + 2 * pad - self.k) // self.stride + 1
        y = torch.empty((N, Ho, Wo, self.cout), dtype=x.dtype, device=self.dev)
        L.call("vk_conv2d", x.data_ptr(), N, H, W, self.cin, self.w.data_ptr(), self.b.data_ptr(), None, y.data_ptr(), self.cout, self.cout,
               self.k, self.k, self.stride, pad, 1, 1, 0, self.dt, self.dt, _stream(self.dev))
        return y


class FPNNeck:
    def __init__(self, lateral, output, precision="fp16", device="cuda:0"):
        """lateral[i] / output[i] = (weight, bias) of level i's 1x1 / 3x3 conv, fine -> coarse (C2..C5)."""
        if not torch.cuda.is_available():
            raise RuntimeError("vltk_amd.fpn needs a GPU: there is no CPU fallback")
        self.dt, self.tdt = _DT[precision]
        self.dev = torch.device(device)
        self.lat = [_Conv(w, b, self.dt, self.dev) for w, b in lateral]
        self.out = [_Conv(w, b, self.dt, self.dev) for w, b in output]

    def __call__(self, feats):
        """feats: NHWC device tensors [C2, ..., C5] -> [P2, ..., P5, P6]."""
        prev = self.lat[-1](feats[-1])
        res = [self.out[-1](prev)]
        for i in range(len(feats) - 2, -1, -1):
            lat = self.lat[i](feats[i])
            N, H, W, Cc = lat.shape
            y = torch.empty_like(lat)
            L.call("vk_upsample2x_add", lat.data_ptr(), prev.data_ptr(), y.data_ptr(), N, H, W, prev.shape[1], prev.shape[2], Cc, self.dt,
                   _stream(self.dev))
            prev = y
            res.insert(0, self.out[i](prev))
        p5 = res[-1]
        N, H, W, Cc = p5.shape
        p6 = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), dtype=p5.dtype, device=self.dev)
        L.call("vk_subsample2", p5.data_ptr(), p6.data_ptr(), N, H, W, Cc, self.dt, _stream(self.dev))
        return res + [p6]


class LastLevelP6P7:
    def __init__(self, w6, b6, w7, b7, precision="fp16", device="cuda:0"):
        self.dt, self.tdt = _DT[precision]
        self.dev = torch.device(device)
        self.p6 = _Conv(w6, b6, self.dt, self.dev, stride=2)
        self.p7 = _Conv(w7, b7, self.dt, self.dev, stride=2)

    def __call__(self, c5):
        p6 = self.p6(c5)
        r = torch.empty_like(p6)
        L.call("vk_relu_copy", p6.data_ptr(), r.data_ptr(), p6.numel(), self.dt, _stream(self.dev))
        return p6, self.p7(r)


class MultiLevelRoIAlign:
    def __init__(self, output_size, scales, sampling_ratio=0, aligned=True, canonical_box_size=224, canonical_level=4, precision="fp16",
                 device="cuda:0"):
        self.P, self.scales = int(output_size), [float(s) for s in scales]
        self.sr, self.aligned = int(sampling_ratio), bool(aligned)
        self.min_level, self.max_level = int(round(-np.log2(scales[0]))), int(round(-np.log2(scales[-1])))
        assert len(scales) == self.max_level - self.min_level + 1, "not a pyramid"          # frcnn.py:1168
        self.cbs, self.cl = float(canonical_box_size), int(canonical_level)
        self.dt, self.tdt = _DT[precision]
        self.dev = torch.device(device)

    def __call__(self, feats, rois):
        """feats: NHWC maps fine -> coarse; rois [K,5] f32 (batch, x1, y1, x2, y2) -> ([K,P,P,C], levels [K])."""
        rois = rois.to(self.dev, torch.float32).contiguous()
        K, nl = rois.shape[0], len(feats)
        lv = torch.zeros(K, dtype=torch.int32, device=self.dev)
        s = _stream(self.dev)
        if nl > 1:
            L.call("vk_assign_levels", rois.data_ptr() + 4, 5, K, self.min_level, self.max_level, self.cbs, self.cl, lv.data_ptr(), s)
        N, _, _, Cc = feats[0].shape
        maps = (C.c_void_p * nl)(*[f.data_ptr() for f in feats])
        Hs = (C.c_int32 * nl)(*[f.shape[1] for f in feats])
        Ws = (C.c_int32 * nl)(*[f.shape[2] for f in feats])
        sc = (C.c_float * nl)(*self.scales)
        out = torch.empty((K, self.P, self.P, Cc), dtype=self.tdt, device=self.dev)
        L.call("vk_roi_align", maps, Hs, Ws, sc, nl, N, Cc, rois.data_ptr(), lv.data_ptr() if nl > 1 else None, K, self.P, self.sr,
               int(self.aligned), out.data_ptr(), self.dt, s)
        return out, lv
