"""Host mirror of the reference's legacy `Preprocess` (vltk/legacy/processing.py:66-150) over the HIP kernel.

Same call contract: `Preprocess(cfg)(images, img_ids) -> (good_ids, images [N,3,Hmax,Wmax], sizes [N,2] (h,w),
scales_yx [N,2] = raw/size)`; `images` are float HWC tensors (BGR, 0-255) -- decoding files is outside the path.
The size rule (ResizeShortestEdge, :41-60) is host logic; the pixels are produced by `vk_preprocess`.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def resized_hw(h, w, min_size, max_size):
    size = min_size
    scale = size * 1.0 / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh = newh * scale
        neww = neww * scale
    return int(newh + 0.5), int(neww + 0.5)


class Preprocess:
    def __init__(self, cfg, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("vltk_amd.Preprocess needs an AMD GPU (HIP device); there is no CPU fallback")
        L.load()
        self.min_size, self.max_size = cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST
        self.pad_value = float(cfg.PAD_VALUE)
        self.size_divisibility = cfg.SIZE_DIVISIBILITY
        self.mean = (C.c_float * 3)(*cfg.MODEL.PIXEL_MEAN)
        self.std = (C.c_float * 3)(*cfg.MODEL.PIXEL_STD)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())

    def __call__(self, images, img_ids):
        if self.size_divisibility > 0:
            raise NotImplementedError()                       # processing.py:145-146
        if not isinstance(images, (list, tuple)):
            images = [images]
        raws, good_ids = [], []
        for img_id, img in zip(img_ids, images):
            if img is None:
                continue
            t = torch.as_tensor(img).to(self.device, torch.float32).contiguous()
            if t.dim() != 3 or t.shape[2] != 3:
                raise ValueError(f"image {img_id}: expected HWC with 3 channels, got {tuple(t.shape)}")
            raws.append(t)
            good_ids.append(img_id)
        if not raws:
            return [], [], [], []
        N = len(raws)
        raw_hw = np.asarray([[r.shape[0], r.shape[1]] for r in raws], dtype=np.int32)
        new_hw = np.asarray([resized_hw(h, w, self.min_size, self.max_size) for h, w in raw_hw], dtype=np.int32)
        Hmax, Wmax = int(new_hw[:, 0].max()), int(new_hw[:, 1].max())
        out = torch.empty((N, 3, Hmax, Wmax), dtype=torch.float32, device=self.device)
        ptrs = (C.c_void_p * N)(*[r.data_ptr() for r in raws])
        L.call("vk_preprocess", ptrs, raw_hw.ctypes.data_as(C.c_void_p), new_hw.ctypes.data_as(C.c_void_p), N, Hmax, Wmax,
               self.mean, self.std, self.pad_value, out.data_ptr(),
               C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        sizes = torch.from_numpy(new_hw.astype(np.int64))
        scales_yx = torch.true_divide(torch.from_numpy(raw_hw.astype(np.int64)), sizes)
        return good_ids, out, sizes, scales_yx
