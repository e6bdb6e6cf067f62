"""Plumbing for the -m gpu parity tests: every call goes through the C ABI
(vltk_amd._lib); torch only owns the device buffers."""
import ctypes as C

import numpy as np
import torch

from vltk_amd import _lib as L

DEV = "cuda:0"
TDT = {L.VK_F32: torch.float32, L.VK_F16: torch.float16}


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rel_err(a, b, floor=1e-6):
    """max |a-b| / max(|b|.max(), floor): the tolerance metric of the parity tests."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor))


def to_nhwc(x_nchw, dt):
    x = x_nchw.to(DEV, torch.float32).contiguous()
    N, Cc, H, W = x.shape
    y = torch.empty((N, H, W, Cc), dtype=TDT[dt], device=DEV)
    L.call("vk_nchw_to_nhwc", P(x), N, Cc, H, W, P(y), dt, stream())
    return y


def to_nchw(y_nhwc, dt):
    N, H, W, Cc = y_nhwc.shape
    out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=DEV)
    L.call("vk_nhwc_to_nchw", P(y_nhwc), N, Cc, H, W, P(out), dt, stream())
    torch.cuda.synchronize()
    return out.cpu()


def pack_conv(w, bn, bias, dt, groups=1):
    """w [cout,cin/groups,kh,kw] f32 numpy; bn = (gamma,beta,mean,var) or None -> device (packed weights, bias)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    cout, cin, kh, kw = w.shape
    cin *= groups
    nbytes = L.load().vk_packed_weight_bytes(cout, cin, kh, kw, groups, dt)
    wp = np.zeros(nbytes, dtype=np.uint8)
    bp = np.zeros(L.load().vk_packed_cout(cout), dtype=np.float32)
    bnp = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float32) for v in bn])) if bn is not None else None
    bi = np.ascontiguousarray(bias, dtype=np.float32) if bias is not None else None
    L.call("vk_pack_conv_weight", w.ctypes.data_as(C.c_void_p),
           bnp.ctypes.data_as(C.c_void_p) if bnp is not None else None,
           bi.ctypes.data_as(C.c_void_p) if bi is not None else None,
           cout, cin, kh, kw, groups, dt, wp.ctypes.data_as(C.c_void_p), bp.ctypes.data_as(C.c_void_p))
    return torch.from_numpy(wp).to(DEV), torch.from_numpy(bp).to(DEV)


def conv2d(x_nchw, w, bn=None, bias=None, residual_nchw=None, stride=1, pad=0, dil=1, relu=False, dt=L.VK_F16,
           out_dt=None, groups=1):
    out_dt = dt if out_dt is None else out_dt
    cout, cin, kh, kw = w.shape
    cin *= groups
    wd, bd = pack_conv(w, bn, bias, dt, groups)
    x = to_nhwc(x_nchw, dt)
    N, H, W, _ = x.shape
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    ldy = (cout + 7) // 8 * 8
    y = torch.zeros((N, Ho, Wo, ldy), dtype=TDT[out_dt], device=DEV)
    res = None
    if residual_nchw is not None:
        assert ldy == cout
        res = to_nhwc(residual_nchw, dt)
    L.call("vk_conv2d", P(x), N, H, W, cin, P(wd), P(bd), P(res), P(y), cout, ldy, kh, kw, stride, pad, dil,
           groups, int(relu), dt, out_dt, stream())
    return to_nchw(y, out_dt)[:, :cout]


def fold_ref(w, bn, dt):
    """CPU reference of the packing: BN folded in float64 -> f32 (-> f16 round trip for the fast mode)."""
    w = torch.as_tensor(w, dtype=torch.float64)
    if bn is not None:
        g, b, m, v = (torch.as_tensor(t, dtype=torch.float64) for t in bn)
        s = g / torch.sqrt(v + 1e-5)
        w = w * s.view(-1, 1, 1, 1)
        bias = (b - m * s).float()
    else:
        bias = None
    w = w.float()
    if dt == L.VK_F16:
        w = w.half().float()
    return w, bias
