"""Helpers shared by the -m gpu tests: every call goes through the C ABI of libtreedet_hip.so."""
import numpy as np
import torch

from treedetection_amd import _lib


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def conv2d_hip(x_nchw, w_oihw, scale=None, bias=None, residual_nchw=None, res_shift=0, stride=1, pad=0, relu=False,
               precision=0, tile_cfg=-1, out_f32=False):
    """x [B,C,H,W] np → y [B,Co,Ho,Wo] np through td_conv2d_nhwc (``out_f32``: float16 tensors, float32 output)."""
    lib = _lib.load()
    dt = torch.float32 if precision == 0 else torch.float16
    x = dev(np.transpose(x_nchw, (0, 2, 3, 1)), dt)
    w = dev(np.transpose(w_oihw, (0, 2, 3, 1)), dt)
    B, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    y = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32 if out_f32 else dt, device="cuda")
    sc = dev(scale, torch.float32) if scale is not None else None
    bi = dev(bias, torch.float32) if bias is not None else None
    rs = dev(np.transpose(residual_nchw, (0, 2, 3, 1)), dt) if residual_nchw is not None else None
    p = lambda t: t.data_ptr() if t is not None else None
    st = lib.td_conv2d_nhwc(p(x), p(w), p(sc), p(bi), p(rs), res_shift, p(y), B, H, W, Cin, Cout, KH, KW, stride, pad,
                            int(relu), precision | ((tile_cfg + 1) << 8) | (0x10000 if out_f32 else 0), _lib.stream_ptr())
    _lib.check(st, "td_conv2d_nhwc")
    torch.cuda.synchronize()
    return y.float().cpu().numpy().transpose(0, 3, 1, 2)


def conv2d_head_hip(x_nchw, w_oihw, bias, head_w, head_b, pad, tile_cfg):
    """fp16 conv (+ bias, ReLU, 256 channels) with its 1x1 head fused (td_conv2d_head_nhwc) → head output [B,n,Ho,Wo] float32."""
    lib = _lib.load()
    x = dev(np.transpose(x_nchw, (0, 2, 3, 1)), torch.float16)
    w = dev(np.transpose(w_oihw, (0, 2, 3, 1)), torch.float16)
    B, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    assert Cout == 256 and head_w.shape[1] == 256
    Ho, Wo = H + 2 * pad - KH + 1, W + 2 * pad - KW + 1
    n = head_w.shape[0]
    hw, hb, bi = dev(head_w.reshape(n, 256), torch.float16), dev(head_b, torch.float32), dev(bias, torch.float32)
    y = torch.full((B, Ho, Wo, n), float("nan"), dtype=torch.float32, device="cuda")
    st = lib.td_conv2d_head_nhwc(x.data_ptr(), w.data_ptr(), bi.data_ptr(), hw.data_ptr(), hb.data_ptr(), y.data_ptr(), B, H, W, Cin,
                                 KH, KW, pad, n, 1 | ((tile_cfg + 1) << 8), _lib.stream_ptr())
    _lib.check(st, "td_conv2d_head_nhwc")
    torch.cuda.synchronize()
    return y.cpu().numpy().transpose(0, 3, 1, 2)
