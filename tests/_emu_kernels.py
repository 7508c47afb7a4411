"""TEST-ONLY stand-ins for ``imagine360_amd.kernels`` written with plain torch ops.

The product has no CPU path.  These functions restate the *contract* of each HIP kernel (argument
meaning, layouts, the wrap / x_off / pad addressing modes) so that ``-m "not gpu"`` tests can exercise
the host logic of imagine360_amd (block walk, layouts, caching, RNG order) in fp32 on a machine without
a GPU, by monkeypatching the ``kernels`` module attributes.  The real kernels are checked against the
oracle in tests/test_kernels_gpu.py; nothing outside tests/ imports this file.
"""
import contextlib

import torch
import torch.nn.functional as F


def pack_attn_bias(bias):
    return (bias.float() * 1.4426950408889634).half()


def attn_bias_blocks(packed):
    return torch.ones((-(-packed.shape[0] // 32), -(-packed.shape[1] // 1024)), dtype=torch.int32)


def attention(q, k, v, heads, scale=None, bias=None, out=None, accumulate=False, out_scale=1.0, kv_group=1,
              bias_alt=None, bias_sel=None, bias_packed=False, bias_blocks=None, bias_blocks_alt=None):
    if bias_sel is not None and int(bias_sel) != 0:
        bias = bias_alt
    if bias is not None and bias_packed:
        bias = bias.float() / 1.4426950408889634
    B, Nq, C = q.shape
    d = C // heads
    if kv_group > 1:
        k, v = k.repeat_interleave(kv_group, 0), v.repeat_interleave(kv_group, 0)
    qh = q.reshape(B, Nq, heads, d).transpose(1, 2).float()
    kh = k.reshape(B, -1, heads, d).transpose(1, 2).float()
    vh = v.reshape(B, -1, heads, d).transpose(1, 2).float()
    s = qh @ kh.transpose(-1, -2) * (d ** -0.5 if scale is None else scale)
    if bias is not None:
        s = s + bias.float()
    o = (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, Nq, C) * out_scale
    if accumulate:
        out += o.to(out.dtype)
        return out
    return o.to(q.dtype)


def attention2(q, k, v, k2, v2, heads, scale=None, out_scale=1.0, out_scale2=1.0, kv_group=1):
    a = attention(q, k, v, heads, scale=scale, out_scale=out_scale, kv_group=kv_group).float()
    b = attention(q, k2, v2, heads, scale=scale, out_scale=out_scale2, kv_group=kv_group).float()
    return (a + b).to(q.dtype)


def temporal_attention(qkv, B, Fr, P, heads, frame_major=False, out=None):
    C = qkv.shape[1] // 3
    x = qkv.reshape(Fr, B, P, 3 * C).permute(1, 0, 2, 3) if frame_major else qkv.reshape(B, Fr, P, 3 * C)
    t = x.permute(0, 2, 1, 3).reshape(B * P, Fr, 3 * C)
    o = attention(t[..., :C], t[..., C:2 * C], t[..., 2 * C:], heads).reshape(B, P, Fr, C)
    o = (o.permute(2, 0, 1, 3) if frame_major else o.permute(0, 2, 1, 3)).reshape(B * Fr * P, C).contiguous()
    if out is not None:
        out.copy_(o)
        return out
    return o


def shard_pack(src, dst, B, Fl, P, W, PP, unpack=False):
    C = src.shape[-1]
    if unpack:
        t = src.reshape(W, Fl, B, PP, C).permute(2, 1, 0, 3, 4).reshape(B, Fl, W * PP, C)[:, :, :P]
        dst.copy_(t.reshape(dst.shape))
    else:
        t = F.pad(src.reshape(B, Fl, P, C), (0, 0, 0, W * PP - P)).reshape(B, Fl, W, PP, C).permute(2, 1, 0, 3, 4)
        dst.copy_(t.reshape(dst.shape))
    return dst


def _pad_w(x_nchw, pad):
    return x_nchw if pad <= 0 else torch.cat([x_nchw[..., -pad:], x_nchw, x_nchw[..., :pad]], dim=-1)


def _cat(x):
    return torch.cat(list(x), dim=-1) if isinstance(x, (tuple, list)) else x


def group_norm_stats(x, gamma, beta, groups, eps, pad=0):
    x = _cat(x)
    N, H, W, C = x.shape
    xp = _pad_w(x.permute(0, 3, 1, 2).float(), pad).reshape(N, groups, -1)
    mean = xp.mean(-1)
    var = xp.var(-1, unbiased=False)
    rstd = (var + eps).rsqrt()
    cpg = C // groups
    scale = rstd.repeat_interleave(cpg, 1) * gamma.float()[None]
    shift = beta.float()[None] - mean.repeat_interleave(cpg, 1) * scale
    return scale, shift


def group_norm_apply(x, scale, shift, silu, pad=0):
    x = _cat(x)
    y = _pad_w(x.permute(0, 3, 1, 2).float(), pad).permute(0, 2, 3, 1) * scale[:, None, None, :] + shift[:, None, None, :]
    return (F.silu(y) if silu else y).to(x.dtype).contiguous()


def group_norm(x, gamma, beta, groups, eps, silu=False, pad=0):
    scale, shift = group_norm_stats(x, gamma, beta, groups, eps, pad)
    return group_norm_apply(x, scale, shift, silu, pad)


def pack_conv_weight(w, cin_pad=None):
    Cout, Cin, kh, kw = w.shape
    cin_pad = cin_pad or ((Cin + 31) // 32) * 32
    cout_pad = ((Cout + 127) // 128) * 128
    out = torch.zeros(cout_pad, kh * kw, cin_pad, dtype=w.dtype, device=w.device)
    out[:Cout, :, :Cin] = w.detach().permute(0, 2, 3, 1).reshape(Cout, kh * kw, Cin)
    return out


def pack_conv_up2_weight(w):
    from imagine360_amd import kernels
    w4 = kernels.up2_weights(w).to(w.dtype)
    return torch.stack([pack_conv_weight(w4[i]) for i in range(4)])


def conv_up2(x, w4_packed, cout, bias=None, wrap=False):
    """Four 2x2 convolutions of the low-resolution input, one per output parity (same pre-summed weights as the kernel)."""
    N, H, W, Cin = x.shape
    g = x.permute(0, 3, 1, 2).float()
    g = torch.cat([g[..., -1:], g, g[..., :1]], dim=-1) if wrap else F.pad(g, (1, 1, 0, 0))
    g = F.pad(g, (0, 0, 1, 1))
    y = torch.zeros(N, cout, 2 * H, 2 * W)
    for parity in range(4):
        py, px = parity >> 1, parity & 1
        w = w4_packed[parity][:cout].reshape(cout, 2, 2, Cin).permute(0, 3, 1, 2).float()
        y[:, :, py::2, px::2] = F.conv2d(g[:, :, py:py + H + 1, px:px + W + 1], w)
    if bias is not None:
        y = y + bias.float()[None, :, None, None]
    return y.permute(0, 2, 3, 1).to(x.dtype).contiguous()


def conv2d(x, w_packed, cout, bias=None, stride=1, up=False, wrap=False, x_off=0, wout=None, temb=None,
           imgs_per_temb=1, res=None, y_off=0, gn_stats=False):
    N, Hin, Win, Cin = x.shape
    taps = w_packed.shape[1]
    k = 3 if taps == 9 else 1
    w = w_packed[:cout].reshape(cout, k, k, Cin).permute(0, 3, 1, 2).float()
    g = x.permute(0, 3, 1, 2).float()
    if up:
        g = F.interpolate(g, scale_factor=2.0, mode="nearest")
    Wc = g.shape[-1]
    if k == 1:
        y = F.conv2d(g, w)
    else:
        if stride == 2 and x_off == 1 and y_off == 1:        # taps at 2o .. 2o+2: zero pad right/bottom only
            g = F.pad(g, (0, 1, 0, 1))
            x_off = 0
        else:
            assert y_off == 0
            g = _pad_w(g, 1) if wrap else F.pad(g, (1, 1, 0, 0))
            g = F.pad(g, (0, 0, 1, 1))
        y = F.conv2d(g, w, stride=stride)
    if wout is None:
        wout = Wc // stride
    y = y[..., x_off:x_off + wout] if stride == 1 else y[..., :wout]
    y = y.permute(0, 2, 3, 1)
    if bias is not None:
        y = y + bias.float()
    if temb is not None:
        y = y + temb.float().repeat_interleave(imgs_per_temb, 0)[:, None, None, :]
    if res is not None:
        y = y + res.float()
    return y.to(x.dtype).contiguous()


def conv1x1_cat(xa, xb, w_packed, cout, bias=None, res=None):
    return conv2d(torch.cat([xa, xb], dim=-1), w_packed, cout, bias=bias, res=res)


def linear(x, w_packed, n, bias=None, res=None, row_stats=False, gn_hw=None):
    """The statistics are those of the STORED (rounded) output, per 160-column slice, like the kernel's."""
    k = x.shape[-1]
    y = F.linear(x.float(), w_packed[:n, 0].float(), None if bias is None else bias.float())
    if res is not None:
        y = y + res.float().reshape(y.shape)
    y = y.to(x.dtype)
    if not row_stats:
        return y
    t = y.float().reshape(-1, n // 160, 160) if n % 160 == 0 else y.float().reshape(-1, 1, n)      # (narrow test models: one slice)
    return y, torch.stack([t.sum(-1), (t * t).sum(-1)], dim=-1).contiguous()


def _row_affine(x, w_packed, rows, c1, c2, stats, eps):
    k = x.shape[-1]
    s = stats.sum(dim=1)
    mu = s[:, 0] / k
    rstd = ((s[:, 1] / k - mu * mu).clamp_min(0) + eps).rsqrt()
    acc = x.float().reshape(-1, k) @ w_packed[:rows, 0].float().t()
    return (acc - mu[:, None] * c1[None]) * rstd[:, None] + c2[None]


def linear_ln(x, w_packed, c1, c2, stats, eps, n, tab=None, tab_div=1, tab_has_c2=False):
    y = _row_affine(x, w_packed, n, c1, c2, stats, eps)
    if tab is not None:
        r = torch.arange(y.shape[0])
        y = y + (tab - c2[None, :] if tab_has_c2 else tab)[(r // tab_div) % tab.shape[0]]
    return y.to(x.dtype).reshape(x.shape[:-1] + (n,))


def linear_geglu_ln(x, w_packed, c1, c2, stats, eps, inner):
    """w_packed / c1 / c2 are what ``pack_geglu`` of this module returns: the plain (value | gate) row order."""
    y = _row_affine(x, w_packed, 2 * inner, c1, c2, stats, eps)
    return geglu(y).to(x.dtype).reshape(x.shape[:-1] + (inner,))


def interleave_geglu(weight, bias):
    return weight, bias


def circular_pad_w(x, pad):
    return torch.cat([x[..., -pad:, :], x, x[..., :pad, :]], dim=-2).contiguous()


def circular_pad_hw(x, left, right, top=0, bottom=0):
    H, W = x.shape[-2], x.shape[-1]
    y = F.pad(x.reshape(1, -1, H, W), (left, right, top, bottom), mode="circular")
    return y.reshape(x.shape[:-2] + y.shape[-2:]).contiguous()


def cfg_ddim_update(uncond, cond, sample, guidance, cx, cv, coef_dev=None):
    if coef_dev is not None:
        guidance, cx, cv = (float(x) for x in coef_dev)
    return (cx * sample.float() + cv * (uncond.float() + guidance * (cond.float() - uncond.float()))).to(sample.dtype)


def layer_norm(x, gamma, beta, eps=1e-5, pre=None, post=None, post_div=1):
    C = x.shape[-1]
    t = x.reshape(-1, C).float()
    r = torch.arange(t.shape[0])
    if pre is not None:
        t = t + pre.float()[r % pre.shape[0]]
    y = F.layer_norm(t, (C,), gamma.float(), beta.float(), eps)
    if post is not None:
        y = y + post.float()[(r // post_div) % post.shape[0]]
    return y.to(x.dtype).reshape(x.shape)


def geglu(h):
    a, g = h.float().chunk(2, dim=-1)
    return (a * F.gelu(g)).to(h.dtype)


def softmax_rows(x, scale=1.0, out=None):
    y = torch.softmax(x.float() * scale, dim=-1).to(x.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def pack_geglu(weight, bias):
    return weight.reshape(weight.shape[0], 1, weight.shape[1]), bias


def linear_geglu(x, w, b, inner):
    return geglu(F.linear(x, w[:, 0], b))


_NAMES = ["layer_norm", "geglu", "pack_geglu", "linear_geglu", "attention", "temporal_attention", "group_norm_stats", "group_norm_apply", "group_norm", "pack_conv_weight",
          "conv2d", "pack_conv_up2_weight", "conv_up2", "circular_pad_w", "circular_pad_hw", "cfg_ddim_update", "softmax_rows", "attention2", "pack_attn_bias",
          "conv1x1_cat", "linear", "linear_ln", "linear_geglu_ln", "interleave_geglu", "shard_pack"]


def carry_gn(view, src):
    return view


@contextlib.contextmanager
def patched_kernels():
    """Route ``imagine360_amd.kernels.<fn>`` to the torch stand-ins for the duration of a CPU test."""
    from imagine360_amd import kernels
    saved = {n: getattr(kernels, n) for n in _NAMES}
    try:
        for n in _NAMES:
            setattr(kernels, n, globals()[n])
        yield
    finally:
        for n, f in saved.items():
            setattr(kernels, n, f)


@contextlib.contextmanager
def routed_gemms(min_tokens=0):
    """Inside ``patched_kernels``: make the host code take the branches it takes on the GPU for large token counts --
    token-major Linears through ``kernels.linear`` / ``conv2d`` with row statistics, LayerNorms folded into the consuming
    GEMMs, fused GEGLU -- so that the derived operands (gamma-scaled weights, c1 / c2 vectors, positional tables) are
    checked on CPU against the reference fixtures."""
    from imagine360_amd import layers
    saved = (layers.ROUTE_MIN_TOKENS, layers.ROUTE_ON_CPU, layers.ROUTE_N_MULT)
    layers.ROUTE_MIN_TOKENS, layers.ROUTE_ON_CPU, layers.ROUTE_N_MULT = min_tokens, True, 64
    try:
        yield
    finally:
        layers.ROUTE_MIN_TOKENS, layers.ROUTE_ON_CPU, layers.ROUTE_N_MULT = saved
