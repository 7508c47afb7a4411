"""torch-tensor front end of libim360_kernels.so (C ABI in include/im360_kernels.h).

Every function here launches a hand-written gfx950 kernel on ``torch.cuda.current_stream()``.
There is NO fallback: a missing library or a non-GPU tensor raises.  Model code calls these
through the module (``kernels.attention(...)``).
"""
import ctypes
import os

import torch

# IM360_KERNELS_LIB: another build of the SAME library for the A/B tools (`make -C csrc ablate-lib` -> libim360_kernels_ablate.so:
# the shipped kernels plus the measured-and-rejected variants); never a different implementation
_LIB_PATH = os.environ.get("IM360_KERNELS_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libim360_kernels.so")
_lib = None

_I64, _F32, _INT, _PTR = ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_void_p

_SIGNATURES = {
    "im360_abi_version": (_INT, []),
    "im360_build_flags": (_INT, []),
    "im360_last_error": (ctypes.c_char_p, []),
    "im360_attn_fwd": (_INT, [_PTR] * 5 + [_I64] * 15 + [_F32, _F32, _INT, _INT, _PTR, _PTR, _PTR, _PTR, _PTR, _I64]),
    "im360_attn_fwd2": (_INT, [_PTR] * 6 + [_I64] * 19 + [_F32, _F32, _F32, _INT, _PTR]),
    "im360_temporal_attn_fwd": (_INT, [_PTR] * 4 + [_I64] * 11 + [_F32, _INT, _PTR]),
    "im360_shard_pack": (_INT, [_PTR] * 2 + [_I64] * 6 + [_INT, _PTR]),
    "im360_gn_num_slabs": (_I64, [_I64] * 3),
    "im360_groupnorm_stats": (_INT, [_PTR] * 6 + [_I64] * 6 + [_F32, _INT, _PTR]),
    "im360_groupnorm_apply": (_INT, [_PTR] * 4 + [_I64] * 5 + [_INT, _INT, _PTR]),
    "im360_groupnorm_stats_cat": (_INT, [_PTR] * 7 + [_I64] * 7 + [_F32, _INT, _PTR]),
    "im360_groupnorm_fused": (_INT, [_PTR] * 7 + [_I64] * 7 + [_F32, _INT, _INT, _PTR]),
    "im360_groupnorm_apply_cat": (_INT, [_PTR] * 5 + [_I64] * 6 + [_INT, _INT, _PTR]),
    "im360_conv1x1_cat_fwd": (_INT, [_PTR] * 6 + [_I64] * 6 + [_INT, _PTR]),
    "im360_linear_fwd": (_INT, [_PTR] * 6 + [_I64] * 3 + [_INT, _PTR, _PTR]),
    "im360_linear_ln_fwd": (_INT, [_PTR] * 5 + [_I64, _F32, _PTR, _I64, _I64, _PTR] + [_I64] * 3 + [_INT, _PTR]),
    "im360_linear_geglu_ln": (_INT, [_PTR] * 5 + [_I64, _F32, _PTR] + [_I64] * 3 + [_INT, _PTR]),
    "im360_conv_fwd": (_INT, [_PTR] * 6 + [_I64] * 14 + [_INT, _PTR, _PTR]),
    "im360_conv_gn_slabs": (_I64, [_I64] * 6),
    "im360_conv_ksplit_plan": (_I64, [_I64] * 9),
    "im360_conv_fwd_ksplit": (_INT, [_PTR] * 6 + [_I64] * 14 + [_INT, _PTR, _PTR] + [_PTR, _I64, _PTR, _I64]),
    "im360_groupnorm_partial": (_INT, [_PTR] * 2 + [_I64] * 4 + [_INT, _PTR]),
    "im360_groupnorm_finalize": (_INT, [_PTR, _I64, _I64, _PTR, _I64, _I64] + [_PTR] * 4 + [_I64] * 3 + [_F32, _INT, _PTR]),
    "im360_groupnorm_partial_pad": (_INT, [_PTR] * 2 + [_I64] * 5 + [_INT, _PTR]),
    "im360_groupnorm_apply_partials": (_INT, [_PTR, _PTR, _PTR, _I64, _PTR, _I64, _PTR, _PTR, _PTR] + [_I64] * 7 + [_F32, _INT, _INT, _PTR]),
    "im360_pack_conv_weight": (_INT, [_PTR] * 2 + [_I64] * 5 + [_INT, _PTR]),
    "im360_attn_pack_bias": (_INT, [_PTR] * 2 + [_I64, _INT, _PTR]),
    "im360_conv_up2_fwd": (_INT, [_PTR] * 4 + [_I64] * 6 + [_INT, _PTR]),
    "im360_remap_cubic_wrap_u8": (_INT, [_PTR] * 5 + [_I64] * 7 + [_PTR]),
    "im360_max_rect": (_INT, [_PTR, _I64, _I64, _PTR]),
    "im360_circular_pad_w": (_INT, [_PTR] * 2 + [_I64] * 4 + [_INT, _PTR]),
    "im360_circular_pad_hw": (_INT, [_PTR] * 2 + [_I64] * 8 + [_PTR]),
    "im360_cfg_ddim_update": (_INT, [_PTR] * 4 + [_I64] + [_F32] * 3 + [_INT, _PTR, _PTR]),
    "im360_layernorm": (_INT, [_PTR] * 6 + [_I64] * 5 + [_F32, _INT, _PTR]),
    "im360_geglu": (_INT, [_PTR] * 2 + [_I64] * 2 + [_INT, _PTR]),
    "im360_linear_geglu": (_INT, [_PTR] * 4 + [_I64] * 3 + [_INT, _PTR]),
    "im360_softmax_rows": (_INT, [_PTR] * 2 + [_I64] * 4 + [_F32, _INT, _PTR]),
    "im360_tuning_set": (_INT, [_INT, _INT]),
    "im360_prof_enable": (None, [ctypes.c_uint]),
    "im360_prof_collect": (_INT, [_INT, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]),
}

ABI_VERSION = 5          # include/im360_kernels.h: what im360_abi_version() of a matching library returns

PROF_KINDS = {"attn": 0, "temporal": 1, "conv": 2, "gn_stats": 3, "gn_apply": 4, "misc": 5, "gemm": 6, "attn_warp": 7, "attn_x2": 8}


# Algorithmic work of the launches, by class (bench.py's roofline legs): None = off, else {class: [flops, bytes, launches]}
STATS = None


# Launch mix by shape (bench.py weights the measured HBM traffic of tools/hbm_traffic.py with it): None = off, else
# {(class, shape key): [launches, algorithmic bytes]}
SHAPES = None


def _count(kind, flops, nbytes, executed=None, shape=None):
    """``flops``: the reference algorithm's flops of the launch; ``executed``: what the kernel actually multiplies when it
    differs (the sub-pixel upsample convolutions run 4 / 9 of them) -- hardware utilisation is priced on executed flops.
    ``shape``: a key naming the launch's shape class (token count excluded), for the launch-mix statistics."""
    if STATS is not None:
        s = STATS.setdefault(kind, [0.0, 0.0, 0, 0.0])
        s[0] += flops
        s[1] += nbytes
        s[2] += 1
        s[3] += flops if executed is None else executed
    if SHAPES is not None and shape is not None:
        h = SHAPES.setdefault((kind, shape), [0, 0.0])
        h[0] += 1
        h[1] += nbytes


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    """Load libim360_kernels.so (built by ``__graft_entry__.build()`` / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: build it with `make -C imagine360_amd/csrc` "
                               "(hipcc --offload-arch=gfx950); imagine360_amd has no CPU fallback")
        L = ctypes.CDLL(_LIB_PATH)
        # the version first: a stale library that lacks a newer symbol must fail with the rebuild message, not with an
        # AttributeError from the loop below
        try:
            ver = L.im360_abi_version
        except AttributeError:
            raise RuntimeError(f"{_LIB_PATH} exports no im360_abi_version: rebuild it with `make -C imagine360_amd/csrc`") from None
        ver.restype, ver.argtypes = _SIGNATURES["im360_abi_version"]
        have = ver()
        if have != ABI_VERSION:
            # (version 1 -> 2 added a trailing pointer to two entry points; a stale library would read garbage for it)
            raise RuntimeError(f"{_LIB_PATH} reports C-ABI version {have}, this binding is written for version {ABI_VERSION}: "
                               "rebuild it with `make -C imagine360_amd/csrc`")
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().im360_last_error().decode()}")


def _dt(t):
    if t.dtype == torch.bfloat16:
        return 0
    if t.dtype == torch.float16:
        return 1
    raise TypeError(f"imagine360_amd kernels take bfloat16/float16 tensors, got {t.dtype}")


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("imagine360_amd kernels need tensors on the MI355X (cuda) device; no CPU fallback")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------ attention
def pack_attn_bias(bias):
    """bias (16-bit, any shape) -> fp16 ``bias * log2(e)`` clamped to +-60000 (finite: -inf masks become an exact zero
    weight instead of poisoning the identity-slice MFMA that adds them): the form ``attention(..., bias_packed=True)``
    feeds to the matrix pipe (head dim 32, Nk % 8 == 0).  Once per cached mask."""
    _dev(bias)
    b = bias.contiguous()
    out = torch.empty(b.shape, dtype=torch.float16, device=b.device)
    _check(lib().im360_attn_pack_bias(_p(b), _p(out), b.numel(), _dt(b), _stream()), "im360_attn_pack_bias")
    return out


def attn_bias_blocks(packed):
    """Block map of a packed bias matrix [Nq, Nk] (fp16, ``pack_attn_bias``): int32 [ceil(Nq / 32), ceil(Nk / 1024)], bit h % 32 of
    word h // 32 of row r is set when the 32 x 32 block (query rows 32 r .., keys 32 h ..) holds a non-zero entry.  The attention
    kernel skips the bias of blocks whose bit is clear; WarpAttn shifts its masks so that the background is exactly zero
    (softmax is invariant under a per-row constant)."""
    nq, nk = packed.shape
    rq, rk = -(-nq // 32), -(-nk // 32)
    nz = torch.zeros((rq * 32, rk * 32), dtype=torch.bool, device=packed.device)
    nz[:nq, :nk] = packed != 0
    blocks = nz.view(rq, 32, rk, 32).any(dim=3).any(dim=1)                      # [rq, rk]
    words = -(-rk // 32)
    bits = torch.zeros((rq, words * 32), dtype=torch.int64, device=packed.device)
    bits[:, :rk] = blocks.to(torch.int64)
    w = (bits.view(rq, words, 32) << torch.arange(32, device=packed.device, dtype=torch.int64)).sum(dim=2)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()


def can_pack_attn_bias(d, nk):
    return d == 32 and nk % 8 == 0


def attention(q, k, v, heads, scale=None, bias=None, out=None, accumulate=False, out_scale=1.0, kv_group=1,
              bias_alt=None, bias_sel=None, bias_packed=False, bias_blocks=None, bias_blocks_alt=None):
    """q [B, Nq, heads*d], k/v [B / kv_group, Nk, heads*d] (last dim contiguous, any row/batch stride),
    bias [Nq, Nk] shared by every (batch, head); with ``bias_sel`` (device int32 scalar) the kernel picks
    ``bias_alt`` when it is non-zero.  ``bias_packed``: bias / bias_alt come from ``pack_attn_bias``; ``bias_blocks`` /
    ``bias_blocks_alt``: their block maps (``attn_bias_blocks``) -- 32 x 32 blocks of zeros are skipped.
    Returns out [B, Nq, heads*d]."""
    _dev(q, k, v, bias, out)
    B, Nq, C = q.shape
    Nk = k.shape[1]
    d = C // heads
    assert k.shape[0] * kv_group == B and v.shape[0] == k.shape[0] and v.shape[1] == Nk and k.shape[2] == C and v.shape[2] == C
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    if out is None:
        assert not accumulate
        out = torch.empty((B, Nq, C), dtype=q.dtype, device=q.device)
    else:
        _written(out)
    assert out.stride(2) == 1
    if bias is not None:
        assert bias.shape == (Nq, Nk) and bias.stride(1) == 1 and bias.dtype == (torch.float16 if bias_packed else q.dtype)
        assert bias_alt is None or bias_alt.dtype == bias.dtype
    if scale is None:
        scale = d ** -0.5
    rc = lib().im360_attn_fwd(_p(q), _p(k), _p(v), _p(bias), _p(out), B, heads, Nq, Nk, d,
                              q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                              out.stride(0), out.stride(1), bias.stride(0) if bias is not None else 0, kv_group,
                              float(scale), float(out_scale), int(accumulate), _dt(q) + (256 if bias_packed else 0), _stream(), _p(bias_alt), _p(bias_sel),
                              _p(bias_blocks), _p(bias_blocks_alt), bias_blocks.shape[1] if bias_blocks is not None else 0)
    _check(rc, "im360_attn_fwd")
    _count("attn" if bias is None else "attn_warp", 4.0 * B * heads * Nq * Nk * d, q.element_size() * (2 * B * Nq * C + 2 * k.shape[0] * Nk * C)
           + (0 if bias is None else bias.element_size() * Nq * Nk))
    return out


def attention2(q, k, v, k2, v2, heads, scale=None, out_scale=1.0, out_scale2=1.0, kv_group=1):
    """out_scale * attn(q, k, v) + out_scale2 * attn(q, k2, v2) in ONE launch (two independent softmaxes over two
    key / value sets: the text and the IP-adapter tokens of IPCrossAttention, animatediff/models/attention.py:113-148).
    Shapes as in ``attention``; head dim 64."""
    _dev(q, k, v, k2, v2)
    B, Nq, C = q.shape
    d = C // heads
    for t in (k, v, k2, v2):
        assert t.shape[0] * kv_group == B and t.shape[2] == C and t.stride(2) == 1
    assert v.shape[1] == k.shape[1] and v2.shape[1] == k2.shape[1] and q.stride(2) == 1
    out = torch.empty((B, Nq, C), dtype=q.dtype, device=q.device)
    if scale is None:
        scale = d ** -0.5
    rc = lib().im360_attn_fwd2(_p(q), _p(k), _p(v), _p(k2), _p(v2), _p(out), B, heads, Nq, k.shape[1], k2.shape[1], d,
                               q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                               k2.stride(0), k2.stride(1), v2.stride(0), v2.stride(1), out.stride(0), out.stride(1),
                               kv_group, float(scale), float(out_scale), float(out_scale2), _dt(q), _stream())
    _check(rc, "im360_attn_fwd2")
    _count("attn_x2", 4.0 * B * heads * Nq * (k.shape[1] + k2.shape[1]) * d,
           q.element_size() * (2 * B * Nq * C + 2 * k.shape[0] * (k.shape[1] + k2.shape[1]) * C))
    return out


def temporal_attention(qkv, B, F, P, heads, frame_major=False, out=None):
    """qkv [B*F*P, 3*C] = fused (q | k | v) projection of token-major activations [B, F, P, C].
    Attention over the F axis per (batch, pixel, head).  Returns [B*F*P, C].
    ``frame_major``: rows (and the result's) are ordered [F, B, P] instead -- the receive layout of the frame <-> pixel
    all-to-all (``shard_pack``), read in place through the kernel's strides.  ``out``: preallocated result."""
    _dev(qkv, out)
    C = qkv.shape[1] // 3
    assert qkv.shape[0] == B * F * P and qkv.stride(1) == 1
    rs = qkv.stride(0)
    if out is None:
        out = torch.empty((B * F * P, C), dtype=qkv.dtype, device=qkv.device)
    else:
        _written(out)
    assert out.shape == (B * F * P, C) and out.is_contiguous() and out.dtype == qkv.dtype
    d = C // heads
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    fs, bs = (B * P, P) if frame_major else (P, F * P)          # frame / batch strides in rows
    rc = lib().im360_temporal_attn_fwd(_p(q), _p(k), _p(v), _p(out), B, F, P, heads, d,
                                       fs * rs, rs, bs * rs, fs * C, C, bs * C,
                                       float(d ** -0.5), _dt(qkv), _stream())
    _check(rc, "im360_temporal_attn_fwd")
    _count("temporal", 4.0 * B * P * heads * F * F * d, qkv.element_size() * 4 * B * F * P * C)
    return out


def shard_pack(src, dst, B, Fl, P, W, PP, unpack=False):
    """Frame-sharded tokens [B, Fl, P, C] <-> the all-to-all buffer [W, Fl, B, PP, C] (``im360_shard_pack``); both tensors are
    caller-owned (pre-sized, so the exchange can be graph-captured).  ``unpack``: buffer (src) -> tokens (dst)."""
    _dev(src, dst)
    C = src.shape[-1]
    assert src.is_contiguous() and dst.is_contiguous() and dst.shape[-1] == C and src.dtype == dst.dtype and src.element_size() == 2
    _written(dst)
    tok, buf = (dst, src) if unpack else (src, dst)
    assert tok.numel() == B * Fl * P * C and buf.numel() == W * Fl * B * PP * C, (tok.shape, buf.shape)
    rc = lib().im360_shard_pack(_p(src), _p(dst), B, Fl, P, C, W, PP, int(bool(unpack)), _stream())
    _check(rc, "im360_shard_pack")
    _count("misc", 0.0, 2 * tok.element_size() * tok.numel())
    return dst


# ------------------------------------------------------------------------------------------ group norm
# GroupNorm statistics from the producer's epilogue: conv2d(..., gn_stats=True) / linear(..., gn_hw=pixels)
# attach `(partial sums, slabs per image)` to the tensor they return when the launch can write them (im360_conv_gn_slabs);
# group_norm_stats() then skips its own pass over that tensor.  The tag belongs to the tensor OBJECT and dies with a view /
# reshape (carry_gn moves it onto a reshaped view of the same rows) or an in-place write (checked through _version).
GN_FROM_PRODUCER = True


def _gn_of(t):
    g = getattr(t, "_im360_gn", None)
    if g is None or g[2] != t._version:
        return None
    return g[0], g[1]


def _tag_gn(t, partial, slabs):
    t._im360_gn = (partial, slabs, t._version)
    return t


def _written(t):
    """The ctypes kernels write through raw pointers and do not bump ``t._version`` (the tag's invalidation rule): every wrapper
    that writes into a CALLER-SUPPLIED tensor (``out=`` / ``accumulate`` / a destination buffer) drops a producer tag it may carry."""
    if t is not None and getattr(t, "_im360_gn", None) is not None:
        del t._im360_gn
    return t


def carry_gn(view, src):
    """``view`` is a reshape of ``src`` with the same rows (tokens [n, hw, c] <-> images [n, h, w, c]): keep the producer's partial sums."""
    g = getattr(src, "_im360_gn", None)
    if g is not None and g[2] == src._version and view.numel() == src.numel():
        view._im360_gn = (g[0], g[1], view._version)
    return view


def group_norm_stats(x, gamma, beta, groups, eps, pad=0):
    """x [N, H, W, C] channels-last, or a pair (xa [N, H, W, C1], xb [N, H, W, C2]) standing for their channel
    concatenation (never materialised).  Returns fp32 (scale, shift) [N, C] with GN(x) = x*scale + shift;
    pad > 0 = statistics of the circularly W-padded tensor."""
    xa, xb = x if isinstance(x, (tuple, list)) else (x, None)
    _dev(xa, xb, gamma, beta)
    N, H, W, C1 = xa.shape
    C = C1 + (xb.shape[-1] if xb is not None else 0)
    assert xa.is_contiguous() and gamma.dtype == xa.dtype and beta.dtype == xa.dtype and gamma.numel() == C
    scale = torch.empty((N, C), dtype=torch.float32, device=xa.device)
    shift = torch.empty((N, C), dtype=torch.float32, device=xa.device)
    srcs = [xa] if xb is None else [xa, xb]
    if pad == 0 and GN_FROM_PRODUCER and any(_gn_of(t) is not None for t in srcs):
        # at least one source carries partial sums from the epilogue of the kernel that produced it (conv2d / linear with
        # gn_stats): only the others get a statistics pass, then one finalize over both sets of partial sums
        parts, read = [], 0
        for t in srcs:
            g = _gn_of(t)
            if g is None:
                St = lib().im360_gn_num_slabs(N, H, W)
                buf = torch.empty((N * St * 2 * t.shape[-1],), dtype=torch.float32, device=t.device)
                assert t.is_contiguous() and t.shape[:3] == xa.shape[:3] and t.dtype == xa.dtype
                _check(lib().im360_groupnorm_partial(_p(t), _p(buf), N, H, W, t.shape[-1], _dt(t), _stream()), "im360_groupnorm_partial")
                g = (buf, St)
                read += t.element_size() * t.numel()
            parts.append(g)
        (pa, Sa), (pb, Sb) = parts[0], (parts[1] if len(parts) > 1 else (None, 0))
        rc = lib().im360_groupnorm_finalize(_p(pa), Sa, C1, _p(pb), Sb, C - C1, _p(gamma), _p(beta), _p(scale), _p(shift), N, groups,
                                            H * W, float(eps), _dt(xa), _stream())
        _check(rc, "im360_groupnorm_finalize")
        _count("gn_stats", 0.0, read + 4 * sum(p_.numel() for p_, _ in parts))
        return scale, shift
    S = lib().im360_gn_num_slabs(N, H, W)
    partial = torch.empty((N * S * 2 * C,), dtype=torch.float32, device=xa.device)
    if xb is None:
        rc = lib().im360_groupnorm_stats(_p(xa), _p(gamma), _p(beta), _p(partial), _p(scale), _p(shift),
                                         N, H, W, C, groups, pad, float(eps), _dt(xa), _stream())
        _check(rc, "im360_groupnorm_stats")
    else:
        assert xb.is_contiguous() and xb.shape[:3] == xa.shape[:3] and xb.dtype == xa.dtype
        rc = lib().im360_groupnorm_stats_cat(_p(xa), _p(xb), _p(gamma), _p(beta), _p(partial), _p(scale), _p(shift),
                                             N, H, W, C1, C - C1, groups, pad, float(eps), _dt(xa), _stream())
        _check(rc, "im360_groupnorm_stats_cat")
    _count("gn_stats", 0.0, xa.element_size() * N * H * W * C)
    return scale, shift


def group_norm_apply(x, scale, shift, silu, pad=0):
    """y [N, H, W + 2 pad, C] = act(x * scale + shift), circular along W; x as in ``group_norm_stats`` (a pair writes the
    normalised concatenation directly)."""
    xa, xb = x if isinstance(x, (tuple, list)) else (x, None)
    _dev(xa, xb, scale, shift)
    N, H, W, C1 = xa.shape
    C = C1 + (xb.shape[-1] if xb is not None else 0)
    assert xa.is_contiguous() and scale.shape == (N, C)
    y = torch.empty((N, H, W + 2 * pad, C), dtype=xa.dtype, device=xa.device)
    if xb is None:
        rc = lib().im360_groupnorm_apply(_p(xa), _p(scale), _p(shift), _p(y), N, H, W, C, pad, int(bool(silu)),
                                         _dt(xa), _stream())
        _check(rc, "im360_groupnorm_apply")
    else:
        assert xb.is_contiguous() and xb.shape[:3] == xa.shape[:3] and xb.dtype == xa.dtype
        rc = lib().im360_groupnorm_apply_cat(_p(xa), _p(xb), _p(scale), _p(shift), _p(y), N, H, W, C1, C - C1, pad,
                                             int(bool(silu)), _dt(xa), _stream())
        _check(rc, "im360_groupnorm_apply_cat")
    _count("gn_apply", 0.0, xa.element_size() * (N * H * W * C + y.numel()))
    return y


# group_norm(): True = ONE launch (im360_groupnorm_fused: statistics, per-image arrival counter, in-kernel reduction,
# normalisation; same bits).  Measured SLOWER than the three-kernel path -- 0.299 vs 0.264 ms on 640 x 32x32x320, 0.315 vs
# 0.101 ms on 32 x 64x128x320 (64 slabs per image wait for each other) -- the arrival waits and the write-through publish
# of the partials cost more than the HBM pass the cache-resident second read saves; and it must not be captured in a
# hipGraph together with kernels it could wait behind.  Kept as an A/B variant, off.
GN_FUSED = False


def group_norm_partials(x, pad=0):
    """Per-image partial sums of x [N, H, W, C] as (buffer fp32 [N][S][2][C], S): the ones the tensor's producer wrote in its
    epilogue (pad == 0 only: the padded statistics weight the wrapped columns twice), else one statistics pass over x."""
    g = _gn_of(x) if (pad == 0 and GN_FROM_PRODUCER) else None
    if g is not None:
        return g
    N, H, W, C = x.shape
    assert x.is_contiguous()
    S = lib().im360_gn_num_slabs(N, H, W)
    buf = torch.empty((N * S * 2 * C,), dtype=torch.float32, device=x.device)
    if pad:
        _check(lib().im360_groupnorm_partial_pad(_p(x), _p(buf), N, H, W, C, pad, _dt(x), _stream()), "im360_groupnorm_partial_pad")
    else:
        _check(lib().im360_groupnorm_partial(_p(x), _p(buf), N, H, W, C, _dt(x), _stream()), "im360_groupnorm_partial")
    _count("gn_stats", 0.0, x.element_size() * x.numel() + 4 * buf.numel())
    return buf, S


# group_norm(): "partials" (default since round 6) = statistics pass only where the producer did not leave partial sums, then ONE
# launch that rebuilds scale / shift per workgroup and normalises (im360_groupnorm_apply_partials: no finalize launch, the
# concatenation in one launch instead of two); "three" = rounds 2 - 5's statistics + finalize + apply (same bits; the A/B partner);
# GN_FUSED below = the single-launch variant with arrival counters (measured slower).
GN_MODE = os.environ.get("IM360_GN_MODE", "partials")


def group_norm(x, gamma, beta, groups, eps, silu=False, pad=0):
    """act(GroupNorm(x)) [N, H, W + 2 pad, C]; x [N, H, W, C] or a pair standing for a channel concatenation (see GN_MODE)."""
    if not GN_FUSED and GN_MODE == "partials":
        xa, xb = x if isinstance(x, (tuple, list)) else (x, None)
        _dev(xa, xb, gamma, beta)
        N, H, W, C1 = xa.shape
        C2 = xb.shape[-1] if xb is not None else 0
        assert xa.is_contiguous() and gamma.dtype == xa.dtype and beta.dtype == xa.dtype and gamma.numel() == C1 + C2
        assert xb is None or (xb.is_contiguous() and xb.shape[:3] == xa.shape[:3] and xb.dtype == xa.dtype)
        pa, Sa = group_norm_partials(xa, pad)
        pb, Sb = group_norm_partials(xb, pad) if xb is not None else (None, 0)
        y = torch.empty((N, H, W + 2 * pad, C1 + C2), dtype=xa.dtype, device=xa.device)
        rc = lib().im360_groupnorm_apply_partials(_p(xa), _p(xb), _p(pa), Sa, _p(pb), Sb, _p(gamma), _p(beta), _p(y), N, H, W, C1, C2,
                                                  groups, pad, float(eps), int(bool(silu)), _dt(xa), _stream())
        _check(rc, "im360_groupnorm_apply_partials")
        _count("gn_apply", 0.0, xa.element_size() * (N * H * W * (C1 + C2) + y.numel()))
        return y
    if not GN_FUSED:
        scale, shift = group_norm_stats(x, gamma, beta, groups, eps, pad)
        return group_norm_apply(x, scale, shift, silu, pad)
    xa, xb = x if isinstance(x, (tuple, list)) else (x, None)
    _dev(xa, xb, gamma, beta)
    N, H, W, C1 = xa.shape
    C2 = xb.shape[-1] if xb is not None else 0
    C = C1 + C2
    assert xa.is_contiguous() and gamma.dtype == xa.dtype and beta.dtype == xa.dtype and gamma.numel() == C
    assert xb is None or (xb.is_contiguous() and xb.shape[:3] == xa.shape[:3] and xb.dtype == xa.dtype)
    S = lib().im360_gn_num_slabs(N, H, W)
    partial = torch.empty((N * S * 2 * C,), dtype=torch.float32, device=xa.device)
    counter = torch.empty((N,), dtype=torch.int32, device=xa.device)
    y = torch.empty((N, H, W + 2 * pad, C), dtype=xa.dtype, device=xa.device)
    rc = lib().im360_groupnorm_fused(_p(xa), _p(xb), _p(gamma), _p(beta), _p(partial), _p(counter), _p(y), N, H, W, C1, C2,
                                     groups, pad, float(eps), int(bool(silu)), _dt(xa), _stream())
    _check(rc, "im360_groupnorm_fused")
    _count("gn_apply", 0.0, xa.element_size() * (N * H * W * C + y.numel()))          # algorithmic: x read once, y written once
    return y


# ------------------------------------------------------------------------------------------ convolution
def pack_conv_weight(w, cin_pad=None):
    """[Cout, Cin, kh, kw] -> packed [CoutPad128, kh*kw, CinPad32] (zero padded)."""
    _dev(w)
    Cout, Cin, kh, kw = w.shape
    taps = kh * kw
    assert taps in (1, 4, 9)
    cin_pad = cin_pad or ((Cin + 31) // 32) * 32
    cout_pad = ((Cout + 127) // 128) * 128
    out = torch.empty((cout_pad, taps, cin_pad), dtype=w.dtype, device=w.device)
    rc = lib().im360_pack_conv_weight(_p(w.contiguous()), _p(out), Cout, Cin, taps, cout_pad, cin_pad, _dt(w), _stream())
    _check(rc, "im360_pack_conv_weight")
    return out


def up2_weights(w):
    """[Cout, Cin, 3, 3] -> [4, Cout, Cin, 2, 2] (fp32): the taps of conv3x3-after-nearest-x2 that land on the same source
    pixel, summed, per output parity (py, px) = (0,0) (0,1) (1,0) (1,1).  Row parity 0 reads source rows (y - 1, y) with
    (k0, k1 + k2); parity 1 reads (y, y + 1) with (k0 + k1, k2); columns alike."""
    w = w.detach().float()
    rows = (torch.stack([w[:, :, 0], w[:, :, 1] + w[:, :, 2]], dim=2), torch.stack([w[:, :, 0] + w[:, :, 1], w[:, :, 2]], dim=2))   # [Cout, Cin, 2, 3]
    out = []
    for py in (0, 1):
        r = rows[py]
        for px in (0, 1):
            out.append(torch.stack([r[..., 0], r[..., 1] + r[..., 2]], dim=-1) if px == 0 else torch.stack([r[..., 0] + r[..., 1], r[..., 2]], dim=-1))
    return torch.stack(out)


def pack_conv_up2_weight(w):
    """Conv weight [Cout, Cin, 3, 3] (Cin % 64 == 0) -> the four packed 4-tap weights of ``conv_up2`` [4, CoutPad, 4, Cin]."""
    _dev(w)
    w4 = up2_weights(w).to(w.dtype)
    return torch.stack([pack_conv_weight(w4[i].contiguous()) for i in range(4)]).contiguous()


def chunk_major(w_packed):
    """Packed token-major weight [rows, 1, K] (``pack_conv_weight`` / ``pack_geglu``) -> the same elements in CHUNK-MAJOR order
    [K / 32][rows][32] (returned with the original shape): the operand layout of the A/B kernel behind knob conv_ring 11, where a
    32-channel half stage of a weight tile is one contiguous block of whole 128-byte lines."""
    r, t, k = w_packed.shape
    assert t == 1 and k % 32 == 0
    return w_packed.reshape(r, k // 32, 32).permute(1, 0, 2).contiguous().reshape(r, 1, k)


def conv_up2(x, w4_packed, cout, bias=None, wrap=False):
    """nearest-x2 upsample + conv3x3 (pad 1, ``wrap``: circular along W) of x [N, Hin, Win, Cin] -> [N, 2 Hin, 2 Win, Cout],
    as four 2 x 2 convolutions of the low-resolution input (4 / 9 of the MACs)."""
    _dev(x, w4_packed, bias)
    N, Hin, Win, Cin = x.shape
    assert x.is_contiguous() and w4_packed.is_contiguous() and w4_packed.shape[0] == 4 and w4_packed.shape[2] == 4 and w4_packed.shape[3] == Cin
    y = torch.empty((N, 2 * Hin, 2 * Win, cout), dtype=x.dtype, device=x.device)
    rc = lib().im360_conv_up2_fwd(_p(x), _p(w4_packed), _p(bias), _p(y), N, Hin, Win, Cin, cout, int(wrap), _dt(x), _stream())
    _check(rc, "im360_conv_up2_fwd")
    _count("conv", 2.0 * N * 4 * Hin * Win * cout * Cin * 9, x.element_size() * (x.numel() + y.numel() + 9 * cout * Cin),
           executed=2.0 * N * 4 * Hin * Win * cout * Cin * 4, shape=f"up2:{Cin}:{cout}:{2 * Hin}x{2 * Win}{'w' if wrap else ''}")
    return y


def conv2d(x, w_packed, cout, bias=None, stride=1, up=False, wrap=False, x_off=0, wout=None,
           temb=None, imgs_per_temb=1, res=None, y_off=0, gn_stats=False):
    """x [N, Hin, Win, Cin] -> y [N, Hout, Wout, Cout] (3x3 pad 1 or 1x1, see im360_conv_fwd)."""
    _dev(x, w_packed, bias, temb, res)
    N, Hin, Win, Cin = x.shape
    taps = w_packed.shape[1]
    assert x.is_contiguous() and w_packed.shape[2] == Cin, (x.shape, w_packed.shape)
    hc, wc = (2 * Hin, 2 * Win) if up else (Hin, Win)
    hout = hc // stride
    if wout is None:
        wout = wc // stride
    y = torch.empty((N, hout, wout, cout), dtype=x.dtype, device=x.device)
    if res is not None:
        assert res.shape == y.shape and res.is_contiguous()
    if temb is not None:
        assert temb.is_contiguous() and temb.shape[1] == cout and temb.shape[0] * imgs_per_temb == N
    gn, slabs = None, 0
    if gn_stats and GN_FROM_PRODUCER and not up:
        slabs = lib().im360_conv_gn_slabs(N, hout, wout, Cin, cout, taps)
        if slabs > 0:
            gn = torch.empty((N * slabs * 2 * cout,), dtype=torch.float32, device=x.device)
    # K-split (knob conv_ksplit): launches whose tile count is not a whole number of rounds of the chip run every tile in `parts` workgroups
    parts = lib().im360_conv_ksplit_plan(N, hout, wout, Cin, cout, taps, int(up), int(wrap), int(gn is not None)) if taps == 9 else 1
    if parts > 1:
        tiles = -(-(N * hout * wout) // 256) * (cout // 320)
        ws = torch.empty((tiles * (parts - 1) * 160 * 512,), dtype=torch.float32, device=x.device)
        cnt = torch.zeros((tiles,), dtype=torch.int32, device=x.device)
        rc = lib().im360_conv_fwd_ksplit(_p(x), _p(w_packed), _p(bias), _p(temb), _p(res), _p(y),
                                         N, Hin, Win, Cin, hout, wout, cout, taps, stride, int(up), int(wrap), x_off, y_off,
                                         imgs_per_temb, _dt(x), _stream(), _p(gn), _p(ws), ws.numel() * 4, _p(cnt), tiles)
        _check(rc, "im360_conv_fwd_ksplit")
    else:
        rc = lib().im360_conv_fwd(_p(x), _p(w_packed), _p(bias), _p(temb), _p(res), _p(y),
                                  N, Hin, Win, Cin, hout, wout, cout, taps, stride, int(up), int(wrap), x_off, y_off,
                                  imgs_per_temb, _dt(x), _stream(), _p(gn))
        _check(rc, "im360_conv_fwd")
    if gn is not None:
        _tag_gn(y, gn, slabs)
    if STATS is not None or SHAPES is not None:
        es = x.element_size()
        linear = taps == 1 and Hin == 1 and Win == 1
        _count("gemm" if linear else "conv", 2.0 * N * hout * wout * Cin * cout * taps,
               es * (x.numel() + cout * taps * Cin + y.numel() * (2 if res is not None else 1)),
               shape=(f"lin:{Cin}:{cout}:{'res' if res is not None else 'nores'}" if linear else
                      f"conv{taps}:{Cin}:{cout}:{hout}x{wout}:s{stride}{'u' if up else ''}{'w' if wrap else ''}{'o' if x_off else ''}"))
    return y


def can_conv1x1_cat(c1, c2):
    return c1 % 64 == 0 and c2 % 64 == 0


def conv1x1_cat(xa, xb, w_packed, cout, bias=None, res=None):
    """1x1 convolution of the channel concatenation [xa | xb] (never materialised): xa [N, H, W, C1], xb [N, H, W, C2],
    w_packed [CoutPad, 1, C1 + C2] -> [N, H, W, Cout] (+ bias, + res).  C1, C2 multiples of 64."""
    _dev(xa, xb, w_packed, bias, res)
    N, H, W, C1 = xa.shape
    C2 = xb.shape[-1]
    assert xa.is_contiguous() and xb.is_contiguous() and xb.shape[:3] == xa.shape[:3] and xb.dtype == xa.dtype
    assert w_packed.shape[1] == 1 and w_packed.shape[2] == C1 + C2, (w_packed.shape, C1, C2)
    y = torch.empty((N, H, W, cout), dtype=xa.dtype, device=xa.device)
    assert res is None or (res.shape == y.shape and res.is_contiguous())
    rc = lib().im360_conv1x1_cat_fwd(_p(xa), _p(xb), _p(w_packed), _p(bias), _p(res), _p(y), N, H, W, C1, C2, cout,
                                     _dt(xa), _stream())
    _check(rc, "im360_conv1x1_cat_fwd")
    _count("conv", 2.0 * N * H * W * (C1 + C2) * cout,
           xa.element_size() * (xa.numel() + xb.numel() + cout * (C1 + C2) + y.numel() * (2 if res is not None else 1)),
           shape=f"conv1cat:{C1}+{C2}:{cout}:{H}x{W}")
    return y


# ------------------------------------------------------------------------------------------ token-major linears
ROW_SLICE = 160          # columns per (sum, sum of squares) pair of the row statistics (one wave's share of a 320-column tile)


def linear(x, w_packed, n, bias=None, res=None, row_stats=False, gn_hw=None):
    """x [..., K] @ W^T + bias (+ res) -> [..., n] on the persistent ring kernel (n % 320 == 0, K % 32 == 0; w_packed from
    ``pack_conv_weight`` of the [n, K, 1, 1] view).  ``row_stats``: also return fp32 [M, n / 160, 2] = per row and
    160-column slice (sum, sum of squares) of the stored output -- the LayerNorm statistics a consumer with the
    normalisation folded in (``linear_ln`` / ``linear_geglu_ln``) needs."""
    _dev(x, w_packed, bias, res)
    k = x.shape[-1]
    m = x.numel() // k
    assert x.is_contiguous() and w_packed.shape[2] == k and w_packed.shape[1] == 1 and n % 320 == 0
    y = torch.empty(x.shape[:-1] + (n,), dtype=x.dtype, device=x.device)
    assert res is None or (res.is_contiguous() and res.numel() == y.numel() and res.dtype == x.dtype)
    st = torch.empty((m, n // ROW_SLICE, 2), dtype=torch.float32, device=x.device) if row_stats else None
    # gn_hw: the rows are images of gn_hw pixels each; when those are whole 256-row tiles the epilogue also writes GroupNorm
    # partial sums (one slab per tile) for the consumer's GroupNorm
    gn = None
    if gn_hw and GN_FROM_PRODUCER and gn_hw % 256 == 0 and m % gn_hw == 0 and k % 64 == 0:
        gn = torch.empty(((m // 256) * 2 * n,), dtype=torch.float32, device=x.device)
    rc = lib().im360_linear_fwd(_p(x), _p(w_packed), _p(bias), _p(res), _p(y), _p(st), m, k, n, _dt(x), _stream(), _p(gn))
    _check(rc, "im360_linear_fwd")
    if gn is not None:
        _tag_gn(y, gn, gn_hw // 256)
    _count("gemm", 2.0 * m * k * n, x.element_size() * (x.numel() + n * k + y.numel() * (2 if res is not None else 1))
           + (0 if st is None else 4 * st.numel()), shape=f"lin{'+stats' if row_stats else ''}:{k}:{n}:{'res' if res is not None else 'nores'}")
    return (y, st) if row_stats else y


def linear_ln(x, w_packed, c1, c2, stats, eps, n, tab=None, tab_div=1, tab_has_c2=False):
    """Linear(LayerNorm(x)) with the normalisation folded into the GEMM: x [..., K] RAW rows, ``stats`` their statistics
    from the producer (``linear(..., row_stats=True)``), w_packed = pack(gamma (.) W), c1 = row sums of the rounded
    gamma (.) W, c2 = W beta + bias (fp32 [n]); ``tab`` fp32 [Q, n]: row r also gets tab[(r // tab_div) % Q].
    The kernel takes table rows that INCLUDE c2 (one column vector per tile, fetched by LDS-DMA): ``tab_has_c2`` says the
    caller folded it already (``layers.ln_linear`` caches the folded table), otherwise it is added here."""
    _dev(x, w_packed, c1, c2, stats, tab)
    if tab is not None and not tab_has_c2:
        tab = (tab + c2[None, :]).contiguous()
    k = x.shape[-1]
    m = x.numel() // k
    assert x.is_contiguous() and w_packed.shape[2] == k and n % 320 == 0
    assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.shape[0] == m and stats.shape[2] == 2
    for t in (c1, c2):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n
    assert tab is None or (tab.dtype == torch.float32 and tab.is_contiguous() and tab.shape[-1] == n)
    y = torch.empty(x.shape[:-1] + (n,), dtype=x.dtype, device=x.device)
    rc = lib().im360_linear_ln_fwd(_p(x), _p(w_packed), _p(c1), _p(c2), _p(stats), stats.shape[1], float(eps), _p(tab),
                                   tab_div, tab.shape[0] if tab is not None else 1, _p(y), m, k, n, _dt(x), _stream())
    _check(rc, "im360_linear_ln_fwd")
    _count("gemm", 2.0 * m * k * n, x.element_size() * (x.numel() + n * k + y.numel()) + 4 * stats.numel(), shape=f"lin_ln:{k}:{n}")
    return y


def linear_geglu_ln(x, w_packed, c1, c2, stats, eps, inner):
    """GEGLU(LayerNorm(x)) in one launch: operands as in ``linear_ln`` with the rows of gamma (.) W, c1, c2 in the
    interleaved order of ``pack_geglu``."""
    _dev(x, w_packed, c1, c2, stats)
    k = x.shape[-1]
    m = x.numel() // k
    assert x.is_contiguous() and w_packed.shape[2] == k and w_packed.shape[0] >= 2 * inner
    assert stats.dtype == torch.float32 and stats.is_contiguous() and stats.shape[0] == m and stats.shape[2] == 2
    for t in (c1, c2):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == 2 * inner
    y = torch.empty(x.shape[:-1] + (inner,), dtype=x.dtype, device=x.device)
    rc = lib().im360_linear_geglu_ln(_p(x), _p(w_packed), _p(c1), _p(c2), _p(stats), stats.shape[1], float(eps), _p(y),
                                     m, k, inner, _dt(x), _stream())
    _check(rc, "im360_linear_geglu_ln")
    _count("gemm", 2.0 * m * k * 2 * inner, x.element_size() * (x.numel() + 2 * inner * k + y.numel()) + 4 * stats.numel(),
           shape=f"geglu_ln:{k}:{inner}")
    return y


def fold_layer_norm(weight, bias, gamma, beta):
    """Operands of the LayerNorm-folded GEMMs from a Linear (weight [n, K], bias [n] or None) and the LayerNorm in front
    of it: (gamma (.) W rounded to the weight dtype, c1 = its row sums, c2 = W beta + bias), the vectors in fp32.  c1 is
    taken from the ROUNDED matrix so that the mean term cancels exactly what the MFMA accumulates."""
    wg = (weight.detach().float() * gamma.detach().float()[None, :]).to(weight.dtype)
    c1 = wg.float().sum(dim=1)
    c2 = weight.detach().float() @ beta.detach().float()
    if bias is not None:
        c2 = c2 + bias.detach().float()
    return wg, c1.contiguous(), c2.contiguous()


# ------------------------------------------------------------------------------------------ token-wise
def layer_norm(x, gamma, beta, eps=1e-5, pre=None, post=None, post_div=1):
    """LayerNorm over the last dim of x [..., C] (rows flattened).  ``pre`` [P, C]: added to row r as
    pre[r % P] before normalising; ``post`` [Q, C]: post[(r // post_div) % Q] added to the result."""
    _dev(x, gamma, beta, pre, post)
    C = x.shape[-1]
    assert x.is_contiguous() and gamma.dtype == x.dtype
    rows = x.numel() // C
    y = torch.empty_like(x)
    for t in (pre, post):
        assert t is None or (t.is_contiguous() and t.shape[-1] == C and t.dtype == x.dtype)
    rc = lib().im360_layernorm(_p(x), _p(gamma), _p(beta), _p(pre), _p(post), _p(y), rows, C,
                               pre.shape[0] if pre is not None else 1, post_div,
                               post.shape[0] if post is not None else 1, float(eps), _dt(x), _stream())
    _check(rc, "im360_layernorm")
    _count("misc", 0.0, 2 * x.element_size() * x.numel())
    return y


def geglu(h):
    """h [..., 2*I] -> h[..., :I] * gelu(h[..., I:])."""
    _dev(h)
    assert h.is_contiguous()
    I = h.shape[-1] // 2
    out = torch.empty(h.shape[:-1] + (I,), dtype=h.dtype, device=h.device)
    rc = lib().im360_geglu(_p(h), _p(out), h.numel() // (2 * I), I, _dt(h), _stream())
    _check(rc, "im360_geglu")
    _count("misc", 0.0, h.element_size() * (h.numel() + out.numel()))
    return out


def interleave_geglu(weight, bias):
    """Row order of the fused GEGLU GEMM: 32-row blocks of the value half and the gate half alternate, so packed rows
    [64 q, 64 q + 32) are value rows [32 q, 32 q + 32) and packed rows [64 q + 32, 64 q + 64) the matching gate rows."""
    two_i = weight.shape[0]
    i = two_i // 2
    assert i % 32 == 0
    w = torch.stack([weight[:i].reshape(i // 32, 32, -1), weight[i:].reshape(i // 32, 32, -1)], dim=1).reshape(two_i, -1)
    b = None
    if bias is not None:
        b = torch.stack([bias[:i].reshape(i // 32, 32), bias[i:].reshape(i // 32, 32)], dim=1).reshape(two_i).contiguous()
    return w, b


def pack_geglu(weight, bias):
    """GEGLU projection ``weight`` [2I, K] / ``bias`` [2I] (value half, then gate half) -> operands of
    ``linear_geglu``: 32-row blocks of the two halves interleaved (v0, g0, v1, g1, ...), the weight in the conv
    kernel's packed [rows][1][K] layout."""
    two_i, k = weight.shape
    assert (two_i // 2) % 128 == 0 and k % 64 == 0, (two_i, k)
    w, b = interleave_geglu(weight, bias)
    return pack_conv_weight(w.reshape(two_i, k, 1, 1).contiguous()), b


def linear_geglu(x, w_packed, bias_packed, inner):
    """x [..., K] -> (x W_v^T + b_v) * gelu(x W_g^T + b_g) [..., inner] in one launch (operands from ``pack_geglu``)."""
    _dev(x, w_packed, bias_packed)
    assert x.is_contiguous() and w_packed.shape[2] == x.shape[-1] and w_packed.shape[0] >= 2 * inner
    k = x.shape[-1]
    m = x.numel() // k
    y = torch.empty(x.shape[:-1] + (inner,), dtype=x.dtype, device=x.device)
    rc = lib().im360_linear_geglu(_p(x), _p(w_packed), _p(bias_packed), _p(y), m, k, inner, _dt(x), _stream())
    _check(rc, "im360_linear_geglu")
    _count("gemm", 2.0 * m * k * 2 * inner, x.element_size() * (x.numel() + 2 * inner * k + y.numel()), shape=f"geglu:{k}:{inner}")
    return y


def softmax_rows(x, scale=1.0, out=None):
    """softmax(x * scale) over the last dim of a 2-D tensor (fp32 inside); ``out`` may be ``x`` (in place)."""
    _dev(x, out)
    assert x.dim() == 2 and x.stride(1) == 1
    if out is None:
        out = torch.empty_like(x)
    else:
        _written(out)
    rc = lib().im360_softmax_rows(_p(x), _p(out), x.shape[0], x.shape[1], x.stride(0), out.stride(0), float(scale),
                                  _dt(x), _stream())
    _check(rc, "im360_softmax_rows")
    _count("misc", 0.0, 2 * x.element_size() * x.numel())
    return out


def can_single_head_attention(d, nk):
    """Shapes ``single_head_attention`` takes (GEMM K / softmax row granularity); callers keep a torch path for the rest."""
    return d % 32 == 0 and nk % 32 == 0


def single_head_attention(q, k, v, scale):
    """softmax(q k^T * scale) v for ONE head of any width (the VAE's d = 512 AttentionBlock): scores through the MFMA
    GEMM kernel (k as the weight operand), fp32 row softmax of the 16-bit scores, probabilities x v through the GEMM
    kernel again -- the reference's own op order (baddbmm -> softmax(float) -> bmm, diffusers/models/attention.py:
    336-364).  q [Nq, d], k / v [Nk, d]; d % 32 == 0, Nk % 32 == 0."""
    nq, d = q.shape
    nk = k.shape[0]
    if not can_single_head_attention(d, nk):
        raise NotImplementedError(f"single_head_attention: d={d} and Nk={nk} must be multiples of 32")
    kw = k.contiguous().reshape(nk, 1, d) if nk % 128 == 0 else pack_conv_weight(k.reshape(nk, d, 1, 1))
    # the scale goes into q BEFORE the GEMM, so the 16-bit scores that are stored are the scaled ones -- like the reference's
    # baddbmm(alpha=scale); unscaled d = 512 dot products would cost fp16 a factor sqrt(512) of headroom before inf
    s = conv2d((q * scale).contiguous().reshape(nq, 1, 1, d), kw, nk).reshape(nq, nk)
    softmax_rows(s, 1.0, out=s)
    vt = v.t().contiguous()
    vw = vt.reshape(d, 1, nk) if d % 128 == 0 else pack_conv_weight(vt.reshape(d, nk, 1, 1))
    return conv2d(s.reshape(nq, 1, 1, nk), vw, d).reshape(nq, d)


# ------------------------------------------------------------------------------------------ misc
def circular_pad_w(x, pad):
    """x [..., W, C] channels-last -> [..., W + 2 pad, C]."""
    _dev(x)
    assert x.is_contiguous()
    W, C = x.shape[-2], x.shape[-1]
    rows = x.numel() // (W * C)
    y = torch.empty(x.shape[:-2] + (W + 2 * pad, C), dtype=x.dtype, device=x.device)
    rc = lib().im360_circular_pad_w(_p(x), _p(y), rows, W, C, pad, _dt(x), _stream())
    _check(rc, "im360_circular_pad_w")
    return y


def circular_pad_hw(x, left, right, top=0, bottom=0):
    """``F.pad(x, (left, right, top, bottom), mode="circular")`` on the last two axes of a contiguous W-last tensor of
    any dtype (one HBM pass)."""
    _dev(x)
    assert x.is_contiguous() and x.dim() >= 2
    H, W = x.shape[-2], x.shape[-1]
    n = x.numel() // (H * W)
    y = torch.empty(x.shape[:-2] + (H + top + bottom, W + left + right), dtype=x.dtype, device=x.device)
    rc = lib().im360_circular_pad_hw(_p(x), _p(y), n, H, W, left, right, top, bottom, x.element_size(), _stream())
    _check(rc, "im360_circular_pad_hw")
    return y


def remap_cubic_wrap(img, map_x, map_y, wtab):
    """``cv2.remap(img[n], map_x[m], map_y[m], INTER_CUBIC, BORDER_WRAP)`` for every (n, m): img uint8 [N, H, W, C] on the
    device, maps float32 [M, h, w], wtab int16 [1024, 16] -> uint8 [N, M, h, w, C]."""
    _dev(img, map_x, map_y, wtab)
    assert img.dtype == torch.uint8 and img.dim() == 4 and img.is_contiguous()
    assert map_x.dtype == torch.float32 and map_x.shape == map_y.shape and map_x.dim() == 3 and map_x.is_contiguous() and map_y.is_contiguous()
    assert wtab.dtype == torch.int16 and wtab.shape == (1024, 16) and wtab.is_contiguous()
    N, H, W, C = img.shape
    M, h, w = map_x.shape
    out = torch.empty((N, M, h, w, C), dtype=torch.uint8, device=img.device)
    rc = lib().im360_remap_cubic_wrap_u8(_p(img), _p(map_x), _p(map_y), _p(wtab), _p(out), N, M, H, W, C, h, w, _stream())
    _check(rc, "im360_remap_cubic_wrap_u8")
    return out


def max_rect(mask):
    """Largest all-ones rectangle of a host mask [H, W] -> (top, left, width, height) (reference scan order / ties)."""
    import numpy as np
    m = np.ascontiguousarray(np.asarray(mask) == 1, dtype=np.uint8)
    rect = (ctypes.c_int64 * 4)()
    rc = lib().im360_max_rect(m.ctypes.data, m.shape[0], m.shape[1], ctypes.addressof(rect))
    _check(rc, "im360_max_rect")
    return tuple(int(v) for v in rect)


def cfg_ddim_update(uncond, cond, sample, guidance, cx, cv, coef_dev=None):
    """x_prev = cx * sample + cv * (uncond + guidance * (cond - uncond)); ``coef_dev`` = device float32[3]
    (guidance, cx, cv) read by the kernel instead of the scalars (graph replay)."""
    _dev(uncond, cond, sample)
    assert uncond.is_contiguous() and cond.is_contiguous() and sample.is_contiguous()
    assert uncond.shape == cond.shape == sample.shape and uncond.dtype == sample.dtype
    out = torch.empty_like(sample)
    rc = lib().im360_cfg_ddim_update(_p(uncond), _p(cond), _p(sample), _p(out), sample.numel(),
                                     float(guidance), float(cx), float(cv), _dt(sample), _stream(), _p(coef_dev))
    _check(rc, "im360_cfg_ddim_update")
    return out


# ------------------------------------------------------------------------------------------ tuning knobs
KNOBS = {"attn_qb": 0, "conv_big": 1, "conv_bk": 2, "tattn_scalar": 3, "conv_ring": 4, "attn_hl": 5, "conv_dbg": 6, "conv_halo": 7, "conv_cm": 8, "ln_packed": 9,
         "ring_groups": 10, "attn_x": 11, "attn_ds": 12, "attn_one": 13, "attn_dbg": 14, "attn_hg": 15, "conv_small": 16, "attn_w3": 17, "attn_pipe": 18, "conv_stag": 19, "conv_persist": 20, "gn_apply": 21, "tattn_nt": 22, "nt": 23, "g4": 24, "gn_wgs": 25, "conv_ksplit": 26}


ATTN_PIPE_DEFAULT = -1         # the library's default for the attn_pipe knob (abi.cpp)


def ablate_build():
    """True when the library was built with `make ablate` (rejected A/B variants and ablation kernels compiled in)."""
    return bool(lib().im360_build_flags() & 1)


def tuning_set(name, value):
    """A/B switches of the launchers (tools/bench_kernels.py).  Defaults are the measured best, see DESIGN.md section 3b';
    conv_halo / conv_cm change the fp32 summation order, attn_x (0 against 1 - 3) and attn_ds the 16-bit rounding points,
    conv_dbg / attn_dbg break results on purpose, the others do not change results."""
    _check(lib().im360_tuning_set(KNOBS[name], int(value)), "im360_tuning_set")


# ------------------------------------------------------------------------------------------ profiling
def prof_enable(kinds):
    mask = 0
    for k in kinds:
        mask |= 1 << PROF_KINDS[k]
    lib().im360_prof_enable(mask)


def prof_collect(kind):
    ms, n = ctypes.c_double(0.0), ctypes.c_long(0)
    lib().im360_prof_collect(PROF_KINDS[kind], ctypes.byref(ms), ctypes.byref(n))
    return ms.value, n.value
