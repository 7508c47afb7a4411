"""Functional fp32 restatement of the AnimateDiff-derived UNet3D building blocks.

TEST INFRASTRUCTURE (see package docstring).  Tensors use the reference's layout
``[b, c, f, h, w]``; ``sd`` is a flat state dict, ``p`` a key prefix ending in '.'.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- storage emulation (calibration only)
# None: plain fp32 (the oracle proper, what every golden fixture was generated with).  A 16-bit dtype: the OUTPUT of every
# primitive below (Linear, LayerNorm, GroupNorm, conv, attention) is rounded to that dtype and back while the arithmetic
# stays fp32 -- a model of "16-bit storage, fp32 accumulation" that contains no kernel at all.  Tests use it to calibrate
# their tolerances: the error of this emulation against the fp32 oracle is what storage rounding ALONE costs on a given
# network, so a product error far above it points at a kernel, not at the number format.
STORE_DTYPE = None


def _st(x):
    return x if STORE_DTYPE is None else x.to(STORE_DTYPE).float()


class storage:
    """``with storage(torch.bfloat16): ...`` -- see STORE_DTYPE."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global STORE_DTYPE
        self.saved, STORE_DTYPE = STORE_DTYPE, self.dtype

    def __exit__(self, *a):
        global STORE_DTYPE
        STORE_DTYPE = self.saved
        return False


# ----------------------------------------------------------------------------- primitives
def linear(sd, p, x):
    return _st(F.linear(x, sd[p + "weight"], sd.get(p + "bias")))


def layer_norm(sd, p, x, eps=1e-5):
    w = sd[p + "weight"]
    return _st(F.layer_norm(x, (w.shape[0],), w, sd[p + "bias"], eps))


def conv2d_frames(sd, p, x, stride=1, padding=1):
    """InflatedConv3d = nn.Conv2d on (b f) c h w  (animatediff/models/resnet.py:19-27)."""
    b, c, f, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    y = _st(F.conv2d(y, sd[p + "weight"], sd.get(p + "bias"), stride=stride, padding=padding))
    return y.reshape(b, f, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def group_norm_frames(sd, p, x, groups, eps):
    """InflatedGroupNorm: per-(b f) statistics (animatediff/models/resnet.py:9-17)."""
    b, c, f, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    y = _st(F.group_norm(y, groups, sd[p + "weight"], sd[p + "bias"], eps))
    return y.reshape(b, f, c, h, w).permute(0, 2, 1, 3, 4)


def sdpa(q, k, v, heads, bias=None, scale=None):
    """softmax(q k^T * scale + bias) v on [B, N, heads*d], scale defaults to d^-1/2
    (diffusers/models/attention_processor.py:562-591, 636-643)."""
    B, Nq, C = q.shape
    d = C // heads
    qh = q.reshape(B, Nq, heads, d).transpose(1, 2)
    kh = k.reshape(B, -1, heads, d).transpose(1, 2)
    vh = v.reshape(B, -1, heads, d).transpose(1, 2)
    sc = d ** -0.5 if scale is None else scale

    def one(qh, kh, vh, bias):
        s = torch.matmul(qh, kh.transpose(-1, -2)) * sc
        if bias is not None:
            s = s + bias
        return torch.matmul(s.softmax(-1), vh)
    # (batch, head) chunks that keep the score matrix under ~2 GB (cfg2-sized fixtures: 8192^2 and 8192 x 20480 score
    # matrices); the entries are independent, so the numbers do not change
    Nk = kh.shape[2]
    per = max(1, int(2e9 // (4 * Nq * Nk)))
    if per >= B * heads or (bias is not None and bias.dim() > 2):
        o = one(qh, kh, vh, bias)
    else:
        qf, kf, vf = qh.reshape(B * heads, Nq, d), kh.reshape(B * heads, Nk, d), vh.reshape(B * heads, Nk, d)
        o = torch.empty(B * heads, Nq, d, dtype=qh.dtype)
        for i in range(0, B * heads, per):
            o[i:i + per] = one(qf[i:i + per], kf[i:i + per], vf[i:i + per], bias)
        o = o.reshape(B, heads, Nq, d)
    return _st(o.transpose(1, 2).reshape(B, Nq, C))


def geglu_ff(sd, p, x):
    """FeedForward(GEGLU) (diffusers/models/attention_lora.py:493-547, activations.py:93-125)."""
    h = linear(sd, p + "net.0.proj.", x)
    a, gate = h.chunk(2, dim=-1)
    return linear(sd, p + "net.2.", a * F.gelu(gate))


# ----------------------------------------------------------------------------- embeddings
def timestep_sincos(t, dim, flip_sin_to_cos=True, shift=0.0):
    """get_timestep_embedding (diffusers/models/embeddings.py:26-66)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / (half - shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([emb.sin(), emb.cos()], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def timestep_mlp(sd, p, x):
    """TimestepEmbedding: Linear -> SiLU -> Linear (embeddings.py:190-236)."""
    return linear(sd, p + "linear_2.", F.silu(linear(sd, p + "linear_1.", x)))


# ----------------------------------------------------------------------------- resnet / samplers
def resnet_block(sd, p, x, temb, groups=32, eps=1e-5):
    """ResnetBlock3D.forward (animatediff/models/resnet.py:221-254)."""
    h = F.silu(group_norm_frames(sd, p + "norm1.", x, groups, eps))
    h = conv2d_frames(sd, p + "conv1.", h)
    if temb is not None:
        h = h + linear(sd, p + "time_emb_proj.", F.silu(temb))[:, :, None, None, None]
    h = F.silu(group_norm_frames(sd, p + "norm2.", h, groups, eps))
    h = conv2d_frames(sd, p + "conv2.", h)
    if (p + "conv_shortcut.weight") in sd:
        x = conv2d_frames(sd, p + "conv_shortcut.", x, padding=0)
    return _st(x + h)


def downsample(sd, p, x):
    """Downsample3D: conv3x3 stride 2 pad 1 (resnet.py:117-140)."""
    return conv2d_frames(sd, p + "conv.", x, stride=2, padding=1)


def upsample(sd, p, x):
    """Upsample3D: nearest x2 on (h, w) then conv3x3 (resnet.py:71-114)."""
    x = F.interpolate(x, scale_factor=(1.0, 2.0, 2.0), mode="nearest")
    return conv2d_frames(sd, p + "conv.", x)


# ----------------------------------------------------------------------------- spatial transformer
def spatial_transformer(sd, p, x, ctx, heads, num_tokens, groups=32, xformers=False):
    """Transformer3DModel.forward + BasicTransformerBlock.forward + IPCrossAttention.forward
    (animatediff/models/attention.py:246-301, 461-508, 65-156); use_linear_projection=True.

    Reference quirk, pinned by running the reference here: IPCrossAttention.__init__ overwrites
    ``self.scale`` (the d^-1/2 logit scale of its Attention base class) with the IP-adapter
    scale 1.0 (attention.py:50,62), so the non-xformers path ``_attention`` (used on CPU) runs
    both cross attentions with logit scale 1.0, while the xformers path
    (attention_processor.py:636-643) uses xformers' default d^-1/2.  ``xformers`` selects which
    of the two reference behaviours is restated."""
    b, c, f, h, w = x.shape
    res = x
    y = group_norm_frames(sd, p + "norm.", x, groups, 1e-6)
    y = y.permute(0, 2, 3, 4, 1).reshape(b * f, h * w, c)
    y = linear(sd, p + "proj_in.", y)
    ctx = ctx.repeat_interleave(f, dim=0)                      # 'b n c -> (b f) n c'
    tb = p + "transformer_blocks.0."
    # self attention
    n = layer_norm(sd, tb + "norm1.", y)
    a = sdpa(linear(sd, tb + "attn1.to_q.", n), linear(sd, tb + "attn1.to_k.", n),
             linear(sd, tb + "attn1.to_v.", n), heads)
    y = _st(linear(sd, tb + "attn1.to_out.0.", a) + y)          # (the 16-bit residual stream is re-rounded by every add)
    # text + IP cross attention sharing the query, scale 1.0
    n = layer_norm(sd, tb + "norm2.", y)
    end = ctx.shape[1] - num_tokens
    text, ip = ctx[:, :end], ctx[:, end:]
    q = linear(sd, tb + "attn2.to_q.", n)
    cs = None if xformers else 1.0
    a = sdpa(q, linear(sd, tb + "attn2.to_k.", text), linear(sd, tb + "attn2.to_v.", text), heads, scale=cs)
    a = a + 1.0 * sdpa(q, linear(sd, tb + "attn2.to_k_ip.", ip), linear(sd, tb + "attn2.to_v_ip.", ip), heads,
                       scale=cs)
    y = _st(linear(sd, tb + "attn2.to_out.0.", a) + y)
    # feed forward
    y = _st(geglu_ff(sd, tb + "ff.", layer_norm(sd, tb + "norm3.", y)) + y)
    y = linear(sd, p + "proj_out.", y)
    y = y.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3)
    return _st(y + res)


# ----------------------------------------------------------------------------- motion module
def temporal_pe(d_model, length):
    """PositionalEncoding buffer (animatediff/models/motion_module.py:262-280)."""
    position = torch.arange(length).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(length, d_model)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def motion_module(sd, p, x, heads=8, groups=32):
    """VanillaTemporalModule -> TemporalTransformer3DModel -> TemporalTransformerBlock ->
    VersatileAttention (motion_module.py:52-96, 158-185, 247-259, 343-429)."""
    p = p + "temporal_transformer."
    b, c, f, h, w = x.shape
    res = x
    y = group_norm_frames(sd, p + "norm.", x, groups, 1e-6)
    y = y.permute(0, 2, 3, 4, 1).reshape(b * f, h * w, c)
    y = linear(sd, p + "proj_in.", y)
    tb = p + "transformer_blocks.0."
    pe = temporal_pe(c, f)
    for i in range(2):                                            # two Temporal_Self blocks
        n = layer_norm(sd, tb + f"norms.{i}.", y)
        t = n.reshape(b, f, h * w, c).permute(0, 2, 1, 3).reshape(b * h * w, f, c)   # (b f) d c -> (b d) f c
        t = t + pe[None]
        ab = tb + f"attention_blocks.{i}."
        a = sdpa(linear(sd, ab + "to_q.", t), linear(sd, ab + "to_k.", t), linear(sd, ab + "to_v.", t), heads)
        a = linear(sd, ab + "to_out.0.", a)
        a = a.reshape(b, h * w, f, c).permute(0, 2, 1, 3).reshape(b * f, h * w, c)
        y = _st(a + y)
    y = _st(geglu_ff(sd, tb + "ff.", layer_norm(sd, tb + "ff_norm.", y)) + y)
    y = linear(sd, p + "proj_out.", y)
    y = y.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3)
    return _st(y + res)


# ----------------------------------------------------------------------------- IP adapter
def _ln_ff_nobias(sd, p, x):
    """resampler.FeedForward: LN -> Linear(no bias) -> GELU -> Linear(no bias)
    (animatediff/models/resampler.py:15-22)."""
    h = layer_norm(sd, p + "0.", x)
    h = F.linear(h, sd[p + "1.weight"])
    return F.linear(F.gelu(h), sd[p + "3.weight"])


def _temporal_self_attn(sd, pa, pn, x, heads):
    """x: [b, f, d, c]; attention over f per (b d) (resampler.py:238-241, 254-258)."""
    b, f, d, c = x.shape
    t = x.permute(0, 2, 1, 3).reshape(b * d, f, c)
    n = layer_norm(sd, pn, t)
    a = sdpa(linear(sd, pa + "to_q.", n), linear(sd, pa + "to_k.", n), linear(sd, pa + "to_v.", n), heads)
    t = linear(sd, pa + "to_out.0.", a) + t
    return t.reshape(b, d, f, c).permute(0, 2, 1, 3)


def _avgpool_frames(x, k=4):
    b, f, d, c = x.shape
    t = x.permute(0, 2, 3, 1).reshape(b * d, c, f)
    t = F.avg_pool1d(t, kernel_size=k)
    return t.reshape(b, d, c, -1).permute(0, 3, 1, 2)


def temporal_projection(sd, p, x, heads=8):
    """TemporalProjection.forward with spacial_compress + compress_video_features
    (animatediff/models/resampler.py:231-267).  x: [b, f, 4096, 256] -> [b, f/16, 256, 1024]."""
    b, f, d, c = x.shape
    s = int(math.sqrt(d))
    y = x.reshape(b * f, s, s, c).permute(0, 3, 1, 2)
    y = F.conv2d(y, sd[p + "patch_embed.weight"], sd[p + "patch_embed.bias"], stride=4)
    y = y.permute(0, 2, 3, 1).reshape(b, f, -1, y.shape[1])
    y = _temporal_self_attn(sd, p + "attn_temp.", p + "norm_temp.", y, heads)
    y = _ln_ff_nobias(sd, p + "ff.", layer_norm(sd, p + "norm1.", y)) + y
    y = _avgpool_frames(y)
    y = _temporal_self_attn(sd, p + "attn_temp_2.", p + "norm_temp_2.", y, heads)
    y = _ln_ff_nobias(sd, p + "ff_2.", layer_norm(sd, p + "norm2.", y)) + y
    y = _avgpool_frames(y)
    return y


def resampler(sd, p, x, heads=12, depth=4):
    """Resampler.forward + PerceiverAttention.forward (resampler.py:132-160, 57-80)."""
    lat = sd[p + "latents"].repeat(x.shape[0], 1, 1)
    x = linear(sd, p + "proj_in.", x)
    for i in range(depth):
        pa = p + f"layers.{i}.0."
        xn = layer_norm(sd, pa + "norm1.", x)
        ln = layer_norm(sd, pa + "norm2.", lat)
        q = F.linear(ln, sd[pa + "to_q.weight"])
        kv = F.linear(torch.cat([xn, ln], dim=1), sd[pa + "to_kv.weight"])
        k, v = kv.chunk(2, dim=-1)
        a = sdpa(q, k, v, heads)             # (q s)(k s)^T with s = d^-1/4  ==  q k^T d^-1/2
        lat = F.linear(a, sd[pa + "to_out.weight"]) + lat
        lat = _ln_ff_nobias(sd, p + f"layers.{i}.1.", lat) + lat
    lat = linear(sd, p + "proj_out.", lat)
    return layer_norm(sd, p + "norm_out.", lat)
