"""AnimateDiff-derived UNet3DConditionModel of Imagine360, rebuilt channels-last on HIP kernels.

Mirrors the reference's class names, constructor keywords, attribute tree and state-dict keys
(animatediff/models/{unet,unet_blocks,attention,motion_module,resnet,resampler}.py) so reference
checkpoints and the LoRA merge by dotted name (inference_dual_p2e.py:175-195) keep working, but the
arithmetic is different by design: activations stay token-major ``[N, H, W, C]`` end to end (no
'b c f h w' <-> '(b f) (h w) c' shuffles), every GroupNorm/SiLU/conv/attention runs in a hand-written
gfx950 kernel, QKV projections are fused, and the pano branch's circular padding is folded into the
kernels' addressing (``pano=True``).
"""
import math
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels
from .layers import (DerivedCache, FeedForward, InflatedConv3d, InflatedGroupNorm, QKVAttention, from_cl, layer_norm,
                     linear, linear_residual, ln_linear,
                     to_cl)


# ----------------------------------------------------------------------------------------------
# embeddings
class Timesteps(nn.Module):
    """Sinusoidal embedding, flip_sin_to_cos, shift 0 (diffusers/models/embeddings.py:26-66, 239-252)."""

    def __init__(self, num_channels, flip_sin_to_cos=True, downscale_freq_shift=0.0):
        super().__init__()
        self.num_channels, self.flip, self.shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, t):
        half = self.num_channels // 2
        freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / (half - self.shift))
        ang = t[:, None].float() * freq[None, :]
        s, c = ang.sin(), ang.cos()
        return torch.cat([c, s], dim=-1) if self.flip else torch.cat([s, c], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, out_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


# ----------------------------------------------------------------------------------------------
# resnet / samplers
class ResnetBlock3D(nn.Module):
    """GN+SiLU -> conv3x3 (+temb) -> GN+SiLU -> conv3x3 (+1x1 shortcut) + residual
    (animatediff/models/resnet.py:143-254).

    ``pano=True`` fuses the reference's pad_pano(2) / unpad_pano(2) around the block
    (src/models/MVGenModel.py:276-281) *including its side effects*: norm1 statistics count the wrapped
    columns twice, conv1 runs on the W+4 wide tensor (zero padding outside it) and norm2 statistics
    are taken over all W+4 columns."""

    def __init__(self, *, in_channels, out_channels=None, temb_channels=512, groups=32, eps=1e-6, **_):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = InflatedGroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = InflatedConv3d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = InflatedGroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = InflatedConv3d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.conv_shortcut = None
        if in_channels != out_channels:
            self.conv_shortcut = InflatedConv3d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward_cl(self, x, temb, frames, pano=False):
        """x [N, H, W, C], or a pair (xa, xb) standing for torch.cat([xa, xb], -1) -- the decoder's skip connections
        (MVGenModel.py:407-437): GroupNorm and the 1x1 shortcut read the two tensors in place, the concatenation is never
        written."""
        pad = 2 if pano else 0
        if isinstance(x, (tuple, list)):
            x = tuple(x)
            xa, xb = x
            if self.conv_shortcut is None or not kernels.can_conv1x1_cat(xa.shape[-1], xb.shape[-1]) or xb.shape[-1] % 8:
                x = torch.cat([xa, xb], dim=-1)
        w = (x[0] if isinstance(x, tuple) else x).shape[2]
        h = self.norm1.forward_cl(x, silu=True, pad=pad)
        t = self.time_emb_proj(F.silu(temb)).contiguous() if (temb is not None and self.time_emb_proj is not None) else None
        # (gn_stats: the convolution's epilogue leaves the GroupNorm partial sums of what it stores for the next norm --
        #  resnet.py:221-243 is norm -> SiLU -> conv twice, and every module that follows opens with a GroupNorm)
        h = self.conv1.forward_cl(h, temb=t, imgs_per_temb=frames, gn_stats=True)
        h = self.norm2.forward_cl(h, silu=True)
        if isinstance(x, tuple):
            short = self.conv_shortcut.forward_cat(*x)
        else:
            short = x if self.conv_shortcut is None else self.conv_shortcut.forward_cl(x)
        return self.conv2.forward_cl(h, x_off=pad, wout=w, res=short, gn_stats=True)

    def forward(self, input_tensor, temb):
        x, f = to_cl(input_tensor)
        return from_cl(self.forward_cl(x, temb, f), f)


class Downsample3D(nn.Module):
    """conv3x3 stride 2 (resnet.py:117-140); ``pano``: pad 2 -> conv -> unpad 1 == circular W."""

    def __init__(self, channels, use_conv=True, out_channels=None, padding=1, name="conv"):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.conv = InflatedConv3d(channels, self.out_channels, 3, stride=2, padding=padding)

    def forward_cl(self, x, pano=False):
        # (the next ResnetBlock's norm1 reads this tensor: its GroupNorm partial sums come out of the epilogue; the panorama
        #  branch normalises the PADDED tensor there -- edge columns weigh twice -- and takes its own statistics pass)
        return self.conv.forward_cl(x, wrap=pano, gn_stats=not pano)

    def forward(self, hidden_states):
        x, f = to_cl(hidden_states)
        return from_cl(self.forward_cl(x), f)


class Upsample3D(nn.Module):
    """nearest x2 + conv3x3 (resnet.py:71-114), the upsample folded into the conv's input index;
    ``pano``: pad 1 -> up -> conv -> unpad 2 == circular W on the upsampled grid."""

    def __init__(self, channels, use_conv=True, use_conv_transpose=False, out_channels=None, name="conv"):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.conv = InflatedConv3d(channels, self.out_channels, 3, padding=1)

    def forward_cl(self, x, pano=False):
        return self.conv.forward_cl(x, up=True, wrap=pano)

    def forward(self, hidden_states, output_size=None):
        x, f = to_cl(hidden_states)
        return from_cl(self.forward_cl(x), f)


# ----------------------------------------------------------------------------------------------
# spatial transformer
class IPCrossAttention(QKVAttention):
    """Text + IP-adapter cross attention sharing one query (animatediff/models/attention.py:23-156).

    Reference quirk kept switchable: the class overwrites the logit scale d^-1/2 with the adapter
    scale 1.0 (attention.py:50,62), so its non-xformers path runs with logit scale 1.0 while the
    xformers path (the shipped config) uses d^-1/2.  ``_use_memory_efficient_attention_xformers``
    (set by ``enable_xformers_memory_efficient_attention``) selects which one is reproduced; both run
    the same HIP kernel."""

    def __init__(self, query_dim, cross_attention_dim, image_cross_attention_dim, heads, dim_head, scale=1.0,
                 num_tokens=4):
        super().__init__(query_dim, cross_attention_dim, heads, dim_head)
        self.scale, self.num_tokens = scale, num_tokens
        self.image_cross_attention_dim, self.cross_attention_dim = image_cross_attention_dim, cross_attention_dim
        self.to_k_ip = nn.Linear(image_cross_attention_dim or query_dim, query_dim, bias=False)
        self.to_v_ip = nn.Linear(image_cross_attention_dim or query_dim, query_dim, bias=False)

    def project_context(self, ctx):
        """ctx [B, n_text + num_tokens, d] -> (k_text, v_text, k_ip, v_ip); once per video, shared by frames."""
        end = ctx.shape[1] - self.num_tokens
        text, ip = ctx[:, :end], ctx[:, end:]
        if self.image_cross_attention_dim != self.cross_attention_dim:
            ip = ip[:, :, :self.image_cross_attention_dim]
        return self.to_k(text), self.to_v(text), self.to_k_ip(ip), self.to_v_ip(ip)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, frames=None, residual=None,
                norm=None, stats=None, row_stats=False):
        """hidden_states [(b f), n, c]; encoder_hidden_states [(b f), 141, d] (reference form) or [b, 141, d]
        with ``frames`` given (one context per video).  ``residual`` is added to the result (fused into the output
        projection).  ``norm``: the LayerNorm in front of the block, applied to hidden_states here (folded into the query
        projection when the rows' statistics ``stats`` are given); ``row_stats``: return (out, statistics of out's rows)."""
        kv_group = 1
        if frames is not None and encoder_hidden_states.shape[0] * frames == hidden_states.shape[0]:
            kv_group = frames
        kt, vt, ki, vi = self.project_context(encoder_hidden_states)
        if norm is None:
            q = linear(self.to_q, hidden_states)
        else:
            q = ln_linear(norm, self.to_q.weight, self.to_q.bias, hidden_states, stats, self._derived, "to_q_packed")
        ls = self.dim_head ** -0.5 if self._use_memory_efficient_attention_xformers else self.scale
        if self.dim_head == 64:
            # both key / value sets in one launch: Q is read once, nothing is accumulated through HBM
            out = kernels.attention2(q, kt, vt, ki, vi, self.heads, scale=ls, out_scale=1.0, out_scale2=self.scale, kv_group=kv_group)
        else:
            out = kernels.attention(q, kt, vt, self.heads, scale=ls, kv_group=kv_group)
            kernels.attention(q, ki, vi, self.heads, scale=ls, kv_group=kv_group, out=out, accumulate=True,
                              out_scale=self.scale)
        return self.out_proj(out, residual, row_stats=row_stats)


class BasicTransformerBlock(nn.Module):
    """LN -> self-attn, LN -> text+IP cross-attn, LN -> GEGLU FF (attention.py:323-508)."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, cross_attention_dim, image_cross_attention_dim,
                 scale=1.0, num_tokens=4):
        super().__init__()
        self.attn1 = QKVAttention(dim, None, num_attention_heads, attention_head_dim)
        self.norm1 = nn.LayerNorm(dim)
        self.attn2 = IPCrossAttention(dim, cross_attention_dim, image_cross_attention_dim, num_attention_heads,
                                      attention_head_dim, scale=scale, num_tokens=num_tokens)
        self.norm2 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)
        self.norm3 = nn.LayerNorm(dim)

    def set_use_memory_efficient_attention_xformers(self, flag):
        self.attn1._use_memory_efficient_attention_xformers = flag
        self.attn2._use_memory_efficient_attention_xformers = flag

    def forward(self, hidden_states, encoder_hidden_states=None, frames=None, stats=None, **_):
        """``stats``: LayerNorm statistics of hidden_states' rows when its producer wrote them (layers.gemm_linear).  Each
        of the three LayerNorms is folded into the GEMM that consumes it whenever the statistics came with the rows; the
        output projections (+ residual) write the statistics for the next one."""
        y = hidden_states
        a = self.attn1.self_attention(y, norm=self.norm1, stats=stats)
        y, st = self.attn1.out_proj(a, residual=y, row_stats=True)
        y, st = self.attn2(y, encoder_hidden_states, frames=frames, residual=y, norm=self.norm2, stats=st, row_stats=True)
        return self.ff(y, residual=y, ln=self.norm3, stats=st)


@dataclass
class Transformer3DModelOutput:
    sample: torch.Tensor


class Transformer3DModel(nn.Module):
    """GN -> Linear -> transformer block -> Linear -> + residual, per frame (attention.py:170-301)."""

    def __init__(self, num_attention_heads, attention_head_dim, in_channels, cross_attention_dim,
                 image_cross_attention_dim=1024, norm_num_groups=32, scale=1.0, num_tokens=4, **_):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.norm = InflatedGroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(
            inner, num_attention_heads, attention_head_dim, cross_attention_dim, image_cross_attention_dim,
            scale=scale, num_tokens=num_tokens)])
        self.proj_out = nn.Linear(in_channels, inner)

    def forward_cl(self, x, ctx, frames):
        """x [N, H, W, C]; ctx [N / frames, n_ctx, d] (one context per video)."""
        n, h, w, c = x.shape
        y = self.norm.forward_cl(x).reshape(n, h * w, c)
        y, st = linear(self.proj_in, y, row_stats=True)
        for blk in self.transformer_blocks:
            y = blk(y, ctx, frames=frames, stats=st)
            st = None
        out = linear_residual(self.proj_out, y, x.reshape(n, h * w, c), gn_hw=h * w)
        return kernels.carry_gn(out.reshape(n, h, w, c), out)

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, return_dict=True):
        x, f = to_cl(hidden_states)
        out = from_cl(self.forward_cl(x, encoder_hidden_states, f), f)
        return Transformer3DModelOutput(sample=out) if return_dict else (out,)


# ----------------------------------------------------------------------------------------------
# motion module
class PositionalEncoding(nn.Module):
    """Sinusoidal frame-index table ``pe`` [1, max_len, d] (motion_module.py:262-280)."""

    def __init__(self, d_model, dropout=0.0, max_len=24):
        super().__init__()
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)


class VersatileAttention(QKVAttention):
    """Temporal self-attention over frames per (batch, pixel, head) (motion_module.py:304-429): add the
    frame PE, one fused QKV GEMM, then the strided HIP kernel -- no '(b f) d c -> (b d) f c' copies."""

    def __init__(self, query_dim, heads, dim_head, temporal_position_encoding=True,
                 temporal_position_encoding_max_len=24):
        super().__init__(query_dim, None, heads, dim_head)
        self.pos_encoder = PositionalEncoding(query_dim, max_len=temporal_position_encoding_max_len) \
            if temporal_position_encoding else None
        self.is_cross_attention = False
        self.frame_shard = None          # imagine360_amd.dist.FrameShard when frames are split across GPUs

    def frame_pe(self, frames, dtype, f0=None):
        """PE rows of ``frames`` frames starting at ``f0`` (default: this rank's first frame) [frames, C] in the activation
        dtype (cached), or None."""
        if self.pos_encoder is None:
            return None
        if f0 is None:
            f0 = self.frame_shard.f0 if self.frame_shard is not None else 0
        # (the reference's `x + pe[:, :x.size(1)]` raises on a clip longer than the table; a short slice here would make the
        #  kernels wrap frame positions onto wrong rows -- ADVICE r5)
        if f0 + frames > self.pos_encoder.pe.shape[1]:
            raise ValueError(f"frames {f0} .. {f0 + frames - 1} exceed temporal_position_encoding_max_len = {self.pos_encoder.pe.shape[1]}")
        key = (f0, frames, dtype, self.pos_encoder.pe.device)
        if getattr(self, "_pe_key", None) != key:
            self._pe_key, self._pe_val = key, self.pos_encoder.pe[0, f0:f0 + frames].to(dtype).contiguous()
        return self._pe_val

    def forward(self, tokens, batch, frames, pixels, residual=None, norm=None, stats=None, row_stats=False, frame_major=False):
        """tokens [batch*frames*pixels, C] token-major.  ``norm`` None: ALREADY normalised and with the frame PE added;
        else the block's LayerNorm, applied here together with the PE add -- both folded into the QKV GEMM when the rows'
        statistics ``stats`` are given (the PE rows go through the projection once, as a per-frame table).  ``frames`` =
        frames held by this rank.  ``residual`` is added to the result inside the output projection; ``row_stats``:
        return (out, statistics of out's rows).
        ``frame_major``: the rows are ordered (frame, batch, pixel) and hold ALL frames of this rank's pixel range -- the
        pixel-sharded layout a frame-sharded motion module works in between its two all-to-alls
        (TemporalTransformer3DModel.forward_cl): the PE row of a token is row // (batch * pixels), no exchange here."""
        c = tokens.shape[-1]
        sh = None if frame_major else self.frame_shard
        if norm is None:
            qkv = self.qkv(tokens)
        elif frame_major:
            qkv = self.qkv_ln(norm, tokens, stats, post=self.frame_pe(frames, tokens.dtype, f0=0), post_div=batch * pixels)
        else:
            qkv = self.qkv_ln(norm, tokens, stats, post=self.frame_pe(frames, tokens.dtype), post_div=pixels)
        if frame_major:
            a = kernels.temporal_attention(qkv, batch, frames, pixels, self.heads, frame_major=True)
        elif sh is None:
            a = kernels.temporal_attention(qkv, batch, frames, pixels, self.heads)
        else:
            # (attention-boundary exchange, ``FrameShard(boundary="attention")``: round 3's form, 4 C per token and attention)
            # frame-sharded -> pixel-sharded (one all-to-all over xGMI; pack = one kernel), attention over ALL frames reading
            # the receive buffer in place and writing the return trip's send buffer, and back (all-to-all + one unpack kernel)
            q = sh.frames_to_pixels(qkv.reshape(batch, frames, pixels, 3 * c))
            pp = sh.pixels_per_rank(pixels)
            a = kernels.temporal_attention(q, batch, sh.total, pp, self.heads, frame_major=True,
                                           out=sh.pixel_result_buffer(qkv, batch, pixels, c))
            a = sh.pixels_to_frames(a, batch, pixels).reshape(-1, c)
        return self.out_proj(a, residual, row_stats=row_stats)


class TemporalTransformerBlock(nn.Module):
    def __init__(self, dim, num_attention_heads, attention_head_dim, attention_block_types, max_len, pe=True):
        super().__init__()
        self.attention_blocks = nn.ModuleList([VersatileAttention(dim, num_attention_heads, attention_head_dim, pe, max_len)
                                               for _ in attention_block_types])
        self.norms = nn.ModuleList([nn.LayerNorm(dim) for _ in attention_block_types])
        self.ff = FeedForward(dim)
        self.ff_norm = nn.LayerNorm(dim)

    def forward(self, y, batch, frames, pixels, stats=None, frame_major=False):
        """``stats``: LayerNorm statistics of y's rows from its producer, or None (see BasicTransformerBlock.forward).
        ``frame_major``: rows ordered (frame, batch, pixel), see VersatileAttention.forward."""
        for attn, norm in zip(self.attention_blocks, self.norms):
            y, stats = attn(y, batch, frames, pixels, residual=y, norm=norm, stats=stats, row_stats=True, frame_major=frame_major)   # LN, + PE[frame], attention
        return self.ff(y, residual=y, ln=self.ff_norm, stats=stats)


class TemporalTransformer3DModel(nn.Module):
    def __init__(self, in_channels, num_attention_heads, attention_head_dim, num_layers, attention_block_types,
                 temporal_position_encoding, temporal_position_encoding_max_len, norm_num_groups=32):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        self.norm = InflatedGroupNorm(norm_num_groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([TemporalTransformerBlock(
            inner, num_attention_heads, attention_head_dim, attention_block_types, temporal_position_encoding_max_len,
            temporal_position_encoding) for _ in range(num_layers)])
        self.proj_out = nn.Linear(in_channels, inner)

    frame_shard = None          # imagine360_amd.dist.FrameShard(boundary="module") when frames are split across GPUs

    def forward_cl(self, x, frames):
        if self.frame_shard is not None:
            return self._forward_pixel_sharded(x, frames, self.frame_shard)
        n, h, w, c = x.shape
        y = self.norm.forward_cl(x).reshape(n * h * w, c)
        y, st = linear(self.proj_in, y, row_stats=True)
        for blk in self.transformer_blocks:
            y = blk(y, n // frames, frames, h * w, stats=st)
            st = None
        out = linear_residual(self.proj_out, y, x.reshape(n * h * w, c), gn_hw=h * w)
        return kernels.carry_gn(out.reshape(n, h, w, c), out)

    def _forward_pixel_sharded(self, x, frames, sh):
        """Frame-sharded module with the exchange at the MODULE boundary.  Everything between the
        module's GroupNorm and its residual add -- proj_in, both (LayerNorm, + frame PE, QKV, attention over frames,
        out-projection + residual), the feed-forward, proj_out: animatediff/models/motion_module.py:158-185, 230-258 --
        acts on ONE pixel across frames or on one token, so it runs on pixel-sharded rows of ALL frames: one C-wide
        all-to-all after the (per-image, hence frame-local) GroupNorm and one C-wide all-to-all in front of the residual
        add = 2 C per token and module, where the exchange around each attention moved 3 C out + C back, twice = 8 C.
        x [b * fl, h, w, c] holds this rank's ``frames`` = fl frames; rows between the exchanges are ordered
        (frame, batch, pixel) over all F frames and this rank's ceil(p / W) pixels (zero rows pad the last rank's range:
        they pass through the per-row ops and are dropped by the unpack).  Rounding differs from the unsharded path in one
        place: proj_out's result is rounded to 16 bits before the residual is added (unsharded: the add is fused into the
        GEMM epilogue and rounds once)."""
        n, h, w, c = x.shape
        b, p = n // frames, h * w
        y = self.norm.forward_cl(x)
        rows = sh.frames_to_pixels(y.reshape(b, frames, p, c))              # [F * b * pp, c]
        pp = sh.pixels_per_rank(p)
        y, st = linear(self.proj_in, rows, row_stats=True)
        for blk in self.transformer_blocks:
            y = blk(y, b, sh.total, pp, stats=st, frame_major=True)
            st = None
        y = linear(self.proj_out, y)
        back = sh.pixels_to_frames(y, b, p)                                  # [b, fl, p, c]
        return back.reshape(n, h, w, c) + x


class VanillaTemporalModule(nn.Module):
    """Motion module (motion_module.py:52-96); ``proj_out`` is zero-initialised like the reference."""

    def __init__(self, in_channels, num_attention_heads=8, num_transformer_block=2,
                 attention_block_types=("Temporal_Self", "Temporal_Self"), temporal_position_encoding=False,
                 temporal_position_encoding_max_len=24, temporal_attention_dim_div=1, zero_initialize=True, **_):
        super().__init__()
        self.temporal_transformer = TemporalTransformer3DModel(
            in_channels, num_attention_heads, in_channels // num_attention_heads // temporal_attention_dim_div,
            num_transformer_block, attention_block_types, temporal_position_encoding, temporal_position_encoding_max_len)
        if zero_initialize:
            for p in self.temporal_transformer.proj_out.parameters():
                nn.init.zeros_(p)

    def forward_cl(self, x, frames):
        return self.temporal_transformer.forward_cl(x, frames)

    def forward(self, input_tensor, temb=None, encoder_hidden_states=None, attention_mask=None, anchor_frame_idx=None):
        x, f = to_cl(input_tensor)
        return from_cl(self.forward_cl(x, f), f)


# ----------------------------------------------------------------------------------------------
# IP adapter: TemporalProjection + Resampler (step-invariant; the dual model hoists them out of the loop)
class _NoBiasFF(nn.Sequential):
    """LayerNorm -> Linear(no bias) -> GELU -> Linear(no bias) (animatediff/models/resampler.py:15-22)."""

    def __init__(self, dim, mult=4):
        super().__init__(nn.LayerNorm(dim), nn.Linear(dim, dim * mult, bias=False), nn.GELU(),
                         nn.Linear(dim * mult, dim, bias=False))


class TemporalProjection(nn.Module):
    """SAM features [b, f, 4096, 256] -> 4x4 patch embed -> temporal attn/FF -> avgpool 4 -> again -> avgpool 4
    (resampler.py:194-267)."""

    def __init__(self, *, dim, dim_head=64, heads=8, compress_video_features=False, kernel_size=4):
        super().__init__()
        self.compress_video_features = compress_video_features
        self.spacial_compress = dim < 1024
        d = dim * 4 if self.spacial_compress else dim
        if self.spacial_compress:
            self.patch_embed = nn.Conv2d(dim, dim * 4, kernel_size=4, stride=4, bias=True)
        self.attn_temp = QKVAttention(d, None, heads, dim_head)
        self.norm_temp = nn.LayerNorm(d)
        self.ff = _NoBiasFF(d)
        self.norm1 = nn.LayerNorm(d)
        self.kernel_size = kernel_size
        if compress_video_features:
            self.attn_temp_2 = QKVAttention(d, None, heads, dim_head)
            self.norm_temp_2 = nn.LayerNorm(d)
            self.ff_2 = _NoBiasFF(d)
            self.norm2 = nn.LayerNorm(d)

    def _attn(self, attn, norm, y):
        b, f, d, c = y.shape
        a = kernels.temporal_attention(attn.qkv(norm(y).reshape(-1, c)), b, f, d, attn.heads)
        return attn.out_proj(a).reshape(b, f, d, c) + y

    def _pool(self, y):
        b, f, d, c = y.shape
        k = self.kernel_size
        fo = f // k
        return y[:, :fo * k].reshape(b, fo, k, d, c).mean(dim=2)

    def forward(self, x):
        b, f, d, c = x.shape
        if self.spacial_compress:
            s = int(math.sqrt(d))
            # 4x4 stride-4 conv == Linear over (ky, kx, c) patches
            p = x.reshape(b * f, s // 4, 4, s // 4, 4, c).permute(0, 1, 3, 2, 4, 5).reshape(b, f, (s // 4) ** 2, 16 * c)
            w = self.patch_embed.weight.permute(0, 2, 3, 1).reshape(self.patch_embed.out_channels, 16 * c)
            y = F.linear(p, w, self.patch_embed.bias)
        else:
            y = x
        y = self._attn(self.attn_temp, self.norm_temp, y)
        y = self.ff(self.norm1(y)) + y
        if self.compress_video_features:
            y = self._pool(y)
            y = self._attn(self.attn_temp_2, self.norm_temp_2, y)
            y = self.ff_2(self.norm2(y)) + y
            y = self._pool(y)
        return y


class PerceiverAttention(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        self.heads, self.dim_head = heads, dim_head
        inner = dim_head * heads
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)

    def forward(self, x, latents):
        x, latents = self.norm1(x), self.norm2(latents)
        q = self.to_q(latents)
        kv = self.to_kv(torch.cat((x, latents), dim=-2))
        inner = self.heads * self.dim_head
        return self.to_out(kernels.attention(q, kv[..., :inner], kv[..., inner:], self.heads))


class Resampler(nn.Module):
    """Perceiver resampler -> ``num_queries`` IP tokens (resampler.py:83-160)."""

    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                 ff_mult=4, **_):
        super().__init__()
        self.latents = nn.Parameter(torch.randn(1, num_queries, dim) / dim ** 0.5)
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.layers = nn.ModuleList([nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                                                    _NoBiasFF(dim, ff_mult)]) for _ in range(depth)])

    def forward(self, x, relative_postion_tensor=None):
        latents = self.latents.repeat(x.size(0), 1, 1)
        x = self.proj_in(x)
        for attn, ff in self.layers:
            latents = attn(x, latents) + latents
            latents = ff(latents) + latents
        return self.norm_out(self.proj_out(latents))


# ----------------------------------------------------------------------------------------------
# UNet blocks
def _motion(in_channels, use, kwargs):
    return VanillaTemporalModule(in_channels=in_channels, **kwargs) if use else None


class CrossAttnDownBlock3D(nn.Module):
    has_cross_attention = True

    def __init__(self, *, in_channels, out_channels, temb_channels, num_layers, resnet_eps, resnet_groups,
                 cross_attention_dim, attn_num_head_channels, add_downsample, downsample_padding, use_motion_module,
                 motion_module_kwargs, image_cross_attention_dim, scale, num_tokens, **_):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(in_channels=in_channels if i == 0 else out_channels,
                                                    out_channels=out_channels, temb_channels=temb_channels,
                                                    eps=resnet_eps, groups=resnet_groups) for i in range(num_layers)])
        self.attentions = nn.ModuleList([Transformer3DModel(
            attn_num_head_channels, out_channels // attn_num_head_channels, out_channels, cross_attention_dim,
            image_cross_attention_dim, resnet_groups, scale, num_tokens) for _ in range(num_layers)])
        self.motion_modules = nn.ModuleList([_motion(out_channels, use_motion_module, motion_module_kwargs)
                                             for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample3D(out_channels, out_channels=out_channels,
                                                        padding=downsample_padding, name="op")]) if add_downsample else None

    def forward_cl(self, x, temb, ctx, frames, pano=False):
        outs = ()
        for res, attn, mm in zip(self.resnets, self.attentions, self.motion_modules):
            x = res.forward_cl(x, temb, frames, pano)
            x = attn.forward_cl(x, ctx, frames)
            if mm is not None:
                x = mm.forward_cl(x, frames)
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0].forward_cl(x, pano)
            outs += (x,)
        return x, outs


class DownBlock3D(nn.Module):
    has_cross_attention = False

    def __init__(self, *, in_channels, out_channels, temb_channels, num_layers, resnet_eps, resnet_groups,
                 add_downsample, downsample_padding, use_motion_module, motion_module_kwargs, **_):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(in_channels=in_channels if i == 0 else out_channels,
                                                    out_channels=out_channels, temb_channels=temb_channels,
                                                    eps=resnet_eps, groups=resnet_groups) for i in range(num_layers)])
        self.motion_modules = nn.ModuleList([_motion(out_channels, use_motion_module, motion_module_kwargs)
                                             for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample3D(out_channels, out_channels=out_channels,
                                                        padding=downsample_padding, name="op")]) if add_downsample else None

    def forward_cl(self, x, temb, ctx, frames, pano=False, use_motion=True):
        outs = ()
        for res, mm in zip(self.resnets, self.motion_modules):
            x = res.forward_cl(x, temb, frames, pano)
            if mm is not None and use_motion:
                x = mm.forward_cl(x, frames)
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0].forward_cl(x, pano)
            outs += (x,)
        return x, outs


class UNetMidBlock3DCrossAttn(nn.Module):
    has_cross_attention = True

    def __init__(self, *, in_channels, temb_channels, resnet_eps, resnet_groups, cross_attention_dim,
                 attn_num_head_channels, use_motion_module, motion_module_kwargs, image_cross_attention_dim, scale,
                 num_tokens, **_):
        super().__init__()
        mk = lambda: ResnetBlock3D(in_channels=in_channels, out_channels=in_channels, temb_channels=temb_channels,
                                   eps=resnet_eps, groups=resnet_groups)
        self.attentions = nn.ModuleList([Transformer3DModel(
            attn_num_head_channels, in_channels // attn_num_head_channels, in_channels, cross_attention_dim,
            image_cross_attention_dim, resnet_groups, scale, num_tokens)])
        self.resnets = nn.ModuleList([mk(), mk()])
        self.motion_modules = nn.ModuleList([_motion(in_channels, use_motion_module, motion_module_kwargs)])

    def forward_cl(self, x, temb, ctx, frames, pano=False):
        x = self.resnets[0].forward_cl(x, temb, frames, pano)
        for attn, res, mm in zip(self.attentions, self.resnets[1:], self.motion_modules):
            x = attn.forward_cl(x, ctx, frames)
            if mm is not None:
                x = mm.forward_cl(x, frames)
            x = res.forward_cl(x, temb, frames, pano)
        return x


class CrossAttnUpBlock3D(nn.Module):
    has_cross_attention = True

    def __init__(self, *, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, resnet_eps,
                 resnet_groups, cross_attention_dim, attn_num_head_channels, add_upsample, use_motion_module,
                 motion_module_kwargs, image_cross_attention_dim, scale, num_tokens, **_):
        super().__init__()
        resnets = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock3D(in_channels=rin + skip, out_channels=out_channels, temb_channels=temb_channels,
                                         eps=resnet_eps, groups=resnet_groups))
        self.resnets = nn.ModuleList(resnets)
        self.attentions = nn.ModuleList([Transformer3DModel(
            attn_num_head_channels, out_channels // attn_num_head_channels, out_channels, cross_attention_dim,
            image_cross_attention_dim, resnet_groups, scale, num_tokens) for _ in range(num_layers)])
        self.motion_modules = nn.ModuleList([_motion(out_channels, use_motion_module, motion_module_kwargs)
                                             for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample3D(out_channels, out_channels=out_channels)]) if add_upsample else None


class UpBlock3D(nn.Module):
    has_cross_attention = False

    def __init__(self, *, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, resnet_eps,
                 resnet_groups, add_upsample, use_motion_module, motion_module_kwargs, **_):
        super().__init__()
        resnets = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            rin = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock3D(in_channels=rin + skip, out_channels=out_channels, temb_channels=temb_channels,
                                         eps=resnet_eps, groups=resnet_groups))
        self.resnets = nn.ModuleList(resnets)
        self.motion_modules = nn.ModuleList([_motion(out_channels, use_motion_module, motion_module_kwargs)
                                             for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample3D(out_channels, out_channels=out_channels)]) if add_upsample else None


_BLOCKS = {"CrossAttnDownBlock3D": CrossAttnDownBlock3D, "DownBlock3D": DownBlock3D,
           "CrossAttnUpBlock3D": CrossAttnUpBlock3D, "UpBlock3D": UpBlock3D}


@dataclass
class UNet3DConditionOutput:
    sample: torch.Tensor


class _Cfg(dict):
    __getattr__ = dict.get


class UNet3DConditionModel(nn.Module):
    """animatediff/models/unet.py:60-358 (constructor), 632-856 (forward), 858-909 (from_pretrained_2d)."""

    def __init__(self, sample_size=None, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True,
                 freq_shift=0,
                 down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
                 mid_block_type="UNetMidBlock3DCrossAttn",
                 up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
                 only_cross_attention=False, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 downsample_padding=1, mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
                 cross_attention_dim=1280, attention_head_dim=8, dual_cross_attention=False, use_linear_projection=False,
                 class_embed_type=None, num_class_embeds=None, upcast_attention=False, resnet_time_scale_shift="default",
                 use_motion_module=False, motion_module_resolutions=(1, 2, 4, 8), motion_module_mid_block=False,
                 motion_module_decoder_only=False, motion_module_type=None, motion_module_kwargs=None,
                 unet_use_cross_frame_attention=None, unet_use_temporal_attention=None, image_hidden_size=1280,
                 use_ip_plus_cross_attention=False, scale=1.0, num_tokens=4, use_learnable_scale=False,
                 use_inflated_groupnorm=False, use_fps_condition=False, use_outpaint=False, use_relative_postions=False,
                 adapter_cross_attention_dim=1024, image_cross_attention_dim=1024, ip_plus_condition="image",
                 use_adapter_temporal_projection=False, compress_video_features=False, **unused):
        super().__init__()
        if not use_linear_projection or not use_ip_plus_cross_attention or not use_motion_module:
            raise NotImplementedError("imagine360_amd implements the dual-branch configuration of configs/prompt-dual.yaml "
                                      "(use_linear_projection, use_ip_plus_cross_attention, use_motion_module)")
        motion_module_kwargs = dict(motion_module_kwargs or {})
        self.config = _Cfg(in_channels=in_channels, out_channels=out_channels, block_out_channels=tuple(block_out_channels),
                           layers_per_block=layers_per_block, attention_head_dim=attention_head_dim,
                           cross_attention_dim=cross_attention_dim, norm_num_groups=norm_num_groups, norm_eps=norm_eps,
                           sample_size=sample_size, center_input_sample=center_input_sample, num_tokens=num_tokens,
                           use_outpaint=use_outpaint)
        self.sample_size = sample_size
        c0 = block_out_channels[0]
        ted = c0 * 4
        self.use_relative_postions = use_relative_postions
        self.image_cross_attention_dim = image_cross_attention_dim
        self.ip_plus_condition = ip_plus_condition
        self.conv_in = InflatedConv3d(in_channels * 2 + 1 if use_outpaint else in_channels, c0, kernel_size=3, padding=(1, 1))
        self.time_proj = Timesteps(c0, flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(c0, ted)
        if use_relative_postions == "WithAdapter":
            self.add_cond_proj = Timesteps(c0, flip_sin_to_cos, freq_shift)
            self.add_cond_embedding = TimestepEmbedding(c0 * 6, image_cross_attention_dim)
            self.cond_rp_proj = nn.Linear(image_cross_attention_dim, image_cross_attention_dim // 4 * 3, bias=False)
            self.add_cond_embedding2 = TimestepEmbedding(c0, image_cross_attention_dim // 4)
        elif use_relative_postions:
            raise NotImplementedError("only use_relative_postions='WithAdapter' (prompt-dual.yaml:37) is implemented")
        if use_fps_condition:
            self.fps_embedding = TimestepEmbedding(c0, ted)
            nn.init.zeros_(self.fps_embedding.linear_2.weight)
            nn.init.zeros_(self.fps_embedding.linear_2.bias)
        if ip_plus_condition == "video" and use_adapter_temporal_projection:
            self.temporal_proj = TemporalProjection(dim=image_hidden_size, dim_head=64, heads=8,
                                                    compress_video_features=compress_video_features)
            spc = self.temporal_proj.spacial_compress
        else:
            self.temporal_proj, spc = nn.Identity(), False
        self.image_proj_model = Resampler(dim=adapter_cross_attention_dim, depth=4, dim_head=64, heads=12,
                                          num_queries=num_tokens, embedding_dim=image_hidden_size * 4 if spc else image_hidden_size,
                                          output_dim=image_cross_attention_dim, ff_mult=4)
        if isinstance(attention_head_dim, int):
            attention_head_dim = (attention_head_dim,) * len(down_block_types)
        common = dict(temb_channels=ted, resnet_eps=norm_eps, resnet_groups=norm_num_groups,
                      cross_attention_dim=cross_attention_dim, motion_module_kwargs=motion_module_kwargs,
                      image_cross_attention_dim=image_cross_attention_dim, scale=scale, num_tokens=num_tokens)
        self.down_blocks = nn.ModuleList()
        oc = c0
        for i, t in enumerate(down_block_types):
            ic, oc = oc, block_out_channels[i]
            self.down_blocks.append(_BLOCKS[t](
                in_channels=ic, out_channels=oc, num_layers=layers_per_block, attn_num_head_channels=attention_head_dim[i],
                add_downsample=i != len(block_out_channels) - 1, downsample_padding=downsample_padding,
                use_motion_module=use_motion_module and (2 ** i in motion_module_resolutions) and not motion_module_decoder_only,
                **common))
        self.mid_block = UNetMidBlock3DCrossAttn(in_channels=block_out_channels[-1], attn_num_head_channels=attention_head_dim[-1],
                                                 use_motion_module=use_motion_module and motion_module_mid_block, **common)
        self.up_blocks = nn.ModuleList()
        rc, rh = list(reversed(block_out_channels)), list(reversed(attention_head_dim))
        oc = rc[0]
        self.num_upsamplers = 0
        for i, t in enumerate(up_block_types):
            prev, oc = oc, rc[i]
            ic = rc[min(i + 1, len(block_out_channels) - 1)]
            last = i == len(block_out_channels) - 1
            self.num_upsamplers += 0 if last else 1
            self.up_blocks.append(_BLOCKS[t](
                in_channels=ic, out_channels=oc, prev_output_channel=prev, num_layers=layers_per_block + 1,
                attn_num_head_channels=rh[i], add_upsample=not last,
                use_motion_module=use_motion_module and (2 ** (3 - i) in motion_module_resolutions), **common))
        self.conv_norm_out = InflatedGroupNorm(norm_num_groups, c0, eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = InflatedConv3d(c0, out_channels, kernel_size=3, padding=1)
        self._ip_cache = DerivedCache()

    # ---- reference API --------------------------------------------------------------------------
    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def enable_xformers_memory_efficient_attention(self, *a, **k):
        for m in self.modules():
            if isinstance(m, BasicTransformerBlock):
                m.set_use_memory_efficient_attention_xformers(True)

    def disable_xformers_memory_efficient_attention(self):
        for m in self.modules():
            if isinstance(m, BasicTransformerBlock):
                m.set_use_memory_efficient_attention_xformers(False)

    @classmethod
    def from_config(cls, config, **kwargs):
        import inspect
        ok = set(inspect.signature(cls.__init__).parameters)
        merged = {k: v for k, v in {**dict(config), **kwargs}.items() if k in ok}      # unknown keys are dropped (unet.py:886)
        return cls(**merged)

    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, subfolder=None, unet_additional_kwargs=None):
        """SD-2.1 2-D UNet directory -> 3-D model: widen conv_in 4 -> 9 channels with zeros (unet.py:858-909)."""
        import json
        import os
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        cfg_file = os.path.join(pretrained_model_path, "config.json")
        if not os.path.isfile(cfg_file):
            raise RuntimeError(f"{cfg_file} does not exist")
        with open(cfg_file) as f:
            config = json.load(f)
        config["down_block_types"] = ["CrossAttnDownBlock3D"] * 3 + ["DownBlock3D"]
        config["up_block_types"] = ["UpBlock3D"] + ["CrossAttnUpBlock3D"] * 3
        unet_additional_kwargs = dict(unet_additional_kwargs or {})
        model = cls.from_config(config, **unet_additional_kwargs)
        model_file = os.path.join(pretrained_model_path, "diffusion_pytorch_model.bin")
        if not os.path.isfile(model_file):
            raise RuntimeError(f"{model_file} does not exist")
        state_dict = torch.load(model_file, map_location="cpu")
        if "use_outpaint" in unet_additional_kwargs:
            w = torch.zeros_like(model.conv_in.weight)
            w[:, :4] = state_dict["conv_in.weight"]
            state_dict["conv_in.weight"] = w
            state_dict["conv_in.bias"] = state_dict.get("conv_in.bias", model.conv_in.bias.detach())
        m, u = model.load_state_dict(state_dict, strict=False)
        print(f"### missing keys: {len(m)}; \n### unexpected keys: {len(u)};")
        return model

    # ---- conditioning helpers shared with the dual model -----------------------------------------
    def time_embed(self, timesteps, fps=None):
        emb = self.time_embedding(self.time_proj(timesteps).to(self.dtype))
        if fps is not None:
            emb = emb + self.fps_embedding(self.time_proj(fps.to(self.dtype)).to(self.dtype))
        return emb

    def ip_tokens_clean(self, feats):
        """temporal_proj + image_proj_model (step-invariant); feats [B, F, 4096, 256] -> [B, num_tokens, d].
        Cached on the identity/version of ``feats`` so a denoising loop computes it once."""
        def build():
            t = self.temporal_proj(feats.to(self.dtype))
            return self.image_proj_model(t.reshape(t.shape[0], -1, t.shape[-1]))
        # keyed on the features and on EVERY adapter parameter: a partial load_state_dict or an in-place LoRA merge into
        # any of them must rebuild the tokens
        params = (feats, *self.temporal_proj.parameters(), *self.image_proj_model.parameters())
        return self._ip_cache.get("ip", params, build)

    def relpos_tokens(self, rel_pos, pitchs, n_tokens):
        """Per-frame relative-position / pitch embeddings added to the IP tokens (MVGenModel.py:189-222),
        batched over frames instead of the reference's Python loop."""
        b, f = rel_pos.shape[:2]
        e1 = self.add_cond_proj(rel_pos.reshape(-1).float()).reshape(b * f, -1).to(self.dtype)
        e1 = self.cond_rp_proj(self.add_cond_embedding(e1))
        e2 = self.add_cond_embedding2(self.add_cond_proj(pitchs.reshape(-1).float()).to(self.dtype))
        e = torch.cat([e1, e2], dim=-1).reshape(b, f, -1)
        if n_tokens > f:
            e = torch.cat([e, e[:, -1:].expand(-1, n_tokens - f, -1)], dim=1)
        return e

    def conv_in_cl(self, x, pano=False):
        return self.conv_in.forward_cl(x, wrap=pano)

    def conv_out_cl(self, x, pano=False):
        return self.conv_out.forward_cl(self.conv_norm_out.forward_cl(x, silu=True), wrap=pano)

    # ---- single-branch forward (kept API; the dual pipeline drives the blocks itself) ----------------
    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, attention_mask=None, return_dict=True,
                use_ip_plus_cross_attention=False, reference_images_clip_feat=None, use_fps_condition=False,
                fps_tensor=None, relative_position_tensor=None):
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.int64, device=sample.device)
        timestep = timestep.reshape(-1).expand(sample.shape[0])
        fps = None
        if use_fps_condition:
            fps = torch.as_tensor(fps_tensor, device=sample.device).reshape(-1).expand(sample.shape[0])
        emb = self.time_embed(timestep, fps)
        ctx = encoder_hidden_states
        if use_ip_plus_cross_attention:
            ip = self.ip_tokens_clean(reference_images_clip_feat)
            ctx = torch.cat([ctx, ip.to(ctx.dtype)], dim=1)
        x, f = to_cl(sample.to(self.dtype))
        x = self.conv_in_cl(x)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk.forward_cl(x, emb, ctx, f)
            skips += list(outs)
        x = self.mid_block.forward_cl(x, emb, ctx, f)
        for blk in self.up_blocks:
            for j, res in enumerate(blk.resnets):
                x = res.forward_cl((x, skips.pop()), emb, f)
                if blk.has_cross_attention:
                    x = blk.attentions[j].forward_cl(x, ctx, f)
                if blk.motion_modules[j] is not None:
                    x = blk.motion_modules[j].forward_cl(x, f)
            if blk.upsamplers is not None:
                x = blk.upsamplers[0].forward_cl(x)
        out = from_cl(self.conv_out_cl(x), f)
        return UNet3DConditionOutput(sample=out) if return_dict else (out,)
