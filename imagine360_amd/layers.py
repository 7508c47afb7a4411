"""Channels-last building blocks shared by the UNet, the cross-view blocks and the VAE.

Internal activation layout ("CL"): images ``[N, H, W, C]`` with ``N = batch * frames`` (batch-major),
i.e. token-major ``[N, H*W, C]`` for free.  Public ``forward`` methods of the mirrored reference
classes take/return the reference layout ``[b, c, f, h, w]``; ``forward_cl`` methods are the fast
internal path.  All arithmetic-heavy work goes through ``imagine360_amd.kernels`` (HIP); GEMM-shaped
Linear layers use torch (hipBLASLt), as SURVEY.md section 2b allows.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels


def to_cl(x5):
    """[b, c, f, h, w] -> ([b*f, h, w, c] contiguous, f)."""
    b, c, f, h, w = x5.shape
    return x5.permute(0, 2, 3, 4, 1).reshape(b * f, h, w, c).contiguous(), f


def from_cl(x, f):
    """[b*f, h, w, c] -> [b, c, f, h, w]."""
    n, h, w, c = x.shape
    return x.reshape(n // f, f, h, w, c).permute(0, 4, 1, 2, 3)


class DerivedCache:
    """Tensors derived from parameters (packed conv weights, fused QKV matrices), rebuilt whenever a
    source parameter is modified in place (``_version``) or replaced (``data_ptr``) -- e.g. by
    ``load_state_dict`` or the LoRA merge of inference_dual_p2e.py:175-195."""

    def __init__(self):
        self._store = {}

    def get(self, key, params, build):
        sig = tuple((p.data_ptr(), p._version, p.dtype, p.device, tuple(p.shape), p.stride()) for p in params if p is not None)
        hit = self._store.get(key)
        if hit is None or hit[0] != sig:
            if params and params[0] is not None and params[0].is_cuda and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                # a fill inside a capture would be replayed with every step and, in a two-stream capture, be built on one
                # stream and read unsynchronised on the other: graph_step.GraphedDenoiseStep warms every cache up first
                raise RuntimeError(f"derived-tensor cache miss ({key!r}) during a hipGraph capture: run the step eagerly once before capturing it")
            with torch.no_grad():
                # the entry keeps its sources alive, so their addresses cannot be recycled by another tensor
                hit = (sig, build(), tuple(params))
            self._store[key] = hit
            publish_to_all_streams(hit[1])
        return hit[1]


def publish_to_all_streams(t):
    """A cached tensor is built once on whatever stream was current and then read from every stream the model uses
    (MultiViewBaseModel.dual_stream / cfg_streams): wait for the build before anybody else can be handed the entry.  Cache
    fills are rare (first use, checkpoint load); a fill during a hipGraph capture stays stream-ordered as before."""
    ts = t if isinstance(t, (tuple, list)) else (t,)
    if any(torch.is_tensor(x) and x.is_cuda for x in ts) and not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream().synchronize()


class InflatedConv3d(nn.Conv2d):
    """Per-frame 2-D convolution (animatediff/models/resnet.py:19-27) on the HIP implicit-GEMM kernel.
    Parameters keep nn.Conv2d's names/shapes so reference checkpoints load unchanged."""

    use_up2 = True          # class-wide switch: sub-pixel form of the upsample convolutions (False: A/B against the 9-tap form)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._derived = DerivedCache()

    def packed_weight(self):
        cin = self.in_channels
        cin_pad = ((cin + 31) // 32) * 32
        return self._derived.get("w", (self.weight,), lambda: kernels.pack_conv_weight(self.weight, cin_pad))

    @property
    def cin_padded(self):
        return ((self.in_channels + 31) // 32) * 32

    def forward_cl(self, x, wrap=False, up=False, x_off=0, wout=None, temb=None, imgs_per_temb=1, res=None, y_off=0, gn_stats=False):
        """x [N, H, W, Cin(+zero pad to a multiple of 32)] channels-last."""
        if x.shape[-1] != self.cin_padded:
            x = F.pad(x, (0, self.cin_padded - x.shape[-1]))
        if up and self.use_up2 and self.in_channels % 64 == 0 and self.out_channels % 8 == 0 and res is None and temb is None \
                and x_off == 0 and y_off == 0 and wout is None and self.stride[0] == 1:
            # nearest x2 + conv3x3 as four 2x2 convolutions of the low-resolution input (pre-summed taps): 4/9 of the MACs
            w4 = self._derived.get("w_up2", (self.weight,), lambda: kernels.pack_conv_up2_weight(self.weight))
            return kernels.conv_up2(x, w4, self.out_channels, bias=self.bias, wrap=wrap)
        return kernels.conv2d(x, self.packed_weight(), self.out_channels, bias=self.bias, stride=self.stride[0],
                              up=up, wrap=wrap, x_off=x_off, wout=wout, temb=temb, imgs_per_temb=imgs_per_temb,
                              res=res, y_off=y_off, gn_stats=gn_stats)

    def forward_cat(self, xa, xb, res=None):
        """1x1 convolution of the channel concatenation [xa | xb] without materialising it (the decoder's conv_shortcut)."""
        assert self.kernel_size == (1, 1) and xa.shape[-1] + xb.shape[-1] == self.in_channels
        return kernels.conv1x1_cat(xa, xb, self.packed_weight(), self.out_channels, bias=self.bias, res=res)

    def forward(self, x):
        if x.dim() == 5:
            xc, f = to_cl(x)
            return from_cl(self.forward_cl(xc), f)
        return self.forward_cl(x.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)


class InflatedGroupNorm(nn.GroupNorm):
    """Per-frame GroupNorm (animatediff/models/resnet.py:9-17) on the HIP stats/apply kernels."""

    def stats_cl(self, x, pad=0):
        return kernels.group_norm_stats(x, self.weight, self.bias, self.num_groups, self.eps, pad)

    def forward_cl(self, x, silu=False, pad=0):
        return kernels.group_norm(x, self.weight, self.bias, self.num_groups, self.eps, silu=silu, pad=pad)

    def forward(self, x):
        if x.dim() == 5:
            xc, f = to_cl(x)
            return from_cl(self.forward_cl(xc), f)
        return self.forward_cl(x.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)


def layer_norm(norm, x, pre=None, post=None, post_div=1):
    """nn.LayerNorm parameters, HIP kernel (optionally fused with the positional-encoding adds)."""
    return kernels.layer_norm(x.contiguous(), norm.weight, norm.bias, norm.eps, pre=pre, post=post, post_div=post_div)


ROUTE_MIN_TOKENS = int(os.environ.get("IM360_ROUTE_MIN_TOKENS", 65536))      # token-major GEMMs with at least this many rows go to the MFMA ring kernel (tests lower it; the variable is for A/B runs of bench.py)
GEGLU_FUSED_MAX_K = int(os.environ.get("IM360_GEGLU_MAX_K", 640))       # widest input the fused GEGLU projection takes (see GEGLU.forward)
ROUTE_ON_CPU = False          # tests only: take the routed branch without a GPU (kernels monkeypatched by torch stand-ins)
ROUTE_N_MULT = 320            # output widths the ring kernel's 320-column tiles cover (tests on stand-ins lower it)


def _gemm_kernel_pays(m, k, n):
    """Shapes where the implicit-GEMM conv kernel (as a 1x1 conv with its fused bias / residual epilogue) is at least
    as fast as hipBLASLt + a separate residual add: its 256 x 320 tiles need K % 64 == 0, N % 320 == 0 and enough
    work to fill the chip -- measured (tools/bench_kernels.py linear): every level-0 / level-1 token
    count of cfg2 (>= 65 536 tokens) wins or ties, the 40 960-token level-2 shapes lose to hipBLASLt's deep-K kernels."""
    return k % 64 == 0 and n % ROUTE_N_MULT == 0 and m >= ROUTE_MIN_TOKENS and ((m + 255) // 256) * (n // ROUTE_N_MULT) >= min(512, ROUTE_MIN_TOKENS // 128)


def _routed(x, m, k, n):
    return (x.is_cuda or ROUTE_ON_CPU) and _gemm_kernel_pays(m, k, n)


def gemm_linear(weight, bias, x, res=None, cache=None, key="w1x1", row_stats=False, gn_hw=None):
    """``x @ weight.T + bias (+ res)``.  Large token counts: one launch of the MFMA implicit-GEMM kernel (as a 1x1
    conv over a [M, 1, 1, K] view) with bias and residual in its epilogue -- no separate elementwise pass over the
    activations; otherwise hipBLASLt (+ an add).  ``cache``: DerivedCache holding the packed weight.
    ``row_stats``: return (y, stats) where stats are the per-row LayerNorm statistics of y written by the same launch
    (``kernels.linear``), or None when the GEMM went to hipBLASLt (the consumer then runs the LayerNorm kernel)."""
    n, k = weight.shape
    m = x.numel() // k
    if not _routed(x, m, k, n) or (res is not None and res.shape[-1] != n):
        y = F.linear(x, weight, bias)
        y = y if res is None else y + res
        return (y, None) if row_stats else y
    wp = cache.get(key, (weight,), lambda: kernels.pack_conv_weight(weight.detach().reshape(n, k, 1, 1)))
    if row_stats:
        return kernels.linear(x.contiguous(), wp, n, bias=bias, res=None if res is None else res.contiguous(), row_stats=True)
    if gn_hw:
        # the consumer is a GroupNorm over images of gn_hw tokens: its partial sums come out of this launch's epilogue
        return kernels.linear(x.contiguous(), wp, n, bias=bias, res=None if res is None else res.contiguous(), gn_hw=gn_hw)
    r4 = None if res is None else res.contiguous().reshape(m, 1, 1, n)
    y = kernels.conv2d(x.contiguous().reshape(m, 1, 1, k), wp, n, bias=bias, res=r4)
    return y.reshape(*x.shape[:-1], n)


def ln_linear(norm, weight, bias, x, stats, cache, key, post=None, post_div=1):
    """``Linear(LayerNorm(x) [+ post[(row // post_div) % len(post)]])``.  With the rows' statistics from the producing GEMM
    (``gemm_linear(..., row_stats=True)``) the normalisation is folded into the GEMM (``kernels.linear_ln``): the
    LayerNorm pass over the activations and its output tensor disappear; without them: LayerNorm kernel + ``gemm_linear``."""
    n, k = weight.shape
    m = x.numel() // k
    if stats is None or not _routed(x, m, k, n) or (post is not None and post_div % 256):      # (the kernel adds one table row per 256-row tile)
        return gemm_linear(weight, bias, layer_norm(norm, x, post=post, post_div=post_div), cache=cache, key=key)

    def build():
        wg, c1, c2 = kernels.fold_layer_norm(weight, bias, norm.weight, norm.bias)
        return kernels.pack_conv_weight(wg.reshape(n, k, 1, 1).contiguous()), c1, c2

    wp, c1, c2 = cache.get(key + "_ln", (weight, bias, norm.weight, norm.bias), build)
    tab = None
    if post is not None:
        # the positional rows added AFTER the normalisation go through the projection once: a [Q, n] fp32 table
        # (+ c2: the kernel fetches ONE column vector per tile, rows of the table include the constant term)
        tab = cache.get(key + "_tab", (weight, post, bias, norm.weight, norm.bias), lambda: (post.float() @ weight.detach().float().t() + c2[None, :]).contiguous())
    return kernels.linear_ln(x.contiguous(), wp, c1, c2, stats, norm.eps, n, tab=tab, tab_div=post_div, tab_has_c2=tab is not None)


def _module_cache(mod):
    return mod.__dict__.setdefault("_im360_derived", DerivedCache())


def linear(lin, x, row_stats=False):
    """nn.Linear forward through ``gemm_linear``."""
    return gemm_linear(lin.weight, lin.bias, x, cache=_module_cache(lin), row_stats=row_stats)


def linear_residual(lin, x, res, row_stats=False, gn_hw=None):
    """``lin(x) + res`` through ``gemm_linear``."""
    return gemm_linear(lin.weight, lin.bias, x, res=res, cache=_module_cache(lin), row_stats=row_stats, gn_hw=gn_hw)


class GEGLU(nn.Module):
    """x -> a * gelu(gate), (a | gate) = proj(x)  (diffusers/models/activations.py:93-125)."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x, ln=None, stats=None):
        """``ln`` (nn.LayerNorm): the normalisation in front of the block, applied here -- folded into the GEMM when the
        rows' statistics ``stats`` came with x (``gemm_linear(..., row_stats=True)``), by the LayerNorm kernel otherwise."""
        two_i, k = self.proj.weight.shape
        m = x.numel() // k
        # large token counts: projection, bias and the gated activation in ONE launch of the MFMA GEMM kernel (the
        # 2I-wide intermediate never goes to HBM); otherwise hipBLASLt + the elementwise kernel
        # (measured, tools/bench_kernels.py geglu_fused: wins for K <= 640 at >= 64k tokens, loses at K = 1280 / 40k)
        fused = (x.is_cuda or ROUTE_ON_CPU) and k % 64 == 0 and k <= GEGLU_FUSED_MAX_K and two_i % 256 == 0 and m >= ROUTE_MIN_TOKENS
        cache = _module_cache(self)
        bias = None if self.proj.bias is None else self.proj.bias.detach()
        if fused and ln is not None and stats is not None:
            def build():
                wg, c1, c2 = kernels.fold_layer_norm(self.proj.weight, bias, ln.weight, ln.bias)
                wp, c1p = kernels.pack_geglu(wg, c1)
                return wp, c1p.contiguous(), kernels.interleave_geglu(wg, c2)[1].contiguous()
            ps = tuple(t for t in (self.proj.weight, self.proj.bias, ln.weight, ln.bias) if t is not None)
            wp, c1p, c2p = cache.get("geglu_ln", ps, build)
            return kernels.linear_geglu_ln(x.contiguous(), wp, c1p, c2p, stats, ln.eps, two_i // 2)
        if ln is not None:
            x = layer_norm(ln, x)
        if fused:
            ps = (self.proj.weight,) if self.proj.bias is None else (self.proj.weight, self.proj.bias)
            wp, bp = cache.get("geglu", ps, lambda: kernels.pack_geglu(self.proj.weight.detach(), bias))
            return kernels.linear_geglu(x.contiguous(), wp, bp, two_i // 2)
        return kernels.geglu(self.proj(x))


class FeedForward(nn.Module):
    """GEGLU feed-forward with the reference's parameter names ``net.0.proj`` / ``net.2``
    (diffusers/models/attention_lora.py:493-547; src/modules/transformer.py:20-40)."""

    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x, residual=None, ln=None, stats=None):
        """``ln`` / ``stats``: the LayerNorm in front of the block and (optionally) the statistics of x's rows, see GEGLU."""
        h = self.net[0](x, ln, stats)
        return self.net[2](h) if residual is None else linear_residual(self.net[2], h, residual)


class QKVAttention(nn.Module):
    """Attention projections with the reference's parameter names (``to_q`` / ``to_k`` / ``to_v`` /
    ``to_out.0``; diffusers/models/attention_processor.py:38-215).  Self-attention runs one fused QKV
    GEMM and hands strided views to the HIP flash-attention kernel."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False, out_list=True):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.inner_dim = heads, dim_head, inner
        ctx = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(ctx, inner, bias=bias)
        self.to_v = nn.Linear(ctx, inner, bias=bias)
        if out_list:
            self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        else:
            self.to_out = nn.Linear(inner, query_dim)
        self._derived = DerivedCache()
        self._use_memory_efficient_attention_xformers = False

    def out_proj(self, x, residual=None, row_stats=False):
        """Output projection (+ residual in the GEMM epilogue); ``row_stats``: (y, LayerNorm statistics of y's rows or None)."""
        lin = self.to_out[0] if isinstance(self.to_out, nn.ModuleList) else self.to_out
        if row_stats:
            return gemm_linear(lin.weight, lin.bias, x, res=residual, cache=_module_cache(lin), row_stats=True)
        return lin(x) if residual is None else linear_residual(lin, x, residual)

    def fused_qkv_weight(self):
        ps = (self.to_q.weight, self.to_k.weight, self.to_v.weight)
        return self._derived.get("qkv", ps, lambda: torch.cat([p.detach() for p in ps], dim=0).contiguous())

    def qkv(self, x):
        """x [..., C] -> fused [..., 3*inner] (valid when to_q/k/v share their input)."""
        return gemm_linear(self.fused_qkv_weight(), None, x, cache=self._derived, key="qkv_packed")

    def qkv_ln(self, norm, x, stats, post=None, post_div=1):
        """``qkv(LayerNorm(x) [+ post rows])`` with the normalisation folded into the GEMM when ``stats`` is given."""
        return ln_linear(norm, self.fused_qkv_weight(), None, x, stats, self._derived, "qkv_packed", post=post, post_div=post_div)

    def self_attention(self, x, bias=None, norm=None, stats=None):
        """x [B, N, C] -> attention output before the out projection (``norm``: LayerNorm applied to x first)."""
        qkv = self.qkv(x) if norm is None else self.qkv_ln(norm, x, stats)
        c = self.inner_dim
        return kernels.attention(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], self.heads, bias=bias)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None):
        if encoder_hidden_states is None:
            return self.out_proj(self.self_attention(hidden_states))
        q = linear(self.to_q, hidden_states)
        k, v = self.to_k(encoder_hidden_states), self.to_v(encoder_hidden_states)
        return self.out_proj(kernels.attention(q, k, v, self.heads))
