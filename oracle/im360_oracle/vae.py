"""Functional fp32 restatement of the SD-2.1 AutoencoderKL (diffusers/models/vae.py:565-610,
67-224; unet_2d_blocks.py:320-396, 869-926, 1646-1697; resnet.py:367-495, 77-190;
attention.py:247-379).  TEST INFRASTRUCTURE (see package docstring)."""
import torch
import torch.nn.functional as F

from .cfg import VAECfg


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + "weight"], sd[p + "bias"], stride=stride, padding=padding)


def _gn(sd, p, x, groups, eps=1e-6):
    return F.group_norm(x, groups, sd[p + "weight"], sd[p + "bias"], eps)


def resnet2d(sd, p, x, groups):
    """ResnetBlock2D with temb=None (resnet.py:367-495)."""
    h = _conv(sd, p + "conv1.", F.silu(_gn(sd, p + "norm1.", x, groups)))
    h = _conv(sd, p + "conv2.", F.silu(_gn(sd, p + "norm2.", h, groups)))
    if (p + "conv_shortcut.weight") in sd:
        x = _conv(sd, p + "conv_shortcut.", x, padding=0)
    return x + h


def attention_block(sd, p, x, groups):
    """Single-head AttentionBlock, fp32 softmax (attention.py:328-379)."""
    b, c, h, w = x.shape
    t = _gn(sd, p + "group_norm.", x, groups).reshape(b, c, h * w).transpose(1, 2)
    q = F.linear(t, sd[p + "query.weight"], sd[p + "query.bias"])
    k = F.linear(t, sd[p + "key.weight"], sd[p + "key.bias"])
    v = F.linear(t, sd[p + "value.weight"], sd[p + "value.bias"])
    s = torch.matmul(q, k.transpose(1, 2)) * (c ** -0.5)
    o = torch.matmul(s.softmax(-1), v)
    o = F.linear(o, sd[p + "proj_attn.weight"], sd[p + "proj_attn.bias"])
    return o.transpose(1, 2).reshape(b, c, h, w) + x


def _mid(sd, p, x, groups):
    x = resnet2d(sd, p + "resnets.0.", x, groups)
    x = attention_block(sd, p + "attentions.0.", x, groups)
    return resnet2d(sd, p + "resnets.1.", x, groups)


def encode_moments(sd, cfg: VAECfg, x):
    """AutoencoderKL.encode -> moments [n, 8, h/8, w/8] (vae.py:565-573)."""
    g = cfg.norm_num_groups
    h = _conv(sd, "encoder.conv_in.", x)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block):
            h = resnet2d(sd, f"encoder.down_blocks.{i}.resnets.{j}.", h, g)
        if i < n - 1:                                   # Downsample2D padding=0 -> pad (0,1,0,1)
            h = _conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv.", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    h = _mid(sd, "encoder.mid_block.", h, g)
    h = _conv(sd, "encoder.conv_out.", F.silu(_gn(sd, "encoder.conv_norm_out.", h, g)))
    return _conv(sd, "quant_conv.", h, padding=0)


def sample_posterior(moments, noise):
    """DiagonalGaussianDistribution.sample with explicit noise (vae.py:341-361)."""
    mean, logvar = moments.chunk(2, dim=1)
    return mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise


def decode(sd, cfg: VAECfg, z):
    """AutoencoderKL.decode (vae.py:575-610)."""
    g = cfg.norm_num_groups
    h = _conv(sd, "post_quant_conv.", z, padding=0)
    h = _conv(sd, "decoder.conv_in.", h)
    h = _mid(sd, "decoder.mid_block.", h, g)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            h = resnet2d(sd, f"decoder.up_blocks.{i}.resnets.{j}.", h, g)
        if i < n - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv.", h)
    return _conv(sd, "decoder.conv_out.", F.silu(_gn(sd, "decoder.conv_norm_out.", h, g)))
