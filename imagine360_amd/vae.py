"""SD-2.1 AutoencoderKL (diffusers/models/vae.py:341-638, unet_2d_blocks.py:320-396, 869-926, 1646-1697,
resnet.py:77-190, 367-495, attention.py:247-379) on channels-last HIP kernels, with the reference's
state-dict keys and the ``encode(...).latent_dist.sample()`` / ``decode(z).sample`` surface.

The convolutions and GroupNorm+SiLU passes run on the same kernels as the UNet.  The single-head
d = C (512) mid-block attention runs as GEMM -> fp32 row softmax -> GEMM on the HIP kernels
(``kernels.single_head_attention``), the reference's own op order.
"""
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import kernels
from .layers import InflatedConv3d as Conv2dCL
from .layers import InflatedGroupNorm as GroupNormCL


class _Cfg(dict):
    __getattr__ = dict.get


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, groups=32, eps=1e-6):
        super().__init__()
        self.norm1 = GroupNormCL(groups, in_channels, eps=eps, affine=True)
        self.conv1 = Conv2dCL(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = GroupNormCL(groups, out_channels, eps=eps, affine=True)
        self.conv2 = Conv2dCL(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.conv_shortcut = Conv2dCL(in_channels, out_channels, kernel_size=1, stride=1, padding=0) \
            if in_channels != out_channels else None

    def forward_cl(self, x):
        h = self.conv1.forward_cl(self.norm1.forward_cl(x, silu=True))
        h = self.norm2.forward_cl(h, silu=True)
        short = x if self.conv_shortcut is None else self.conv_shortcut.forward_cl(x)
        return self.conv2.forward_cl(h, res=short)


class AttentionBlock(nn.Module):
    """Single-head spatial attention with fp32 softmax (attention.py:247-379)."""

    def __init__(self, channels, norm_num_groups=32, eps=1e-6):
        super().__init__()
        self.channels = channels
        self.group_norm = GroupNormCL(norm_num_groups, channels, eps=eps, affine=True)
        self.query = nn.Linear(channels, channels)
        self.key = nn.Linear(channels, channels)
        self.value = nn.Linear(channels, channels)
        self.proj_attn = nn.Linear(channels, channels)

    def forward_cl(self, x):
        n, h, w, c = x.shape
        t = self.group_norm.forward_cl(x).reshape(n, h * w, c)
        q, k, v = self.query(t), self.key(t), self.value(t)
        if kernels.can_single_head_attention(c, h * w):
            # one head of width c (512): QK^T and PV on the MFMA GEMM kernel, fp32 row softmax between them, per image
            out = torch.stack([kernels.single_head_attention(q[i], k[i], v[i], c ** -0.5) for i in range(n)])
        else:
            # token counts the GEMM kernel does not tile (e.g. a 360 x 640 frame: 45 x 80 = 3600 tokens): the reference's own
            # op sequence, baddbmm(alpha = scale) -> softmax(float) -> bmm (diffusers/models/attention.py:336-364), on rocBLAS
            s = torch.baddbmm(torch.empty(n, h * w, h * w, dtype=q.dtype, device=q.device), q, k.transpose(1, 2), beta=0, alpha=c ** -0.5)
            out = torch.bmm(torch.softmax(s.float(), dim=-1).to(q.dtype), v)
        return self.proj_attn(out).reshape(n, h, w, c) + x


class _Sampler(nn.Module):
    def __init__(self, channels, down):
        super().__init__()
        self.down = down
        self.conv = Conv2dCL(channels, channels, 3, stride=2 if down else 1, padding=0 if down else 1)

    def forward_cl(self, x):
        if self.down:                                        # F.pad (0,1,0,1) + stride-2 conv: taps at 2o .. 2o+2
            return self.conv.forward_cl(x, x_off=1, y_off=1)
        return self.conv.forward_cl(x, up=True)


class _EncBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([_Sampler(cout, True)]) if add_down else None

    def forward_cl(self, x):
        for r in self.resnets:
            x = r.forward_cl(x)
        return self.downsamplers[0].forward_cl(x) if self.downsamplers is not None else x


class _DecBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.upsamplers = nn.ModuleList([_Sampler(cout, False)]) if add_up else None

    def forward_cl(self, x):
        for r in self.resnets:
            x = r.forward_cl(x)
        return self.upsamplers[0].forward_cl(x) if self.upsamplers is not None else x


class _Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([AttentionBlock(c, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, groups), ResnetBlock2D(c, c, groups)])

    def forward_cl(self, x):
        return self.resnets[1].forward_cl(self.attentions[0].forward_cl(self.resnets[0].forward_cl(x)))


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, groups):
        super().__init__()
        boc = block_out_channels
        self.conv_in = Conv2dCL(in_channels, boc[0], kernel_size=3, stride=1, padding=1)
        self.down_blocks = nn.ModuleList([_EncBlock(boc[max(i - 1, 0)], boc[i], layers_per_block, groups, i != len(boc) - 1)
                                          for i in range(len(boc))])
        self.mid_block = _Mid(boc[-1], groups)
        self.conv_norm_out = GroupNormCL(groups, boc[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2dCL(boc[-1], 2 * out_channels, 3, padding=1)

    def forward_cl(self, x):
        x = self.conv_in.forward_cl(x)
        for b in self.down_blocks:
            x = b.forward_cl(x)
        x = self.mid_block.forward_cl(x)
        return self.conv_out.forward_cl(self.conv_norm_out.forward_cl(x, silu=True))


class Decoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, groups):
        super().__init__()
        rc = list(reversed(block_out_channels))
        self.conv_in = Conv2dCL(in_channels, rc[0], kernel_size=3, stride=1, padding=1)
        self.mid_block = _Mid(rc[0], groups)
        self.up_blocks = nn.ModuleList([_DecBlock(rc[max(i - 1, 0)], rc[i], layers_per_block + 1, groups, i != len(rc) - 1)
                                        for i in range(len(rc))])
        self.conv_norm_out = GroupNormCL(groups, rc[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = Conv2dCL(rc[-1], out_channels, 3, padding=1)

    def forward_cl(self, z):
        x = self.mid_block.forward_cl(self.conv_in.forward_cl(z))
        for b in self.up_blocks:
            x = b.forward_cl(x)
        return self.conv_out.forward_cl(self.conv_norm_out.forward_cl(x, silu=True))


class DiagonalGaussianDistribution:
    """vae.py:341-386; ``sample`` draws from the global torch RNG like the reference."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.sample_on_host = False

    def sample(self, generator=None):
        if self.sample_on_host:       # CPU-generator draw for RNG parity with the reference's CPU path
            noise = torch.randn(self.mean.shape, generator=generator, dtype=torch.float32)
        else:
            noise = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device)
        return self.mean + self.std * noise.to(device=self.parameters.device, dtype=self.parameters.dtype)

    def mode(self):
        return self.mean


@dataclass
class AutoencoderKLOutput:
    latent_dist: DiagonalGaussianDistribution


@dataclass
class DecoderOutput:
    sample: torch.Tensor


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",),
                 up_block_types=("UpDecoderBlock2D",), block_out_channels=(64,), layers_per_block=1, act_fn="silu",
                 latent_channels=4, norm_num_groups=32, sample_size=32, **_):
        super().__init__()
        self.config = _Cfg(in_channels=in_channels, out_channels=out_channels, block_out_channels=tuple(block_out_channels),
                           layers_per_block=layers_per_block, latent_channels=latent_channels,
                           norm_num_groups=norm_num_groups, sample_size=sample_size)
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = Conv2dCL(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = Conv2dCL(latent_channels, latent_channels, 1)
        self.use_slicing = False
        self.sample_on_host = False

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype

    @property
    def device(self):
        return self.quant_conv.weight.device

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kwargs):
        import json
        import os
        if subfolder is not None:
            path = os.path.join(path, subfolder)
        with open(os.path.join(path, "config.json")) as f:
            config = json.load(f)
        model = cls(**{k: v for k, v in config.items() if not k.startswith("_")})
        model.load_state_dict(torch.load(os.path.join(path, "diffusion_pytorch_model.bin"), map_location="cpu"))
        return model

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def encode_moments_cl(self, x_cl):
        return self.quant_conv.forward_cl(self.encoder.forward_cl(x_cl))

    def decode_cl(self, z_cl):
        return self.decoder.forward_cl(self.post_quant_conv.forward_cl(z_cl))

    def encode(self, x, return_dict=True):
        """x [n, 3, H, W]; ``return_dict`` only needs to be truthy (the pipeline passes an int, :434, :459)."""
        mom = self.encode_moments_cl(x.to(self.dtype).permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)
        post = DiagonalGaussianDistribution(mom)
        post.sample_on_host = self.sample_on_host
        return AutoencoderKLOutput(latent_dist=post) if return_dict else (post,)

    def decode(self, z, return_dict=True, batched=False):
        """``batched``: decode the whole batch in one pass even when slicing is enabled (same numbers: every op of the
        decoder is per image; slicing only bounds the reference's activation memory)."""
        zc = z.to(self.dtype).permute(0, 2, 3, 1).contiguous()
        if self.use_slicing and zc.shape[0] > 1 and not batched:
            dec = torch.cat([self.decode_cl(s) for s in zc.split(1)])
        else:
            dec = self.decode_cl(zc)
        dec = dec.permute(0, 3, 1, 2)
        return DecoderOutput(sample=dec) if return_dict else (dec,)

    def forward(self, sample, sample_posterior=False, return_dict=True, generator=None):
        post = self.encode(sample).latent_dist
        z = post.sample(generator=generator) if sample_posterior else post.mode()
        return self.decode(z, return_dict=return_dict)
