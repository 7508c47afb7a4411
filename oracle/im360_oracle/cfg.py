"""Hyper-parameters the reference takes from the SD-2.1 config.json + configs/prompt-dual.yaml
(SURVEY.md appendix A).  Test infrastructure."""
from dataclasses import dataclass, field
from typing import Tuple


@dataclass
class UNetCfg:
    in_channels: int = 4                      # widened to 2*4+1 = 9 by use_outpaint (animatediff/models/unet.py:134-135)
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    attention_head_dim: Tuple[int, ...] = (5, 10, 20, 20)   # used as number of heads (unet.py:228)
    cross_attention_dim: int = 1024
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    # prompt-dual.yaml:16-45
    motion_heads: int = 8
    motion_max_len: int = 64
    num_tokens: int = 64
    image_hidden_size: int = 256
    image_cross_attention_dim: int = 1024
    adapter_cross_attention_dim: int = 1024
    use_outpaint: bool = True
    # False = the reference's CPU path (`_attention`, cross-attention logit scale 1.0 quirk);
    # True = after enable_xformers_memory_efficient_attention() (logit scale d^-1/2). See unet.spatial_transformer.
    xformers: bool = False

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4


@dataclass
class VAECfg:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32


def sd21_unet_cfg(width_div: int = 1) -> UNetCfg:
    """Full-width SD-2.1 UNet (width_div=1) or a reduced-width variant that keeps head dim 64."""
    if width_div == 1:
        return UNetCfg()
    boc = tuple(c // width_div for c in (320, 640, 1280, 1280))
    heads = tuple(max(1, c // 64) for c in boc)
    return UNetCfg(block_out_channels=boc, attention_head_dim=heads)


def sd21_vae_cfg(width_div: int = 1) -> VAECfg:
    if width_div == 1:
        return VAECfg()
    return VAECfg(block_out_channels=tuple(c // width_div for c in (128, 256, 512, 512)))
