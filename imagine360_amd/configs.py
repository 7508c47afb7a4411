"""Hyper-parameters the reference reads from files that are not in its repository (the SD-2.1
checkpoint directory) or from configs/prompt-dual.yaml, and builders for the random-init models used
by the benchmark and the tests."""
import torch

SD21_UNET_CONFIG = dict(
    sample_size=96, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True, norm_num_groups=32,
    norm_eps=1e-5, act_fn="silu", flip_sin_to_cos=True, freq_shift=0, downsample_padding=1, mid_block_scale_factor=1)

# configs/prompt-dual.yaml:16-45
PROMPT_DUAL_UNET_KWARGS = dict(
    use_motion_module=True, use_inflated_groupnorm=True, motion_module_resolutions=(1, 2, 4, 8),
    motion_module_mid_block=True, motion_module_type="Vanilla",
    motion_module_kwargs=dict(num_attention_heads=8, num_transformer_block=1,
                              attention_block_types=("Temporal_Self", "Temporal_Self"),
                              temporal_position_encoding=True, temporal_position_encoding_max_len=64,
                              temporal_attention_dim_div=1, zero_initialize=True),
    unet_use_cross_frame_attention=False, unet_use_temporal_attention=False, use_linear_projection=True,
    use_fps_condition=True, use_temporal_conv=False, use_relative_postions="WithAdapter",
    use_ip_plus_cross_attention=True, ip_plus_condition="video", num_tokens=64,
    use_adapter_temporal_projection=True, compress_video_features=True, image_hidden_size=256, use_outpaint=True)

# configs/prompt-dual.yaml:48-56
NOISE_SCHEDULER_KWARGS = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="linear",
                              steps_offset=1, clip_sample=False, prediction_type="v_prediction",
                              rescale_betas_zero_snr=True)

SD21_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                       layers_per_block=2, norm_num_groups=32, sample_size=768,
                       down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4)


def unet_config(width_div=1):
    cfg = dict(SD21_UNET_CONFIG)
    if width_div != 1:
        boc = tuple(c // width_div for c in SD21_UNET_CONFIG["block_out_channels"])
        cfg["block_out_channels"] = boc
        cfg["attention_head_dim"] = tuple(max(1, c // 64) for c in boc)      # keep head dim 64
    return cfg


def vae_config(width_div=1):
    cfg = dict(SD21_VAE_CONFIG)
    if width_div != 1:
        cfg["block_out_channels"] = tuple(c // width_div for c in SD21_VAE_CONFIG["block_out_channels"])
    return cfg


def build_unet(width_div=1, device=None, motion_heads=None):
    from .unet3d import UNet3DConditionModel
    kw = dict(PROMPT_DUAL_UNET_KWARGS)
    if motion_heads is not None:          # reduced-width test models: keep the temporal head dim a multiple of 8
        kw["motion_module_kwargs"] = dict(kw["motion_module_kwargs"], num_attention_heads=motion_heads)
    with torch.device(device) if device is not None else _null():
        return UNet3DConditionModel.from_config(unet_config(width_div), **kw)


def build_mv_model(width_div=1, device="cuda", dtype=torch.bfloat16, fill=True, xformers=True, motion_heads=None):
    """Random-init dual-branch model with the deterministic filler weights (no checkpoint is available
    offline).  ``xformers`` mirrors the shipped config's enable_xformers_memory_efficient_attention."""
    from .mv_model import MultiViewBaseModel
    from .weights import fill_module_
    with torch.device(device):
        mv = MultiViewBaseModel(build_unet(width_div, motion_heads=motion_heads), build_unet(width_div, motion_heads=motion_heads), pano_pad=True)
    if fill:
        fill_module_(mv)
    mv = mv.to(dtype).eval()
    if xformers:
        mv.unet.enable_xformers_memory_efficient_attention()
        mv.pano_unet.enable_xformers_memory_efficient_attention()
    return mv


def build_vae(width_div=1, device="cuda", dtype=torch.bfloat16, fill=True):
    from .vae import AutoencoderKL
    from .weights import fill_module_
    with torch.device(device):
        vae = AutoencoderKL(**vae_config(width_div))
    if fill:
        fill_module_(vae)
    return vae.to(dtype).eval()


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
