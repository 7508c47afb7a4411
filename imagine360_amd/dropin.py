"""Drop-in installation: make the reference's import paths resolve to imagine360_amd.

    import imagine360_amd.dropin; imagine360_amd.dropin.install()
    from animatediff.models.unet import UNet3DConditionModel            # -> imagine360_amd.unet3d
    from animatediff.pipelines.pipeline_animation_inference_dual import AnimationPipeline
    from src.models.MVGenModel import MultiViewBaseModel
    from src.utils.pano import pad_pano, unpad_pano
    from diffusers import AutoencoderKL, DDIMScheduler

These are the imports of the hot path in inference_dual_p2e.py:20-38 (SURVEY.md section 8b tier 1).  Only the
modules listed there are aliased; the reference's host-side preprocessing (decord, GeoCalib, Qwen-VL,
pano_utils) is out of scope and keeps coming from the reference checkout.  ``install()`` refuses to
shadow modules that are already imported unless ``force=True``.
"""
import importlib.machinery
import sys
import types

from . import mv_model, pano_geometry, pipeline, scheduler, synthetic, unet3d, vae

_ALIASES = {
    "animatediff.models.unet": dict(UNet3DConditionModel=unet3d.UNet3DConditionModel,
                                    UNet3DConditionOutput=unet3d.UNet3DConditionOutput),
    "animatediff.models.resnet": dict(InflatedConv3d=unet3d.InflatedConv3d, InflatedGroupNorm=unet3d.InflatedGroupNorm,
                                      ResnetBlock3D=unet3d.ResnetBlock3D, Upsample3D=unet3d.Upsample3D,
                                      Downsample3D=unet3d.Downsample3D),
    "animatediff.models.attention": dict(Transformer3DModel=unet3d.Transformer3DModel,
                                         BasicTransformerBlock=unet3d.BasicTransformerBlock,
                                         IPCrossAttention=unet3d.IPCrossAttention),
    "animatediff.models.motion_module": dict(VanillaTemporalModule=unet3d.VanillaTemporalModule,
                                             VersatileAttention=unet3d.VersatileAttention),
    "animatediff.models.resampler": dict(Resampler=unet3d.Resampler, TemporalProjection=unet3d.TemporalProjection),
    "animatediff.pipelines.pipeline_animation_inference_dual": dict(
        AnimationPipeline=pipeline.AnimationPipeline, AnimationPipelineOutput=pipeline.AnimationPipelineOutput),
    "src.models.MVGenModel": dict(MultiViewBaseModel=mv_model.MultiViewBaseModel),
    "src.modules.attn_perspano": dict(WarpAttn=mv_model.WarpAttn),
    "src.modules.transformer": dict(SphericalPE=mv_model.SphericalPE),
    "src.utils.pano": dict(pad_pano=pano_geometry.pad_pano, unpad_pano=pano_geometry.unpad_pano,
                           icosahedron_sample_camera=synthetic.icosahedron_angles),
    "diffusers": dict(AutoencoderKL=vae.AutoencoderKL, DDIMScheduler=scheduler.DDIMScheduler),
}


def install(force=False):
    """Register alias modules (and their parent packages) in ``sys.modules``; returns the names installed."""
    done = []
    for name, attrs in _ALIASES.items():
        parts = name.split(".")
        for i in range(1, len(parts) + 1):
            pkg = ".".join(parts[:i])
            existing = sys.modules.get(pkg)
            if existing is not None and not getattr(existing, "__im360_alias__", False):
                if i == len(parts) and not force:
                    raise RuntimeError(f"{pkg} is already imported from {getattr(existing, '__file__', '?')}; "
                                       "call install() before importing the reference, or pass force=True")
                if i < len(parts):
                    continue
            if existing is None or (i == len(parts) and force):
                mod = types.ModuleType(pkg)
                mod.__spec__ = importlib.machinery.ModuleSpec(pkg, None, is_package=i < len(parts))
                mod.__path__ = []
                mod.__im360_alias__ = True
                sys.modules[pkg] = mod
                if i > 1:
                    setattr(sys.modules[".".join(parts[:i - 1])], parts[i - 1], mod)
        m = sys.modules[name]
        for k, v in attrs.items():
            setattr(m, k, v)
        done.append(name)
    return done


def uninstall():
    for name in [n for n, m in sys.modules.items() if getattr(m, "__im360_alias__", False)]:
        del sys.modules[name]
