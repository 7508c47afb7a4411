"""Drop-in installation: make the reference's hot-path classes resolve to imagine360_amd.

    python -m imagine360_amd.dropin inference_dual_p2e.py --config configs/prompt-dual.yaml ...   # unmodified script
or, inside a script / notebook, before the reference imports:
    import imagine360_amd.dropin; imagine360_amd.dropin.install()

``install()`` is an OVERLAY on a reference checkout (which is where inference_dual_p2e.py lives), not a replacement of
its packages: for every module in ``_ALIASES`` it imports the checkout's real module and rebinds only the hot-path names
(SURVEY.md section 8b tier 1: UNet3DConditionModel, AnimationPipeline, MultiViewBaseModel, WarpAttn, pad_pano / unpad_pano,
AutoencoderKL, DDIMScheduler, ...) to the MI355X implementations -- also in modules that already did
``from ... import Name`` -- so everything else the script imports (``DDPMScheduler``, ``is_xformers_available``,
``get_K_R``, ``e2p``, ``save_videos_grid``, ``zero_rank_print``, ``get_anchor_target``, ``flush``, the pano_utils
warps) keeps coming from the checkout.  Only when a real module cannot be imported (its third-party dependency is
absent on the AMD box, e.g. xformers for src.modules.transformer) is a synthetic module with the hot-path names
registered under that name; its parent packages stay the checkout's.  ``uninstall()`` restores everything.
"""
import importlib
import importlib.machinery
import importlib.util
import os
import sys
import types

from . import mv_model, pano_geometry, pipeline, scheduler, synthetic, unet3d, vae


def _flush():
    """src/modules/utils.py:flush = gc.collect + torch.cuda.empty_cache: the MI355X path keeps its allocations."""


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
    "src.modules.utils": dict(flush=_flush),
    "src.utils.pano": dict(pad_pano=pano_geometry.pad_pano, unpad_pano=pano_geometry.unpad_pano),
    "diffusers": dict(AutoencoderKL=vae.AutoencoderKL, DDIMScheduler=scheduler.DDIMScheduler),
}
# ``is_xformers_available`` is answered per CALLER: the reference script and the checkout's own packages see True (the
# xformers code path -- logit scale d^-1/2 in IPCrossAttention -- is what the kernels implement), everything else (the real
# diffusers, any third party) keeps the library's own answer: patching the function globally made every later
# ``if is_xformers_available(): import xformers`` at module scope fail on a box without xformers.
_XFORMERS_CALLERS = ("__main__", "inference_dual_p2e", "animatediff", "src")


def _scoped_xformers_probe(original):
    def is_xformers_available():
        caller = sys._getframe(1).f_globals.get("__name__", "")
        if caller.split(".")[0] in _XFORMERS_CALLERS:
            return True
        return bool(original()) if callable(original) else False
    is_xformers_available.__im360_alias__ = True
    return is_xformers_available
# names a synthetic module must also carry when the real one cannot be imported (the script imports them by name)
_FALLBACK_EXTRAS = {
    "src.utils.pano": dict(icosahedron_sample_camera=synthetic.icosahedron_angles),
}

_MISSING = object()
_undo = []            # (module, name, previous value | _MISSING)
_synthetic = []       # names of modules this file registered


def _rebind(mod, name, value):
    _undo.append((mod, name, getattr(mod, name, _MISSING)))
    setattr(mod, name, value)


def _import_real(name):
    """The checkout's module, or None when it (or a third-party package it needs) cannot be imported."""
    try:
        return importlib.import_module(name)
    except Exception:           # ImportError of a dependency, or an error raised while the module initialises
        sys.modules.pop(name, None)
        return None


def _synthetic_module(name):
    """Register an empty module ``name`` whose parent packages are the real ones when they import, else namespace
    shells that still point at the checkout's directory (so sibling submodules keep importing from it)."""
    parts = name.split(".")
    for i in range(1, len(parts)):
        pkg = ".".join(parts[:i])
        if pkg in sys.modules:
            continue
        if _import_real(pkg) is None:
            shell = types.ModuleType(pkg)
            paths = [os.path.join(p, *parts[:i]) for p in sys.path if os.path.isdir(os.path.join(p, *parts[:i]))]
            shell.__path__ = paths
            shell.__spec__ = importlib.machinery.ModuleSpec(pkg, None, is_package=True)
            shell.__im360_alias__ = True
            sys.modules[pkg] = shell
            _synthetic.append(pkg)
            if i > 1:
                _rebind(sys.modules[".".join(parts[:i - 1])], parts[i - 1], shell)
    mod = types.ModuleType(name)
    mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
    mod.__im360_alias__ = True
    sys.modules[name] = mod
    _synthetic.append(name)
    if len(parts) > 1:
        _rebind(sys.modules[".".join(parts[:-1])], parts[-1], mod)
    return mod


def _preprocess_aliases():
    """Opt-in (``install(preprocess=True)``): the host preprocessing geometry of SURVEY row N3 on the GPU.  Not part of the
    default overlay because the bicubic arithmetic of cv2.remap is restated without OpenCV to check it against."""
    from . import preprocess
    return {
        "src.utils.pano_utils.Equirec2Perspec": dict(Equirectangular=preprocess.Equirectangular),
        "src.utils.pano_utils.Perspec2Equirec": dict(Perspective=preprocess.Perspective),
        "animatediff.utils.video_mask": dict(get_anchor_target=preprocess.get_anchor_target),
        "src.modules.utils": dict(flush=_flush, get_maxrec_cord=preprocess.get_maxrec_cord),
    }


def install(force=False, preprocess=False):
    """Overlay the hot-path names (see the module docstring).  Returns {module name: "overlay" | "synthetic"}.
    ``force`` is accepted for backward compatibility; already-imported reference modules are patched in place.
    ``preprocess``: also route the script's E2P / P2E warps, get_anchor_target and get_maxrec_cord to the MI355X versions."""
    done = {}
    originals = {}
    aliases = dict(_ALIASES)
    if preprocess:
        aliases.update(_preprocess_aliases())
    for name, attrs in aliases.items():
        real = sys.modules.get(name)
        if real is None or getattr(real, "__im360_alias__", False):
            real = _import_real(name) if real is None else real
        if real is not None and not getattr(real, "__im360_alias__", False):
            for k, v in attrs.items():
                old = getattr(real, k, _MISSING)
                if old is not _MISSING and old is not v:
                    originals[id(old)] = v
                _rebind(real, k, v)
            done[name] = "overlay"
        else:
            mod = real if real is not None else _synthetic_module(name)
            for k, v in {**attrs, **_FALLBACK_EXTRAS.get(name, {})}.items():
                setattr(mod, k, v)
            done[name] = "synthetic"
    # the xformers probe: one scoped function on the module that defines it (later ``from ... import`` pick it up); modules
    # that already imported the original keep it unless they are the checkout's own
    iu = sys.modules.get("diffusers.utils.import_utils") or _import_real("diffusers.utils.import_utils")
    if iu is None or getattr(iu, "__im360_alias__", False):
        # no diffusers on this box: a synthetic module carrying only the probe (False for everybody but the script's side)
        iu = iu if iu is not None else _synthetic_module("diffusers.utils.import_utils")
        iu.is_xformers_available = _scoped_xformers_probe(None)
        done["diffusers.utils.import_utils"] = "synthetic"
    elif not getattr(getattr(iu, "is_xformers_available", None), "__im360_alias__", False):
        orig = getattr(iu, "is_xformers_available", None)
        probe = _scoped_xformers_probe(orig)
        _rebind(iu, "is_xformers_available", probe)
        done["diffusers.utils.import_utils"] = "scoped probe"
        for mname, mod in list(sys.modules.items()):
            if isinstance(mod, types.ModuleType) and mname.split(".")[0] in _XFORMERS_CALLERS and orig is not None \
                    and vars(mod).get("is_xformers_available") is orig:
                _rebind(mod, "is_xformers_available", probe)
    # modules that already bound the originals by ``from x import Name``
    for mname, mod in list(sys.modules.items()):
        if mod is None or mname.startswith("imagine360_amd") or not isinstance(mod, types.ModuleType):
            continue
        try:
            items = list(vars(mod).items())
        except Exception:
            continue
        for k, v in items:
            new = originals.get(id(v))
            if new is not None and isinstance(v, (type, types.FunctionType)):
                _rebind(mod, k, new)
    return done


def uninstall():
    while _undo:
        mod, name, old = _undo.pop()
        if old is _MISSING:
            if hasattr(mod, name):
                delattr(mod, name)
        else:
            setattr(mod, name, old)
    while _synthetic:
        sys.modules.pop(_synthetic.pop(), None)


def main(argv=None):
    """``python -m imagine360_amd.dropin <script.py> [args...]``: run an unmodified reference script on the overlay."""
    import runpy
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        raise SystemExit("usage: python -m imagine360_amd.dropin <reference script> [its arguments]")
    script = os.path.abspath(argv[0])
    sys.path.insert(0, os.path.dirname(script))           # the checkout root, like `python script.py` would
    sys.argv = [script] + argv[1:]
    print("imagine360_amd.dropin:", install(), file=sys.stderr)
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
