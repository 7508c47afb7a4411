"""Run in a SUBPROCESS by tests/test_host_logic.py::test_dropin_overlay_runs_the_reference_script (build container only:
needs the reference checkout at /root/reference).  Imports the reference with third-party stand-ins
(oracle/tools/ref_shims.py), overlays imagine360_amd (dropin.install()), then executes the UNMODIFIED reference script
inference_dual_p2e.py as a module -- its whole import block (lines 1-45) and every function definition -- and drives its
own model-loading code (load_unetbranch, unet_load_diffusers_lora, the object construction of lines 382-474) at reduced
width on CPU.  Prints a JSON summary on the last line."""
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "tools"), HERE):
    sys.path.insert(0, p)

import torch  # noqa: E402

import ref_shims  # noqa: E402

ref_shims.install()                       # third-party stand-ins + /root/reference on sys.path + vendored diffusers


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _OmegaConf:
    @staticmethod
    def to_container(x, **_):
        return dict(x)

    @staticmethod
    def load(path):
        raise RuntimeError("offline harness")


_stub("omegaconf", OmegaConf=_OmegaConf)
_stub("geocalib", GeoCalib=object)
sys.modules["loguru"].logger = types.SimpleNamespace(info=lambda *a, **k: None, remove=lambda *a, **k: None, add=lambda *a, **k: None)
tv = sys.modules["torchvision.transforms"]
tv.Compose = lambda fs: (lambda x: x)
tv.Normalize = lambda **k: (lambda x: x)
import cv2  # noqa: E402  (stand-in)

for k, v in dict(imread=None, resize=None, cvtColor=None, COLOR_BGR2RGB=4, COLOR_RGB2BGR=4).items():
    if not hasattr(cv2, k):
        setattr(cv2, k, v)

from imagine360_amd import configs, dropin, mv_model, pipeline, scheduler, unet3d, vae  # noqa: E402
import _emu_kernels as E  # noqa: E402

try:
    from diffusers.utils.import_utils import is_xformers_available as _library_probe
    _library_answer = bool(_library_probe())          # (True under ref_shims: it registers an xformers stand-in)
except Exception:          # noqa: BLE001
    _library_answer = None
how = dropin.install()

src_path = os.path.join(ref_shims.REF_ROOT, "inference_dual_p2e.py")
ref = types.ModuleType("inference_dual_p2e")
ref.__file__ = src_path
exec(compile(open(src_path).read(), src_path, "exec"), ref.__dict__)          # __name__ != "__main__": defines, does not run

out = {"how": how}
# a diffusers submodule imported AFTER the overlay must see the library's own xformers probe (False here: no xformers), not
# the overlay's True -- its module-level `if is_xformers_available(): import xformers` would raise otherwise
import importlib  # noqa: E402

for _m in [m for m in list(sys.modules) if m.startswith("diffusers.models.attention_processor")]:
    del sys.modules[_m]
try:
    _ap = importlib.import_module("diffusers.models.attention_processor")
    out["fresh_diffusers_import"] = {"ok": True, "library_answer": _library_answer,
                                     "probe_for_diffusers": bool(eval("is_xformers_available()", vars(_ap))) if hasattr(_ap, "is_xformers_available") else None}      # (called from inside the module)
except Exception as e:          # noqa: BLE001
    out["fresh_diffusers_import"] = {"ok": False, "error": repr(e)}
out["names"] = {
    "AutoencoderKL": ref.AutoencoderKL is vae.AutoencoderKL,
    "DDIMScheduler": ref.DDIMScheduler is scheduler.DDIMScheduler,
    "DDPMScheduler_from_checkout": ref.DDPMScheduler.__module__.startswith("diffusers."),
    "UNet3DConditionModel": ref.UNet3DConditionModel is unet3d.UNet3DConditionModel,
    "MultiViewBaseModel": ref.MultiViewBaseModel is mv_model.MultiViewBaseModel,
    "AnimationPipeline": ref.AnimationPipeline is pipeline.AnimationPipeline,
    "is_xformers_available": bool(ref.is_xformers_available()),
    "get_K_R_from_checkout": ref.get_K_R.__module__ == "src.utils.pano",
    "save_videos_grid_from_checkout": ref.save_videos_grid.__module__ == "animatediff.utils.util",
    "e2p_from_checkout": callable(ref.e2p),
    "flush_is_noop": ref.flush() is None,
}

torch.set_grad_enabled(False)
wd = 10
with tempfile.TemporaryDirectory() as tmp:
    # a stand-in "pretrained_model_path": SD-2.1-shaped 2-D UNet + VAE directories at reduced width
    os.makedirs(os.path.join(tmp, "unet"))
    os.makedirs(os.path.join(tmp, "vae"))
    ucfg = dict(configs.unet_config(wd))
    json.dump({**ucfg, "_class_name": "UNet2DConditionModel"}, open(os.path.join(tmp, "unet", "config.json"), "w"))
    donor = configs.build_unet(wd)
    from imagine360_amd.weights import fill_module_
    fill_module_(donor)
    sd2d = {k: v.clone() for k, v in donor.state_dict().items() if "motion_modules" not in k and "temporal" not in k}
    sd2d["conv_in.weight"] = sd2d["conv_in.weight"][:, :4].contiguous()       # the 2-D checkpoint has 4 input channels
    torch.save(sd2d, os.path.join(tmp, "unet", "diffusion_pytorch_model.bin"))
    vcfg = dict(configs.vae_config(4))
    json.dump({**vcfg, "_class_name": "AutoencoderKL"}, open(os.path.join(tmp, "vae", "config.json"), "w"))
    v0 = vae.AutoencoderKL(**vcfg)
    torch.save(v0.state_dict(), os.path.join(tmp, "vae", "diffusion_pytorch_model.bin"))
    # a motion LoRA for one attention layer, in the reference's key format
    lora = {}
    tgt = "down_blocks.0.motion_modules.0.temporal_transformer.transformer_blocks.0.attention_blocks.0"
    g = torch.Generator().manual_seed(0)
    c = donor.down_blocks[0].motion_modules[0].temporal_transformer.transformer_blocks[0].attention_blocks[0].to_q.weight.shape[0]
    lora[f"{tgt}.processor.to_q_lora.down.weight"] = torch.randn(4, c, generator=g) * 0.1
    lora[f"{tgt}.processor.to_q_lora.up.weight"] = torch.randn(c, 4, generator=g) * 0.1
    lora[f"{tgt}.processor.to_out_lora.down.weight"] = torch.randn(4, c, generator=g) * 0.1
    lora[f"{tgt}.processor.to_out_lora.up.weight"] = torch.randn(c, 4, generator=g) * 0.1
    torch.save({"state_dict": lora}, os.path.join(tmp, "lora.ckpt"))
    ckpt = {"state_dict": {"module." + k: v for k, v in donor.state_dict().items() if "motion_modules" in k}, "global_step": 1}
    torch.save(ckpt, os.path.join(tmp, "pers_unet.ckpt"))

    kwargs = dict(configs.PROMPT_DUAL_UNET_KWARGS)
    # ---- the reference script's own code, lines 382-474 --------------------------------------------------------------
    noise_scheduler = ref.DDIMScheduler(**ref.OmegaConf.to_container(configs.NOISE_SCHEDULER_KWARGS))
    v = ref.AutoencoderKL.from_pretrained(tmp, subfolder="vae")
    v.requires_grad_(False)
    pers_unet = ref.load_unetbranch(os.path.join(tmp, "lora.ckpt"), 0.5, tmp, torch.float32, True,
                                    os.path.join(tmp, "pers_unet.ckpt"), tmp, kwargs)
    pers_unet.requires_grad_(False)
    pano_unet = ref.load_unetbranch(None, 1.0, tmp, torch.float32, True, "", tmp, kwargs)
    pano_unet.requires_grad_(False)
    mvm = ref.MultiViewBaseModel(pers_unet, pano_unet, pano_pad=True)
    full = mv_model.MultiViewBaseModel(configs.build_unet(wd), configs.build_unet(wd)).state_dict()
    part = {"module." + k: torch.full_like(t, 0.01) for k, t in full.items() if k.startswith("cp_blocks_mid.transformer.attn1.to_out")}
    m_, u_ = mvm.load_state_dict({k.replace("module.", ""): t for k, t in part.items()}, strict=False)
    mvm.requires_grad_(False)
    for mod in (v, pers_unet, pano_unet, mvm):
        mod.to(dtype=torch.float32)
    pipe = ref.AnimationPipeline(pers_unet=pers_unet, pano_unet=pano_unet, mv_base_model=mvm, vae=v, tokenizer=None,
                                 text_encoder=None, scheduler=noise_scheduler, image_encoder=None,
                                 image_encoder_name="SAM").to("cpu")
    pipe.enable_vae_slicing()

    att = pers_unet.down_blocks[0].motion_modules[0].temporal_transformer.transformer_blocks[0].attention_blocks[0]
    want_q = donor.state_dict()[f"{tgt}.to_q.weight"] + 0.5 * lora[f"{tgt}.processor.to_q_lora.up.weight"] @ lora[f"{tgt}.processor.to_q_lora.down.weight"]
    out["built"] = {
        "types": [type(x).__module__ for x in (noise_scheduler, v, pers_unet, mvm, pipe)],
        "conv_in_widened": list(pers_unet.conv_in.weight.shape),
        "conv_in_extra_channels_zero": bool(pers_unet.conv_in.weight[:, 4:].abs().sum() == 0),
        "motion_ckpt_loaded": bool(torch.equal(pers_unet.down_blocks[0].motion_modules[0].temporal_transformer.proj_in.weight,
                                               donor.down_blocks[0].motion_modules[0].temporal_transformer.proj_in.weight)),
        "lora_merged": float((att.to_q.weight - want_q).abs().max()),
        "fused_qkv_sees_lora": float((att.fused_qkv_weight()[:c] - want_q).abs().max()),
        "xformers_flag": bool(pers_unet.down_blocks[0].attentions[0].transformer_blocks[0].attn2._use_memory_efficient_attention_xformers),
        "mv_missing": len(m_), "mv_unexpected": len(u_),
        "slicing": bool(v.use_slicing),
    }
    # ---- the kept single-branch API: UNet3DConditionModel.forward on the script-built model (kernels emulated on CPU) ----
    import random
    from imagine360_amd import synthetic as S
    inp = S.mv_inputs(frames=2, pano_hw=(32, 64), pers_hw=(16, 16), seed=2, sam_frames=16)
    with E.patched_kernels():
        torch.manual_seed(0)
        random.seed(0)
        y = pano_unet(inp["pano_latent"], inp["timestep"], inp["pano_prompt_embd"], use_ip_plus_cross_attention=True,
                      reference_images_clip_feat=inp["reference_images_clip_feat_pano"], use_fps_condition=True,
                      fps_tensor=inp["fps_tensor_pano"]).sample
    out["unet_forward"] = {"shape": list(y.shape), "finite": bool(torch.isfinite(y).all())}
    pano_sd = {k: v.clone() for k, v in pano_unet.state_dict().items()}

dropin.uninstall()
import animatediff.models.unet as real_unet  # noqa: E402

out["uninstalled"] = real_unet.UNet3DConditionModel is not unet3d.UNet3DConditionModel
# the same forward() call on the REFERENCE's UNet3DConditionModel (animatediff/models/unet.py:632-856), same weights
ref_unet = real_unet.UNet3DConditionModel.from_config(
    {**ucfg, "down_block_types": ["CrossAttnDownBlock3D"] * 3 + ["DownBlock3D"],
     "up_block_types": ["UpBlock3D"] + ["CrossAttnUpBlock3D"] * 3}, **kwargs)
miss, unexp = ref_unet.load_state_dict(pano_sd, strict=False)
ref_unet.eval()
for mod in ref_unet.modules():          # the reference's xformers code path (logit scale d^-1/2 in IPCrossAttention), like gen_golden.py
    if type(mod).__name__ == "IPCrossAttention":
        mod._use_memory_efficient_attention_xformers = True
torch.manual_seed(0)
random.seed(0)
y_ref = ref_unet(inp["pano_latent"], inp["timestep"], inp["pano_prompt_embd"], use_ip_plus_cross_attention=True,
                 reference_images_clip_feat=inp["reference_images_clip_feat_pano"], use_fps_condition=True,
                 fps_tensor=inp["fps_tensor_pano"]).sample
out["unet_forward"].update(rel_vs_reference_forward=float((y - y_ref).norm() / y_ref.norm()),
                           ref_missing=len(miss), ref_unexpected=len(unexp))
print(json.dumps(out))
