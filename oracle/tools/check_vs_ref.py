"""Check im360_oracle against the real reference (authoring container only)."""
import random, sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_build as RB
from im360_oracle.cfg import sd21_unet_cfg
from im360_oracle import mv as OMV

def rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()

def main(width_div=5, frames=16):
    torch.set_grad_enabled(False)
    cfg = sd21_unet_cfg(width_div)
    t0 = time.time(); mv = RB.ref_mv(cfg); print("ref build", time.time() - t0)
    sd = {k: v for k, v in mv.state_dict().items()}
    inp = RB.mv_inputs(frames=frames, pano_hw=(32, 64), pers_hw=(16, 16), seed=0)
    cams = RB.icosahedron_cameras(90, 128)
    torch.manual_seed(7); random.seed(7)
    t0 = time.time()
    ref_pers, ref_pano = mv(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True, **inp)
    print("ref fwd", time.time() - t0)
    torch.manual_seed(7); random.seed(7)
    t0 = time.time()
    o_pers, o_pano = OMV.mv_forward(sd, cfg, inp["latents"], inp["pano_latent"], inp["timestep"], inp["prompt_embd"],
        inp["pano_prompt_embd"], cams, inp["fps_tensor_pano"], inp["fps_tensor_pers"],
        inp["reference_images_clip_feat_pano"], inp["reference_images_clip_feat_pers"],
        inp["relative_position_tensor"], inp["pitchs_tensor"], mask_cache={})
    print("oracle fwd", time.time() - t0)
    print("pers rel", rel(o_pers, ref_pers), "pano rel", rel(o_pano, ref_pano), ref_pano.abs().mean().item(), ref_pano.std().item())

if __name__ == "__main__":
    main()
