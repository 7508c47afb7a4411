"""Functional fp32 restatement of AnimationPipeline.__call__
(animatediff/pipelines/pipeline_animation_inference_dual.py:553-824) from the point where
conditioning tensors exist: text embeddings (CLIP is out of scope) and
SAM features are inputs.  TEST INFRASTRUCTURE (see package docstring).

RNG draw order reproduced from the reference (SURVEY.md section 5 'RNG'): init_noise randn,
one randn per VAE-encode chunk (pano chunks, then pers chunks), then per step
randn_like x2 and random.random() x7 inside mv_forward.
"""
import torch
import torch.nn.functional as F

from . import ddim as D
from . import geometry as G
from . import mv as MV
from . import vae as V

VAE_SCALE = 0.18215          # hard-coded in the pipeline (:303, :440, :465)


def init_noise(bs, f, eh, ew, ph, pw, cameras):
    """pipeline...dual.py:361-387: one pano noise; perspective noise = nearest E2P of it."""
    fov, theta, phi = (G._cam_list(cameras, k) for k in ("FoV", "theta", "phi"))
    m = len(fov)
    pano = torch.randn(bs, f, 1, 4, eh, ew)
    pano_out = pano.squeeze(2).permute(0, 2, 1, 3, 4)                       # b c f h w
    per_frame = []
    for i in range(f):
        src = pano[:, i].expand(-1, m, -1, -1, -1).reshape(bs * m, 4, eh, ew)
        nz = G.e2p(src, fov * bs, theta * bs, phi * bs, (ph, pw), mode="nearest")
        per_frame.append(nz.reshape(bs, m, 4, ph, pw))
    pers = torch.stack(per_frame, dim=0).permute(1, 2, 3, 0, 4, 5)          # b m c f h w
    return pano_out, pers


def _encode_chunks(sd_vae, vcfg, x, chunk=8):
    out = []
    for i in range(0, x.shape[0], chunk):
        mom = V.encode_moments(sd_vae, vcfg, x[i:i + chunk])
        out.append(V.sample_posterior(mom, torch.randn(mom.shape[0], mom.shape[1] // 2, *mom.shape[2:])))
    return torch.cat(out)


def masked_latents_pano(sd_vae, vcfg, f, pix_masked, mask):
    """prepare_masked_latents_pano (:427-448).  pix [b f c h w], mask [b f 1 h w]."""
    b = pix_masked.shape[0]
    lat = _encode_chunks(sd_vae, vcfg, pix_masked.reshape(b * f, *pix_masked.shape[2:]))
    lat = lat.reshape(b, f, *lat.shape[1:]).permute(0, 2, 1, 3, 4) * VAE_SCALE
    mask = mask.transpose(2, 1)
    mask = F.interpolate(mask, size=(mask.shape[2], lat.shape[-2], lat.shape[-1]))
    return lat, mask


def masked_latents_pers(sd_vae, vcfg, f, pix_masked, masks):
    """prepare_masked_latents_pers (:451-473).  pix [b f m c h w], masks [b f m 1 h w]."""
    b, _, m = pix_masked.shape[:3]
    lat = _encode_chunks(sd_vae, vcfg, pix_masked.reshape(b * f * m, *pix_masked.shape[3:]))
    lat = lat.reshape(b, f, m, *lat.shape[1:]).permute(0, 2, 3, 1, 4, 5) * VAE_SCALE   # b m c f h w
    mk = masks.permute(0, 3, 1, 2, 4, 5).squeeze(0)                                     # [c, f, m, h, w]
    mk = F.interpolate(mk, size=(m, lat.shape[-2], lat.shape[-1])).unsqueeze(3)          # b f m c h w
    return lat, mk.permute(0, 2, 3, 1, 4, 5)


def decode_latents(sd_vae, vcfg, latents):
    """decode_latents (:301-313): per-frame decode, /0.18215, (x/2+.5).clamp(0,1)."""
    b, c, f, h, w = latents.shape
    z = (latents / VAE_SCALE).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    frames = [V.decode(sd_vae, vcfg, z[i:i + 1]) for i in range(z.shape[0])]
    vid = torch.cat(frames).reshape(b, f, 3, h * 8, w * 8).permute(0, 2, 1, 3, 4)
    return (vid / 2 + 0.5).clamp(0, 1)


def run(sd_mv, ucfg, sd_vae, vcfg, video_batch, text_pano, text_pers, sam_pano, sam_pers,
        num_inference_steps=4, guidance_scale=7.5, mask_cache=None, trace=None):
    """AnimationPipeline.__call__ (:553-824).  text_* already CFG-stacked ([2,77,d] / [2m,77,d],
    uncond first); sam_* [1, Fs, 4096, 256].  Returns videos [1,3,F,H,W] in [0,1]."""
    vb = video_batch
    cameras, f = vb["cameras"], vb["video_length"]
    H, W, ps = vb["pano_H"], vb["pano_W"], vb["pers_size"]
    m = vb["pers_pixel_values"].shape[2]
    pano_pix = vb["pano_pixel_values"] * (vb["pano_mask"] < 0.5)
    pers_pix = vb["pers_pixel_values"] * (vb["pers_masks"] < 0.5)
    acp = D.alphas_cumprod()
    ts = D.timesteps(num_inference_steps)
    pano_latent, pers_latent = init_noise(1, f, H // 8, W // 8, ps // 8, ps // 8, cameras)
    pano_ml, pano_mask = masked_latents_pano(sd_vae, vcfg, f, pano_pix, vb["pano_mask"])
    pers_ml, pers_mask = masked_latents_pers(sd_vae, vcfg, f, pers_pix, vb["pers_masks"])
    feat_pano = torch.cat([sam_pano, sam_pano])
    feat_pers = torch.cat([sam_pers, sam_pers]).unsqueeze(1).expand(-1, m, -1, -1, -1)
    fps = torch.tensor(vb["fps"]).unsqueeze(0)
    fps_pano = torch.cat([fps] * 2)
    fps_pers = torch.cat([fps.unsqueeze(-1).repeat(1, m)] * 2)
    rel = torch.cat([vb["relative_position"].unsqueeze(0)] * 2)
    pitch = torch.cat([vb["pitchs"].unsqueeze(0)] * 2)
    if mask_cache is None:
        mask_cache = {}
    for t in ts:
        in_pano = torch.cat([torch.cat((pano_latent, pano_mask, pano_ml), dim=1)] * 2)
        in_pers = torch.cat([torch.cat((pers_latent, pers_mask, pers_ml), dim=2)] * 2)
        pred_pers, pred_pano = MV.mv_forward(
            sd_mv, ucfg, in_pers, in_pano, t.unsqueeze(0), text_pers, text_pano, cameras, fps_pano, fps_pers,
            feat_pano, feat_pers, rel, pitch, mask_cache=mask_cache)
        u, c = pred_pano.chunk(2)
        pred_pano = u + guidance_scale * (c - u)
        u, c = pred_pers.chunk(2)
        pred_pers = u + guidance_scale * (c - u)
        pano_latent = D.step_v(pred_pano, t, pano_latent, acp, num_inference_steps)
        pers_latent = D.step_v(pred_pers, t, pers_latent, acp, num_inference_steps)
        if trace is not None:
            trace.append(pano_latent.clone())
    video = decode_latents(sd_vae, vcfg, G.pad_pano(pano_latent, 4))
    return G.unpad_pano(video, 32), pano_latent, pers_latent
