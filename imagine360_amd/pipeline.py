"""AnimationPipeline: the DDIM loop that drives the dual-branch model
(animatediff/pipelines/pipeline_animation_inference_dual.py:60-824), same constructor and ``__call__``
surface, rebuilt for the MI355X:

  * the loop body is MultiViewBaseModel.forward (HIP kernels) + ONE fused CFG+DDIM kernel per branch;
    no per-step empty_cache()/flush() syncs, no host round trips (timesteps are host ints);
  * conditioning that the reference recomputes is built once: text embeddings, SAM features, masked
    latents, the nearest-E2P index maps of init_noise, the pad-4 latent for the decode;
  * RNG: ``rng="host"`` draws every Gaussian (init noise, VAE posterior samples, per-step IP noise) from
    the CPU generator in the reference's order, so results are comparable to the reference's CPU path
    on the same seeds; ``rng="device"`` (default) draws on the GPU like the reference's GPU path.

CLIP text encoding and SAM feature extraction are outside the hot path: the
pipeline uses ``text_encoder``/``tokenizer``/``image_encoder`` when given, and also accepts precomputed
``prompt_embeds`` / ``sam_features`` keyword arguments.
"""
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

from . import kernels
from . import pano_geometry as G

VAE_SCALE = 0.18215        # hard-coded in the reference pipeline (:303, :440, :465)


@dataclass
class AnimationPipelineOutput:
    videos: torch.Tensor


class AnimationPipeline:
    def __init__(self, vae, text_encoder, tokenizer, pers_unet, pano_unet, mv_base_model, scheduler,
                 image_encoder=None, image_encoder_name="CLIP"):
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.pers_unet, self.pano_unet, self.mv_base_model, self.scheduler = pers_unet, pano_unet, mv_base_model, scheduler
        self.image_encoder, self.image_encoder_name = image_encoder, image_encoder_name
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)
        self.SAMpredictor = self.SAMProcessor = None
        if image_encoder_name == "SAM" and image_encoder is not None:
            from segment_anything import SamPredictor
            self.SAMpredictor = SamPredictor(image_encoder)
            self.SAMProcessor = self.SAMpredictor.transform
        self.rng = "device"
        self.use_graph = True       # replay one captured hipGraph per step (device RNG only; see graph_step.py)
        self._device = None

    # ---- reference surface ------------------------------------------------------------------------
    def to(self, device):
        self._device = torch.device(device)
        for m in (self.vae, self.text_encoder, self.mv_base_model, self.image_encoder):
            if isinstance(m, torch.nn.Module):
                m.to(device)
        return self

    @property
    def device(self):
        return self._device or self.vae.device

    _execution_device = device

    def enable_vae_slicing(self):
        self.vae.enable_slicing()

    def disable_vae_slicing(self):
        self.vae.disable_slicing()

    def progress_bar(self, iterable=None, total=None):
        from tqdm.auto import tqdm
        return tqdm(iterable, total=total, disable=getattr(self, "_no_progress", False))

    # ---- conditioning producers (outside the hot path) ----------------------------------------------
    def _encode_prompt(self, prompt, device, num_videos_per_prompt, do_classifier_free_guidance, negative_prompt):
        """CLIP text embeddings, uncond first (pipeline...dual.py:210-297)."""
        if self.text_encoder is None or self.tokenizer is None:
            raise RuntimeError("no text_encoder/tokenizer: pass prompt_embeds=(pano [2,77,d], pers [2m,77,d])")
        tok = lambda s: self.tokenizer(s, padding="max_length", max_length=self.tokenizer.model_max_length,
                                       truncation=True, return_tensors="pt").input_ids.to(device)
        emb = self.text_encoder(tok(prompt))[0]
        emb = emb.repeat_interleave(num_videos_per_prompt, dim=0)
        if do_classifier_free_guidance:
            neg = negative_prompt if negative_prompt is not None else [""] * len(prompt)
            neg = [neg] * len(prompt) if isinstance(neg, str) else ["" if n is None else n for n in neg]   # None -> "" like the reference
            un = self.text_encoder(tok(neg))[0].repeat_interleave(num_videos_per_prompt, dim=0)
            emb = torch.cat([un, emb])
        return emb

    def _sam_features(self, anchor):
        """anchor [1, F, 3, h, w] in [-1, 1] -> [1, F, 4096, 256] (pipeline...dual.py:671-718)."""
        if self.SAMpredictor is None:
            raise RuntimeError("no SAM image_encoder: pass sam_features=(pano [1,F,4096,256], pers [1,F,4096,256])")
        imgs = np.uint8(((anchor.float() + 1.0) / 2.0 * 255).squeeze(0).cpu().numpy().transpose(0, 2, 3, 1))
        ts = torch.stack([torch.as_tensor(self.SAMProcessor.apply_image(im), device=anchor.device).permute(2, 0, 1).contiguous()
                          for im in imgs])
        assert ts.shape[0] % 8 == 0
        feats = []
        for i in range(0, ts.shape[0], 8):
            self.SAMpredictor.set_torch_image(ts[i:i + 8], ts[0].shape[:2])
            feats.append(self.SAMpredictor.get_image_embedding().flatten(2).transpose(1, 2))
        return torch.cat(feats).unsqueeze(0)

    # ---- pieces of the hot path -------------------------------------------------------------------
    def _randn(self, shape, device, dtype=torch.float32):
        if self.rng == "host":
            return torch.randn(shape, dtype=torch.float32).to(device=device, dtype=dtype)
        return torch.randn(shape, device=device, dtype=dtype)

    def init_noise(self, bs, video_length, equi_h, equi_w, pers_h, pers_w, cameras, device, latents_dtype=torch.float16):
        """One panorama noise; the perspective noise is its nearest-neighbour E2P resampling, so both branches
        start from consistent noise (pipeline...dual.py:361-387).  The index maps are frame-invariant and
        built once instead of 16 x 20 host-side map builds."""
        pano = self._randn((bs, video_length, 1, 4, equi_h, equi_w), device)
        idx, ok = G.nearest_e2p_index(equi_h, equi_w, pers_h, pers_w, cameras)            # [m, ph, pw]
        idx, ok = idx.to(device), ok.to(device)
        flat = pano.squeeze(2).reshape(bs, video_length, 4, equi_h * equi_w)
        pers = flat[..., idx.reshape(-1)].reshape(bs, video_length, 4, *idx.shape) * ok   # b f c m h w
        return (pano.squeeze(2).permute(0, 2, 1, 3, 4).contiguous().to(latents_dtype),
                pers.permute(0, 3, 2, 1, 4, 5).contiguous().to(latents_dtype))

    def _encode_chunks(self, x, chunk=8, keep_rows=None):
        """VAE-encode images [n, 3, H, W] in chunks of ``chunk`` and sample the posteriors (one randn per chunk, in order).
        ``keep_rows`` (lo, hi): only the rows in [lo, hi) are needed (frame-sharded runs): chunks outside that range are
        not encoded -- their posterior noise is still DRAWN (same shapes, same order) and dropped, so the kept rows get
        exactly the samples of the unsharded run.  Returns the latents of all rows (keep_rows None) or of [lo, hi)."""
        self.vae.sample_on_host = self.rng == "host"
        out = []
        for i in range(0, x.shape[0], chunk):
            xs = x[i:i + chunk]
            n = xs.shape[0]
            if keep_rows is not None and (i + n <= keep_rows[0] or i >= keep_rows[1]):
                shape = (n, 4, xs.shape[-2] // 8, xs.shape[-1] // 8)
                if self.vae.sample_on_host:
                    torch.randn(shape, dtype=torch.float32)
                else:
                    torch.randn(shape, device=x.device)
                continue
            lat = self.vae.encode(xs, n).latent_dist.sample()
            if keep_rows is not None:
                lat = lat[max(keep_rows[0] - i, 0):max(min(keep_rows[1] - i, n), 0)]
            out.append(lat)
        return torch.cat(out)

    def prepare_masked_latents_pano(self, video_length, pix_masked, pano_mask, keep=None):
        """(:427-448) pix [b f c h w] -> latents [b c f h w]; mask nearest-resized to latent resolution.
        ``keep`` (first frame, count): encode only these frames (b == 1); the mask is returned for all frames."""
        b = pix_masked.shape[0]
        rows = None if keep is None else (keep[0], keep[0] + keep[1])
        fl = video_length if keep is None else keep[1]
        assert keep is None or b == 1
        lat = self._encode_chunks(pix_masked.reshape(b * video_length, *pix_masked.shape[2:]), keep_rows=rows)
        lat = lat.reshape(b, fl, *lat.shape[1:]).permute(0, 2, 1, 3, 4) * VAE_SCALE
        mask = pano_mask.transpose(2, 1)
        mask = F.interpolate(mask, size=(mask.shape[2], lat.shape[-2], lat.shape[-1]))
        return lat, mask.to(lat.device)

    def prepare_masked_latents_pers(self, video_length, pix_masked, pers_masks, keep=None):
        """(:451-473) pix [b f m c h w] -> latents [b m c f h w]; ``keep`` as in ``prepare_masked_latents_pano``."""
        b, _, m = pix_masked.shape[:3]
        rows = None if keep is None else (keep[0] * m, (keep[0] + keep[1]) * m)
        fl = video_length if keep is None else keep[1]
        assert keep is None or b == 1
        lat = self._encode_chunks(pix_masked.reshape(b * video_length * m, *pix_masked.shape[3:]), keep_rows=rows)
        lat = lat.reshape(b, fl, m, *lat.shape[1:]).permute(0, 2, 3, 1, 4, 5) * VAE_SCALE
        mk = pers_masks.permute(0, 3, 1, 2, 4, 5).squeeze(0)
        mk = F.interpolate(mk, size=(m, lat.shape[-2], lat.shape[-1])).unsqueeze(3)
        return lat, mk.permute(0, 2, 3, 1, 4, 5).to(lat.device)

    def decode_latents(self, latents, frames_per_call=8):
        """(:301-313) VAE decode -> float32 numpy in [0, 1], [b, 3, f, H, W].  The reference decodes frame by frame; every
        VAE op is per image (GroupNorm, conv, the mid-block attention), so decoding ``frames_per_call`` frames per call
        gives the same numbers with 1/8 of the launches."""
        b, c, f, h, w = latents.shape
        z = (latents / VAE_SCALE).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
        import inspect
        kw = {"batched": True} if "batched" in inspect.signature(self.vae.decode).parameters else {}     # (a user-supplied diffusers VAE has no such keyword)
        frames = [self.vae.decode(z[i:i + frames_per_call].to(self.vae.dtype), **kw).sample
                  for i in range(0, z.shape[0], frames_per_call)]
        video = torch.cat(frames).reshape(b, f, 3, h * 8, w * 8).permute(0, 2, 1, 3, 4)
        return (video / 2 + 0.5).clamp(0, 1).cpu().float().numpy()

    def padding_pano(self, pano, padding=4, latent=False):
        return G.pad_pano(pano, padding if latent else padding * 8)

    def unpadding_pano(self, pano_pad, padding=4, latent=False):
        return G.unpad_pano(pano_pad, padding if latent else padding * 8)

    # ---- the loop ------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt, num_inference_steps=50, guidance_scale_text=7.5, guidance_scale_adapter=7.5,
                 negative_prompt=None, num_videos_per_prompt=1, eta=0.0, generator=None, latents=None,
                 output_type="tensor", return_dict=True, callback=None, callback_steps=1, latents_dtype=torch.float16,
                 video_batch=None, use_outpaint=False, use_ip_plus_cross_attention=False, use_fps_condition=False,
                 ip_plus_condition="image", prompt_embeds=None, sam_features=None, trace=None, frame_shard=None,
                 **kwargs):
        """``frame_shard`` (imagine360_amd.dist.FrameShard): this rank denoises a contiguous chunk of the frames (BASELINE
        configs 4 / 5); all ranks must be called with the same seeds and inputs.  Noise is drawn for the whole clip and
        cut, the VAE encodes / the loop runs / the VAE decodes only the local frames, the motion modules exchange tokens
        with one all-to-all each way, and the decoded frames are all-gathered at the end (the only other collective)."""
        device = self.device
        vb = video_batch
        assert use_outpaint and use_ip_plus_cross_attention, "the dual pipeline runs with use_outpaint and the IP adapter"
        cfg = guidance_scale_text > 1.0
        assert cfg, "the reference only binds its model inputs under classifier-free guidance (:744-751)"
        pano_pix, pano_mask = vb["pano_pixel_values"], vb["pano_mask"]
        pers_pix, pers_masks = vb["pers_pixel_values"], vb["pers_masks"]
        cameras, f = vb["cameras"], vb["video_length"]
        m = pers_pix.shape[2]
        H, W, ps = vb["pano_H"], vb["pano_W"], vb["pers_size"]
        self.mv_base_model.noise_on_host = self.rng == "host"

        pano_pix_masked = (pano_pix * (pano_mask < 0.5)).to(device)
        pers_pix_masked = (pers_pix * (pers_masks < 0.5)).to(device)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        steps_host = self.scheduler._timesteps_host
        pano_latent, pers_latent = self.init_noise(1, f, H // 8, W // 8, ps // 8, ps // 8, cameras, device, latents_dtype)
        sh = frame_shard
        if sh is not None:
            # every rank drew the whole clip's noise from the same seed: cut to the local frames
            pano_latent, pers_latent = sh.take(pano_latent, 2).contiguous(), sh.take(pers_latent, 3).contiguous()
        # frame-sharded: only this rank's frames go through the VAE encoder; the posterior noise of the other frames is still
        # drawn (and dropped), so the samples are those of the unsharded run
        keep = None if sh is None else (sh.f0, sh.local)
        pano_ml, pano_mask_l = self.prepare_masked_latents_pano(f, pano_pix_masked, pano_mask.to(device), keep)
        pers_ml, pers_mask_l = self.prepare_masked_latents_pers(f, pers_pix_masked, pers_masks.to(device), keep)
        if sh is not None:
            pano_mask_l, pers_mask_l = sh.take(pano_mask_l, 2), sh.take(pers_mask_l, 3)
            self.mv_base_model.set_frame_shard(sh)
        try:
            if prompt_embeds is not None:
                text_pano, text_pers = prompt_embeds
            else:
                text_pano = self._encode_prompt([prompt], device, num_videos_per_prompt, cfg, [negative_prompt])
                text_pers = text_pano.repeat_interleave(m, dim=0)     # the reference encodes the same prompt m times (:628, :655)
            text_pano, text_pers = text_pano.to(device, latents_dtype), text_pers.to(device, latents_dtype)
            if sam_features is not None:
                sam_pano, sam_pers = sam_features
            else:
                sam_pano = self._sam_features(vb["anchor_pixels_values"].to(device))
                sam_pers = self._sam_features(vb["anchor_pixels_values_pers"].to(device))
            feat_pano = torch.cat([sam_pano, sam_pano]).to(device, latents_dtype)
            feat_pers = torch.cat([sam_pers, sam_pers]).to(device, latents_dtype).unsqueeze(1).expand(-1, m, -1, -1, -1)
            fps = torch.tensor(vb["fps"], device=device).unsqueeze(0)
            fps_pano = torch.cat([fps] * 2) if use_fps_condition else None
            fps_pers = torch.cat([fps.unsqueeze(-1).repeat(1, m)] * 2) if use_fps_condition else None
            rel = torch.cat([vb["relative_position"].to(device).unsqueeze(0)] * 2)
            pitch = torch.cat([vb["pitchs"].to(device).unsqueeze(0)] * 2)
            ts_dev = [torch.tensor([t], dtype=torch.int64, device=device) for t in steps_host]
            dt = latents_dtype
            # static halves of the model input: mask + masked latent (channels 4..8), duplicated for CFG
            in_pano = torch.cat([torch.cat((pano_latent, pano_mask_l.to(dt), pano_ml.to(dt)), dim=1)] * 2)
            in_pers = torch.cat([torch.cat((pers_latent, pers_mask_l.to(dt), pers_ml.to(dt)), dim=2)] * 2)

            graphed = None
            import torch.distributed as tdist
            capturable = sh is None or (tdist.is_initialized() and tdist.get_backend(sh.group) == "nccl")     # RCCL all-to-alls are stream ops
            if self.use_graph and self.rng == "device" and pano_latent.is_cuda and trace is None and callback is None and capturable:
                from .graph_step import GraphedDenoiseStep
                inputs = dict(latents=in_pers, pano_latent=in_pano, prompt_embd=text_pers, pano_prompt_embd=text_pano,
                              fps_tensor_pano=fps_pano, fps_tensor_pers=fps_pers, reference_images_clip_feat_pano=feat_pano,
                              reference_images_clip_feat_pers=feat_pers, relative_position_tensor=rel, pitchs_tensor=pitch)
                graphed = GraphedDenoiseStep(self.mv_base_model, self.scheduler, inputs, cameras, pano_latent, pers_latent,
                                             guidance_scale_text, use_fps=use_fps_condition, warmup=1)       # one eager step fills every cache
            for i, t in enumerate(self.progress_bar(steps_host)):
                if graphed is not None:
                    pano_latent, pers_latent = graphed.step(t)
                    continue
                in_pano[:, :4] = pano_latent
                in_pers[:, :, :4] = pers_latent
                pred_pers, pred_pano = self.mv_base_model(
                    latents=in_pers, pano_latent=in_pano, timestep=ts_dev[i], prompt_embd=text_pers,
                    pano_prompt_embd=text_pano, cameras=cameras, use_fps_condition=use_fps_condition,
                    use_ip_plus_cross_attention=use_ip_plus_cross_attention, fps_tensor_pano=fps_pano, fps_tensor_pers=fps_pers,
                    reference_images_clip_feat_pano=feat_pano, reference_images_clip_feat_pers=feat_pers,
                    relative_position_tensor=rel, pitchs_tensor=pitch)
                pano_latent = self._cfg_step(pred_pano, guidance_scale_text, t, pano_latent)
                pers_latent = self._cfg_step(pred_pers, guidance_scale_text, t, pers_latent)
                if trace is not None:
                    trace.append(pano_latent.clone())
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, pano_latent)

            video = self.decode_latents(self.padding_pano(pano_latent, latent=True))
            video = self.unpadding_pano(video)
            if sh is not None:                  # latent / video boundary: the only collective besides the motion-module exchanges
                self.mv_base_model.set_frame_shard(None)
                video = sh.gather_frames(torch.from_numpy(np.ascontiguousarray(video)).to(device), 2).cpu().numpy()
                pano_latent = sh.gather_frames(pano_latent, 2)
                pers_latent = sh.gather_frames(pers_latent, 3)
            if output_type == "tensor":
                video = torch.from_numpy(np.ascontiguousarray(video))
            self.last_latents = (pano_latent, pers_latent)
            return AnimationPipelineOutput(videos=video) if return_dict else video
        finally:
            if sh is not None:
                self.mv_base_model.set_frame_shard(None)      # also when the loop raises: the model must not stay sharded

    def _cfg_step(self, pred, g, t, latent):
        u, c = pred.to(latent.dtype).chunk(2)          # latents_dtype may differ from the model dtype (the reference promotes)
        return self.scheduler.fused_cfg_step(u, c, g, t, latent)
