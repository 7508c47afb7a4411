"""Functional fp32 restatement of MultiViewBaseModel.forward and WarpAttn.forward.
TEST INFRASTRUCTURE (see package docstring)."""
import random

import torch
import torch.nn.functional as F

from . import geometry as G
from . import unet as U
from .cfg import UNetCfg


def warp_attn(sd, p, pers_x, equi_x, cameras, opposite=None, masks=None):
    """WarpAttn.forward (src/modules/attn_perspano.py:22-99) with
    BasicTransformerBlock._forward / CrossAttention.forward (src/modules/transformer.py:151-167, 59-74).

    pers_x [(b m), c, f, ph, pw], equi_x [b, c, f, eh, ew].  ``opposite`` None draws the
    reference's ``random.random() < 0.4`` coin (src/utils/utils.py:15-16)."""
    bm, c, f, ph, pw = pers_x.shape
    b, _, _, eh, ew = equi_x.shape
    m = bm // b
    heads = c // 32
    if opposite is None:
        opposite = random.random() < 0.4
    if masks is None:
        masks = G.merged_masks(ph, pw, eh, ew, cameras, opposite)
    pers_masks, equi_masks = masks
    pers_c, equi_c = G.coords(ph, pw, eh, ew, cameras)
    pers_pe = G.spherical_pe(pers_c, c // 4)                      # [m, ph, pw, c]
    equi_pe = G.spherical_pe(equi_c, c // 4)                      # [eh, ew, c]

    # tokens: equi '(b f) (h w) c', pers '(b f) (m h w) c'
    eq = equi_x.permute(0, 2, 3, 4, 1).reshape(b * f, eh * ew, c)
    pr = pers_x.reshape(b, m, c, f, ph, pw).permute(0, 3, 1, 4, 5, 2).reshape(b * f, m * ph * pw, c)
    eq_pe = equi_pe.reshape(1, eh * ew, c)
    pr_pe = pers_pe.reshape(1, m * ph * pw, c)
    # bias = mask[0] broadcast over (b f heads) (transformer.py:68-70)
    bias_e2p = pers_masks.permute(1, 2, 0, 3, 4).reshape(eh * ew, m * ph * pw)   # rows: equi query
    bias_p2e = equi_masks.reshape(m * ph * pw, eh * ew)                          # rows: pers query
    t = p + "transformer."

    def block(x, x_pe, ctx_wpe, bias):
        q_in = U.layer_norm(sd, t + "norm1.", x + x_pe)
        ctx = U.layer_norm(sd, t + "norm1.", ctx_wpe)
        a = U.sdpa(U.linear(sd, t + "attn1.to_q.", q_in), U.linear(sd, t + "attn1.to_k.", ctx),
                   U.linear(sd, t + "attn1.to_v.", ctx), heads, bias=bias)
        x = U._st(U.linear(sd, t + "attn1.to_out.", a) + x)        # (U._st: identity unless a test emulates 16-bit storage)
        h = U.linear(sd, t + "ff.net.0.proj.", U.layer_norm(sd, t + "norm2.", x))
        a_, gate = h.chunk(2, dim=-1)
        return U._st(U.linear(sd, t + "ff.net.2.", a_ * F.gelu(gate)) + x)

    eq_out = block(eq, eq_pe, pr + pr_pe, bias_e2p)
    pr_out = block(pr, pr_pe, eq + eq_pe, bias_p2e)
    pers_out = pr_out.reshape(b, f, m, ph, pw, c).permute(0, 2, 5, 1, 3, 4).reshape(bm, c, f, ph, pw)
    equi_out = eq_out.reshape(b, f, eh, ew, c).permute(0, 4, 1, 2, 3)
    return pers_out, equi_out


def _branch_prep(sd, p, cfg: UNetCfg, timestep, fps, use_fps):
    c0 = cfg.block_out_channels[0]
    emb = U.timestep_mlp(sd, p + "time_embedding.", U.timestep_sincos(timestep, c0))
    if use_fps:
        emb = emb + U.timestep_mlp(sd, p + "fps_embedding.", U.timestep_sincos(fps, c0))
    return emb


def ip_tokens_clean(sd, p, cfg: UNetCfg, feats):
    """temporal_proj + image_proj_model, before the per-step noise (MVGenModel.py:158-184)."""
    t = U.temporal_projection(sd, p + "temporal_proj.", feats)
    t = t.reshape(t.shape[0], -1, t.shape[-1])
    return U.resampler(sd, p + "image_proj_model.", t)


def relpos_tokens(sd, p, cfg: UNetCfg, rel_pos, pitchs, n_tokens):
    """Per-frame relative-position (6 numbers) + pitch embeddings appended to the pano IP
    tokens (src/models/MVGenModel.py:189-222).  rel_pos [B, F, 6], pitchs [B, F]."""
    c0 = cfg.block_out_channels[0]
    B, Fr = rel_pos.shape[:2]
    out = []
    for i in range(Fr):
        e1 = U.timestep_sincos(rel_pos[:, i, :].flatten(), c0).reshape(B, -1)
        e1 = U.timestep_mlp(sd, p + "add_cond_embedding.", e1)
        e1 = F.linear(e1, sd[p + "cond_rp_proj.weight"])
        e2 = U.timestep_sincos(pitchs[:, i].flatten(), c0).reshape(B, -1)
        e2 = U.timestep_mlp(sd, p + "add_cond_embedding2.", e2)
        out.append(torch.cat([e1, e2], dim=-1))
    for _ in range(n_tokens - Fr):
        out.append(out[-1])
    return torch.stack(out, dim=1)


def mv_forward(sd, cfg: UNetCfg, latents, pano_latent, timestep, prompt_embd, pano_prompt_embd, cameras,
               fps_pano, fps_pers, feat_pano, feat_pers, rel_pos, pitchs, use_fps=True, taps=None,
               mask_cache=None):
    """MultiViewBaseModel.forward (src/models/MVGenModel.py:59-481), pano_pad=True,
    use_ip_plus_cross_attention=True, ip_plus_condition='video', use_relative_postions='WithAdapter'.

    latents [b,m,9,f,h,w], pano_latent [b,9,f,H,W], timestep int64[1].  RNG draws, in the
    reference's order: randn_like(pano ip tokens), randn_like(pers ip tokens)
    (MVGenModel.py:186-187), then one random.random() per WarpAttn (utils.py:15)."""
    b, m, c, f, h, w = latents.shape
    x = latents.reshape(b * m, c, f, h, w)
    # cameras arrive as [1, m, ...] even under CFG (pipeline...dual.py:614, MVGenModel.py:100-101)
    cams = {k: v.reshape(-1, *v.shape[2:])[:m] if torch.is_tensor(v) else v for k, v in cameras.items()}
    ts = timestep[:, None].repeat(b, m)
    pano_ts = ts[:, 0].clone()
    ts = ts.reshape(-1)
    U_, P_ = "unet.", "pano_unet."
    emb = _branch_prep(sd, U_, cfg, ts, fps_pers.reshape(-1).float(), use_fps)
    pemb = _branch_prep(sd, P_, cfg, pano_ts, fps_pano.float().expand(b), use_fps)

    x = U.conv2d_frames(sd, U_ + "conv_in.", x)
    px = G.unpad_pano(U.conv2d_frames(sd, P_ + "conv_in.", G.pad_pano(pano_latent, 1)), 1)

    ip_pano = ip_tokens_clean(sd, P_, cfg, feat_pano)
    if feat_pers.stride(1) == 0:      # views share one feature tensor (pipeline...dual.py:713): compute once
        ip_pers = ip_tokens_clean(sd, U_, cfg, feat_pers[:, 0]).repeat_interleave(m, dim=0)
    else:
        ip_pers = ip_tokens_clean(sd, U_, cfg, feat_pers.reshape(b * m, *feat_pers.shape[2:]))
    ip_pano = ip_pano + torch.randn_like(ip_pano) * 0.1
    ip_pers = ip_pers + torch.randn_like(ip_pers) * 0.1
    ip_pano = ip_pano + relpos_tokens(sd, P_, cfg, rel_pos, pitchs, ip_pano.shape[1])
    pctx = torch.cat([pano_prompt_embd, ip_pano], dim=1)
    ctx = torch.cat([prompt_embd, ip_pers], dim=1)

    heads = cfg.attention_head_dim
    nt, mh, g, eps = cfg.num_tokens, cfg.motion_heads, cfg.norm_num_groups, cfg.norm_eps
    xf = cfg.xformers

    def pano_res(p, t, temb):
        return G.unpad_pano(U.resnet_block(sd, p, G.pad_pano(t, 2), temb, g, eps), 2)

    def warp(p, xx, pp):
        opp = random.random() < 0.4
        key = (xx.shape[-2], xx.shape[-1], pp.shape[-2], pp.shape[-1], opp)
        mk = None
        if mask_cache is not None:
            if key not in mask_cache:
                mask_cache[key] = G.merged_masks(key[0], key[1], key[2], key[3], cams, opp)
            mk = mask_cache[key]
        return warp_attn(sd, p, xx, pp, cams, opposite=opp, masks=mk)

    def tap(name, a, b_):
        if taps is not None:
            taps[name] = (a.clone(), b_.clone())

    skips, pskips = [x], [px]
    nlev = len(cfg.block_out_channels)
    for i in range(nlev):
        d = f"down_blocks.{i}."
        for j in range(cfg.layers_per_block):
            x = U.resnet_block(sd, U_ + d + f"resnets.{j}.", x, emb, g, eps)
            px = pano_res(P_ + d + f"resnets.{j}.", px, pemb)
            if i < nlev - 1:                       # CrossAttnDownBlock3D; DownBlock3D motion modules are skipped
                x = U.spatial_transformer(sd, U_ + d + f"attentions.{j}.", x, ctx, heads[i], nt, g, xf)
                x = U.motion_module(sd, U_ + d + f"motion_modules.{j}.", x, mh, g)
                px = U.spatial_transformer(sd, P_ + d + f"attentions.{j}.", px, pctx, heads[i], nt, g, xf)
                px = U.motion_module(sd, P_ + d + f"motion_modules.{j}.", px, mh, g)
            skips.append(x)
            pskips.append(px)
        if i < nlev - 1:
            x = U.downsample(sd, U_ + d + "downsamplers.0.", x)
            px = G.unpad_pano(U.downsample(sd, P_ + d + "downsamplers.0.", G.pad_pano(px, 2)), 1)
            skips.append(x)
            pskips.append(px)
            x, px = warp(f"cp_blocks_encoder.{i}.", x, px)
            tap(f"enc{i}", x, px)

    x = U.resnet_block(sd, U_ + "mid_block.resnets.0.", x, emb, g, eps)
    px = pano_res(P_ + "mid_block.resnets.0.", px, pemb)
    x = U.spatial_transformer(sd, U_ + "mid_block.attentions.0.", x, ctx, heads[-1], nt, g, xf)
    x = U.motion_module(sd, U_ + "mid_block.motion_modules.0.", x, mh, g)
    x = U.resnet_block(sd, U_ + "mid_block.resnets.1.", x, emb, g, eps)
    px = U.spatial_transformer(sd, P_ + "mid_block.attentions.0.", px, pctx, heads[-1], nt, g, xf)
    px = U.motion_module(sd, P_ + "mid_block.motion_modules.0.", px, mh, g)
    px = pano_res(P_ + "mid_block.resnets.1.", px, pemb)
    x, px = warp("cp_blocks_mid.", x, px)
    tap("mid", x, px)

    rheads = list(reversed(heads))
    for i in range(nlev):
        u = f"up_blocks.{i}."
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            px = torch.cat([px, pskips.pop()], dim=1)
            x = U.resnet_block(sd, U_ + u + f"resnets.{j}.", x, emb, g, eps)
            px = pano_res(P_ + u + f"resnets.{j}.", px, pemb)
            if i > 0:                              # CrossAttnUpBlock3D; UpBlock3D motion modules are skipped
                x = U.spatial_transformer(sd, U_ + u + f"attentions.{j}.", x, ctx, rheads[i], nt, g, xf)
                x = U.motion_module(sd, U_ + u + f"motion_modules.{j}.", x, mh, g)
                px = U.spatial_transformer(sd, P_ + u + f"attentions.{j}.", px, pctx, rheads[i], nt, g, xf)
                px = U.motion_module(sd, P_ + u + f"motion_modules.{j}.", px, mh, g)
        if i < nlev - 1:
            x, px = warp(f"cp_blocks_decoder.{i}.", x, px)
            tap(f"dec{i}", x, px)
            x = U.upsample(sd, U_ + u + "upsamplers.0.", x)
            px = G.unpad_pano(U.upsample(sd, P_ + u + "upsamplers.0.", G.pad_pano(px, 1)), 2)

    x = F.silu(U.group_norm_frames(sd, U_ + "conv_norm_out.", x, g, eps))
    x = U.conv2d_frames(sd, U_ + "conv_out.", x)
    px = F.silu(U.group_norm_frames(sd, P_ + "conv_norm_out.", px, g, eps))
    px = G.unpad_pano(U.conv2d_frames(sd, P_ + "conv_out.", G.pad_pano(px, 1)), 1)
    return x.reshape(b, m, *x.shape[1:]), px
