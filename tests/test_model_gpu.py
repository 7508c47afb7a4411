"""Model-level parity on the MI355X: the shipped product path (HIP kernels through the C ABI, 16-bit storage)
against the CPU oracle / reference fixtures, plus size-independent properties at BASELINE cfg2 sizes.

Tolerances (relative L2 of the whole tensor, fp32 oracle as truth): one dual-branch forward through ~60
layers <= 3e-2 in bf16 / 4e-3 in fp16 and <= 1.5x what 16-bit storage alone costs (calibrated in the test by rounding the
oracle's own intermediates); VAE <= 3e-2; two-step pipeline latents+video <= 1e-1 bf16 / 3e-2 fp16 (CFG 7.5 amplifies
the 16-bit error)."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import gold, record as _record, rel  # noqa: E402
from im360_oracle import mv as OMV, vae as OV  # noqa: E402
from im360_oracle.cfg import sd21_unet_cfg, sd21_vae_cfg  # noqa: E402
from imagine360_amd import configs, kernels as K, synthetic as S  # noqa: E402
from imagine360_amd.scheduler import DDIMScheduler  # noqa: E402

torch.set_grad_enabled(False)


def _q(v, dt):
    return v.to(dt).float() if torch.is_floating_point(v) else v


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 3e-2), (torch.float16, 4e-3)])
def test_mv_forward_vs_oracle(dt, tol):
    """One dual-branch forward (~60 layers, 7 WarpAttn) at channels / 5 against the fp32 oracle, the real reference's
    fixture and the taps after every WarpAttn.  The bound is CALIBRATED: the oracle with every primitive's output and the
    residual stream rounded to the 16-bit dtype (im360_oracle.unet.storage: fp32 arithmetic, no kernel involved) measures
    what storage alone costs on this network (1.7e-2 in bf16, 2.1e-3 in fp16); the product must stay within 1.25x of it
    (measured: 1.01 - 1.02x -- the kernels add a few per cent to the rounding of the stored tensors)."""
    from im360_oracle import unet as OU
    dev = torch.device("cuda", 0)
    mv = configs.build_mv_model(5, device=dev, dtype=dt, xformers=True)
    mv.noise_on_host = True
    inp = S.mv_inputs(frames=8, pano_hw=(32, 64), pers_hw=(16, 16), seed=0, sam_frames=16)
    cams = S.icosahedron_cameras(90, 128)
    dinp = S.cast_mv_inputs(inp, dev, dt)          # (pitch / fps / crop rectangle stay float32, as in the pipeline)
    torch.manual_seed(7)
    random.seed(7)
    mv.taps = {}
    pers, pano = mv(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True, **dinp)
    cfg = sd21_unet_cfg(5)
    cfg.xformers = True
    sd = {k: v.float().cpu() for k, v in mv.state_dict().items()}
    torch.manual_seed(7)
    random.seed(7)
    otaps = {}
    masks = {}
    oargs = (sd, cfg, _q(inp["latents"], dt), _q(inp["pano_latent"], dt), inp["timestep"],
             _q(inp["prompt_embd"], dt), _q(inp["pano_prompt_embd"], dt), cams, inp["fps_tensor_pano"],
             inp["fps_tensor_pers"], _q(inp["reference_images_clip_feat_pano"], dt),
             _q(inp["reference_images_clip_feat_pers"], dt), inp["relative_position_tensor"], inp["pitchs_tensor"])
    o_pers, o_pano = OMV.mv_forward(*oargs, taps=otaps, mask_cache=masks)
    torch.manual_seed(7)
    random.seed(7)
    with OU.storage(dt):
        c_pers, c_pano = OMV.mv_forward(*oargs, mask_cache=masks)
    from imagine360_amd.layers import from_cl
    g = gold("mv_forward_w5_xf.npz")             # the reference fixture itself (real reference, xformers semantics)
    errs = dict(pano=rel(pano, o_pano), pers=rel(pers, o_pers), pano_vs_reference_fixture=rel(pano, g["pano"]))
    errs.update({f"tap_{n}": rel(from_cl(te, 8), otaps[n][1]) for n, (tp, te) in mv.taps.items()})   # after each WarpAttn
    cal = dict(storage_only_pano=rel(c_pano, o_pano), storage_only_pers=rel(c_pers, o_pers))
    _record(f"mv_forward_w5_{str(dt).split('.')[-1]}", **errs, **cal)
    assert max(errs.values()) < tol, errs
    assert errs["pano"] <= 1.25 * cal["storage_only_pano"] and errs["pers"] <= 1.25 * cal["storage_only_pers"], (errs, cal)   # measured 1.01 - 1.02x


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 3e-2), (torch.float16, 5e-3)])
def test_vae_vs_oracle(dt, tol):
    dev = torch.device("cuda", 0)
    vae = configs.build_vae(4, device=dev, dtype=dt)
    sd = {k: v.float().cpu() for k, v in vae.state_dict().items()}
    cfg = sd21_vae_cfg(4)
    gen = torch.Generator().manual_seed(9)
    x = torch.rand(2, 3, 64, 96, generator=gen) * 2 - 1
    z = torch.randn(2, 4, 8, 20, generator=gen)
    errs = dict(encode=rel(vae.encode(x.to(dev, dt)).latent_dist.parameters, OV.encode_moments(sd, cfg, _q(x, dt))),
                decode=rel(vae.decode(z.to(dev, dt)).sample, OV.decode(sd, cfg, _q(z, dt))))
    _record(f"vae_w4_{str(dt).split('.')[-1]}", **errs)
    assert max(errs.values()) < tol, errs


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 1e-1), (torch.float16, 3e-2)])
def test_pipeline_vs_reference_fixture(dt, tol):
    """Two DDIM steps + VAE decode on the GPU with host RNG, against the latents / video the REAL reference produced
    (fp32, CPU semantics incl. the logit-scale-1.0 cross-attention quirk) for the same seeds.  CFG 7.5 amplifies the
    16-bit error of the two predictions, hence the looser bound than for a single forward."""
    from imagine360_amd.pipeline import AnimationPipeline
    dev = torch.device("cuda", 0)
    mv = configs.build_mv_model(5, device=dev, dtype=dt, xformers=False)
    vae = configs.build_vae(4, device=dev, dtype=dt)
    pipe = AnimationPipeline(vae, None, None, mv.unet, mv.pano_unet, mv, DDIMScheduler(**configs.NOISE_SCHEDULER_KWARGS), None, "SAM").to(dev)
    pipe.rng, pipe._no_progress = "host", True
    vb = S.video_batch(frames=16, pano_hw=(256, 512), seed=0)
    cond = S.conditioning(frames=16, seed=0)
    g = gold("pipeline_w5.npz")
    trace = []
    torch.manual_seed(21)
    random.seed(21)
    vid = pipe("synthetic", num_inference_steps=2, guidance_scale_text=7.5, negative_prompt="", latents_dtype=dt, video_batch=vb,
               use_outpaint=True, use_ip_plus_cross_attention=True, use_fps_condition=True, ip_plus_condition="video",
               prompt_embeds=(cond["text_pano"], cond["text_pers"]), sam_features=(cond["sam_pano"], cond["sam_pers"]),
               trace=trace).videos
    assert vid.shape == (1, 3, 16, 256, 512) and vid.dtype == torch.float32 and torch.isfinite(vid).all()
    errs = {f"latent_step_{i}": rel(t, g[f"pano_latent_{i}"]) for i, t in enumerate(trace)}
    errs["video"] = rel(vid[:, :, ::3, ::4, ::4], g["video_sub"])
    errs["video_max_abs"] = float((vid[:, :, ::3, ::4, ::4] - g["video_sub"]).abs().max())
    _record(f"pipeline_2_steps_w5_{str(dt).split('.')[-1]}", **errs)
    assert max(v for k, v in errs.items() if k != "video_max_abs") < tol, errs


# ------------------------------------------------------------------ properties at BASELINE cfg2 sizes
def test_full_size_conv_is_equivariant_to_longitude_rotation():
    """Pano L0 resnet conv (32 images 64x128x320): circular addressing => rolling the input along W rolls the output,
    bit for bit (same per-pixel accumulation order)."""
    dt, dev = torch.bfloat16, "cuda"
    g = torch.Generator().manual_seed(1)
    x = torch.randn(32, 64, 128, 320, generator=g).to(dev, dt)
    w = (torch.randn(320, 320, 3, 3, generator=g) * (9 * 320) ** -0.5).to(dev, dt)
    wp = K.pack_conv_weight(w)
    y = K.conv2d(x, wp, 320, wrap=True)
    y2 = K.conv2d(torch.roll(x, 37, dims=2).contiguous(), wp, 320, wrap=True)
    assert torch.equal(torch.roll(y, 37, dims=2), y2)
    # linearity in the input (fp32 accumulate, one rounding): conv(2x) == 2 conv(x) exactly in bf16
    assert torch.equal(K.conv2d((x * 2).contiguous(), wp, 320, wrap=True), y * 2)


def test_full_size_attention_properties():
    """Pano L0 self-attention (32 frames x 5 heads, 8192 tokens, d 64): invariant to a permutation of the keys,
    independent across batch entries, rows of softmax sum to one (V = 1 -> O = 1)."""
    dt, dev = torch.bfloat16, "cuda"
    g = torch.Generator().manual_seed(2)
    B, N, H, D = 4, 8192, 5, 64
    q, k, v = (torch.randn(B, N, H * D, generator=g).to(dev, dt) for _ in range(3))
    o = K.attention(q, k, v, H)
    perm = torch.randperm(N, generator=g).to(dev)
    o2 = K.attention(q, k[:, perm].contiguous(), v[:, perm].contiguous(), H)
    assert rel(o2, o) < 4e-3
    o3 = K.attention(q[1:2].contiguous(), k[1:2].contiguous(), v[1:2].contiguous(), H)
    assert torch.equal(o3[0], o[1])
    ones = torch.ones_like(v)
    assert (K.attention(q, k, ones, H).float() - 1).abs().max() < 1e-2


def test_full_size_groupnorm_and_temporal_properties():
    dt, dev = torch.bfloat16, "cuda"
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(32, 64, 128, 320, generator=g) * 1.7 + 0.3).to(dev, dt)
    gamma, beta = torch.ones(320, device=dev, dtype=dt), torch.zeros(320, device=dev, dtype=dt)
    y = K.group_norm(x, gamma, beta, 32, 1e-5).float().reshape(32, 64 * 128, 32, 10)
    assert y.mean(dim=(1, 3)).abs().max() < 2e-2 and (y.var(dim=(1, 3), unbiased=False) - 1).abs().max() < 2e-2
    # pad-aware statistics == statistics of the explicitly padded tensor
    xp = K.circular_pad_w(x, 2)
    s1, h1 = K.group_norm_stats(x, gamma, beta, 32, 1e-5, pad=2)
    s2, h2 = K.group_norm_stats(xp, gamma, beta, 32, 1e-5, pad=0)
    assert rel(s1, s2) < 1e-5 and (h1 - h2).abs().max() < 1e-4
    # a single frame attends only to itself: temporal attention returns V
    qkv = torch.randn(2 * 1 * 8192, 3 * 320, generator=g).to(dev, dt)
    assert torch.equal(K.temporal_attention(qkv, 2, 1, 8192, 8), qkv[:, 640:])


def test_graph_replayed_step_equals_eager_step():
    """A captured hipGraph step replayed 3 times vs 3 eagerly issued steps (IP noise switched off so both paths see the
    same numbers; the WarpAttn coins come from Python's RNG in both).  Same kernels, same launch order: the replayed
    step reproduces the eager one to the last bit (tolerance left at 1e-5 for GEMM solution changes under capture)."""
    from imagine360_amd.graph_step import GraphedDenoiseStep
    dt, dev = torch.bfloat16, torch.device("cuda", 0)
    mv = configs.build_mv_model(5, device=dev, dtype=dt, xformers=True)
    mv._ip_noise = lambda like: torch.zeros_like(like)
    sch = DDIMScheduler(**configs.NOISE_SCHEDULER_KWARGS)
    sch.set_timesteps(25)
    ts = sch._timesteps_host
    cams = S.icosahedron_cameras(90, 128, device=dev)

    def fresh():
        inp = S.mv_inputs(frames=8, pano_hw=(32, 64), pers_hw=(16, 16), seed=4, sam_frames=16, dtype=dt, device=dev)
        return inp, inp["pano_latent"][:1, :4].clone(), inp["latents"][:1, :, :4].clone()

    inp, pano, pers = fresh()
    random.seed(5)
    first_pred = None
    for i in range(3):
        inp["pano_latent"][:, :4] = pano
        inp["latents"][:, :, :4] = pers
        pp, pn = mv(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True,
                    **{**inp, "timestep": torch.tensor([ts[i]], device=dev)})
        first_pred = pn.clone() if first_pred is None else first_pred
        pano = sch.fused_cfg_step(pn[0:1], pn[1:2], 7.5, ts[i], pano)
        pers = sch.fused_cfg_step(pp[0:1], pp[1:2], 7.5, ts[i], pers)
    inp2, pano2, pers2 = fresh()
    inp2.pop("timestep")
    g = GraphedDenoiseStep(mv, sch, inp2, cams, pano2, pers2, 7.5)      # consumes 7 Python draws while warming up
    random.seed(5)
    for i in range(3):
        g.step(ts[i])
        if i == 0:
            assert rel(g.pred_pano, first_pred) < 1e-5
    assert rel(g.pano_lat, pano) < 1e-5 and rel(g.pers_lat, pers) < 1e-5
    assert torch.isfinite(g.pano_lat.float()).all()


def test_dual_stream_forward_is_bit_identical_eager_and_graphed():
    """``MultiViewBaseModel.dual_stream``: the panorama branch's segments between WarpAttn calls issued on a side stream
    (forked from / joined into the current one).  Same kernels on the same data: eager predictions equal the single-stream
    ones to the last bit, three times in a row (allocator reuse across the two streams), and a hipGraph captured with the
    two parallel branches replays to the same latents as the single-stream graph."""
    from imagine360_amd.graph_step import GraphedDenoiseStep
    dt, dev = torch.bfloat16, torch.device("cuda", 0)
    mv = configs.build_mv_model(5, device=dev, dtype=dt, xformers=True)
    mv._ip_noise = lambda like: torch.zeros_like(like)
    cams = S.icosahedron_cameras(90, 128, device=dev)
    inp = S.mv_inputs(frames=8, pano_hw=(32, 64), pers_hw=(16, 16), seed=6, sam_frames=16, dtype=dt, device=dev)
    outs = {}
    mv.dual_stream_eager = True          # (eager two-stream issue is opt-in: see MultiViewBaseModel._two_streams)
    for dual in (False, True, False, True):
        mv.dual_stream = dual
        random.seed(9)
        pp, pn = mv(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True, **inp)
        torch.cuda.synchronize()
        if dual in outs:
            assert torch.equal(outs[dual][0], pp) and torch.equal(outs[dual][1], pn)
        outs[dual] = (pp.clone(), pn.clone())
    assert torch.equal(outs[False][0], outs[True][0]) and torch.equal(outs[False][1], outs[True][1])
    sch = DDIMScheduler(**configs.NOISE_SCHEDULER_KWARGS)
    sch.set_timesteps(25)
    ts = sch._timesteps_host
    lat = {}
    for dual in (False, True):
        mv.dual_stream = dual
        inp2 = S.mv_inputs(frames=8, pano_hw=(32, 64), pers_hw=(16, 16), seed=6, sam_frames=16, dtype=dt, device=dev)
        pano2, pers2 = inp2["pano_latent"][:1, :4].clone(), inp2["latents"][:1, :, :4].clone()
        inp2.pop("timestep")
        g = GraphedDenoiseStep(mv, sch, inp2, cams, pano2, pers2, 7.5)
        random.seed(5)
        for i in range(3):
            g.step(ts[i])
        torch.cuda.synchronize()
        lat[dual] = (g.pano_lat.clone(), g.pers_lat.clone())
        del g
    mv.dual_stream = False
    assert torch.equal(lat[False][0], lat[True][0]) and torch.equal(lat[False][1], lat[True][1])
    assert torch.isfinite(lat[True][0].float()).all()


# ------------------------------------------------------------------ full-width blocks, cfg4 / cfg5 sized kernels
_BLOCK_ORACLE = {}


@pytest.mark.parametrize("routed", [False, True])
@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 2e-2), (torch.float16, 5e-3)])
@pytest.mark.parametrize("level,c,heads,hw", [(0, 320, 5, (16, 16)), (1, 640, 10, (8, 16)), (2, 1280, 20, (8, 8))])
def test_full_width_block_vs_oracle(dt, tol, level, c, heads, hw, routed):
    """One FULL-WIDTH ResnetBlock3D -> Transformer3DModel (self + text/IP cross attention + GEGLU) -> motion module of
    UNet level 0 / 1 / 2 (320 / 640 / 1280 channels: temporal head dims 40 / 80 / 160, 5 / 10 / 20 spatial heads)
    against the fp32 oracle on the same filler weights; the panorama variant of the ResnetBlock (pad-aware statistics,
    x_off conv) included.  The reduced-width model tests never reach these head dims / tile shapes.

    ``routed``: the token-major Linears take the persistent MFMA kernel regardless of the token count (the production
    rule needs >= 65 536 tokens), i.e. the row statistics from the producers' epilogues, every LayerNorm folded into its
    consuming GEMM and the fused GEGLU run here at full width.

    Calibration: the same blocks through the oracle with every primitive's output ROUNDED to the 16-bit dtype (fp32
    arithmetic, no kernel involved) give the error that storage alone costs; the product must stay within 1.25x of it."""
    from imagine360_amd import layers
    from imagine360_amd.layers import from_cl, to_cl
    from imagine360_amd.mv_model import MultiViewBaseModel
    from imagine360_amd.unet3d import ResnetBlock3D, Transformer3DModel, VanillaTemporalModule
    from imagine360_amd.weights import filler_tensor
    from im360_oracle import unet as OU
    dev = torch.device("cuda", 0)
    pre = f"pano_unet.down_blocks.{level}."
    with torch.device("meta"):
        meta = MultiViewBaseModel(configs.build_unet(1), configs.build_unet(1)).state_dict()
    sd = {k: filler_tensor(k, v.shape) for k, v in meta.items()
          if k.startswith(pre) and torch.is_floating_point(v) and not k.endswith(".pe")}      # (pe is a fixed table)
    sd = {k: v.to(dt).float() for k, v in sd.items()}                      # the oracle sees the 16-bit weights
    cin = {0: 320, 1: 320, 2: 640}[level]
    res = ResnetBlock3D(in_channels=cin, out_channels=c, temb_channels=1280, groups=32, eps=1e-5)
    tr = Transformer3DModel(heads, 64, c, 1024, image_cross_attention_dim=1024, norm_num_groups=32, scale=1.0, num_tokens=64)
    tr.transformer_blocks[0].set_use_memory_efficient_attention_xformers(True)
    mm = VanillaTemporalModule(in_channels=c, **configs.PROMPT_DUAL_UNET_KWARGS["motion_module_kwargs"])
    for mod, sub in ((res, "resnets.0."), (tr, "attentions.0."), (mm, "motion_modules.0.")):
        missing, unexpected = mod.load_state_dict({k[len(pre + sub):]: v for k, v in sd.items() if k.startswith(pre + sub)}, strict=False)
        assert not unexpected and all(m.endswith("pe") for m in missing), (missing, unexpected)
        mod.to(dev, dt)
    g = torch.Generator().manual_seed(31 + level)
    b, f = 2, 16
    x = _q(torch.randn(b, cin, f, *hw, generator=g), dt)
    emb = _q(torch.randn(b, 1280, generator=g), dt)
    ctx = _q(torch.randn(b, 141, 1024, generator=g), dt)
    xc, _ = to_cl(x.to(dev, dt))
    errs, cal = {}, {}
    key = (level, dt)
    if key not in _BLOCK_ORACLE:            # the oracle (fp32 and storage-rounded) once per (level, dtype), shared by both routings
        from im360_oracle import geometry as OG
        o = OU.resnet_block(sd, pre + "resnets.0.", x, emb)
        o2 = OU.spatial_transformer(sd, pre + "attentions.0.", o, ctx, heads, 64, xformers=True)
        o3 = OU.motion_module(sd, pre + "motion_modules.0.", o2)
        op = OG.unpad_pano(OU.resnet_block(sd, pre + "resnets.0.", OG.pad_pano(x, 2), emb), 2)
        with OU.storage(dt):                     # what 16-bit storage alone costs on these blocks
            c1 = OU.resnet_block(sd, pre + "resnets.0.", x, emb)
            c2 = OU.spatial_transformer(sd, pre + "attentions.0.", c1, ctx, heads, 64, xformers=True)
            c3 = OU.motion_module(sd, pre + "motion_modules.0.", c2)
        _BLOCK_ORACLE[key] = (o, o2, o3, op, {"resnet": rel(c1, o), "transformer": rel(c2, o2), "motion": rel(c3, o3)})
    o, o2, o3, op, cal = _BLOCK_ORACLE[key]
    saved = layers.ROUTE_MIN_TOKENS
    if routed:
        layers.ROUTE_MIN_TOKENS = 0
    try:
        for pano in (False, True):
            y = res.forward_cl(xc, emb.to(dev, dt), f, pano)
            errs["resnet_pano" if pano else "resnet"] = rel(from_cl(y, f), op if pano else o)
            if pano:
                continue
            y2 = tr.forward_cl(y, ctx.to(dev, dt), f)
            errs["transformer"] = rel(from_cl(y2, f), o2)
            y3 = mm.forward_cl(y2, f)
            errs["motion"] = rel(from_cl(y3, f), o3)
    finally:
        layers.ROUTE_MIN_TOKENS = saved
    _record(f"full_width_block_L{level}_{str(dt).split('.')[-1]}" + ("_routed" if routed else ""), **errs,
            **{"storage_only_" + k: v for k, v in cal.items()})
    assert max(errs.values()) < tol, errs
    for k, v in cal.items():
        assert errs[k] <= 1.25 * v + 1e-4, (k, errs, cal)       # (round 2 measured 1.03 - 1.04x on every block)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_full_width_cfg1_step_and_vae_frame_vs_reference_fixture(dt):
    """ONE FULL-WIDTH dual-branch forward (both UNets at 320 / 640 / 1280 / 1280 channels, 7 WarpAttn, CFG batch 2) at the
    shapes of BASELINE cfg1 (8 frames, 256x512 equirect, 20 views) on the GPU against what the REAL reference computed in
    fp32 for the same bf16-rounded weights, inputs and seeds (tests/golden/mv_forward_full_cfg1.npz, written by
    oracle/tools/gen_golden.py mvfull, which also checks the oracle against the reference at full width: 2e-6) -- the
    whole-step comparison the reduced-width model tests cannot give -- in two routings: the production rule, and with the
    token-major GEMMs forced onto the MFMA kernel from 16 384 tokens so that the statistics-writing / LayerNorm-folded /
    skip-pair kernels run inside the full model at level 0 and 1 of both branches.
    The bound is calibrated: the fixture carries what 16-bit storage ALONE costs on this network (the oracle with every
    primitive's output rounded, no kernel involved: 8.9e-3 / 9.9e-3 in bf16); the product must stay within 1.25x of it
    (measured 1.02x).  Plus one full-width VAE decode of a 32x64 latent (a 256x512 frame) against the real AutoencoderKL."""
    from imagine360_amd import layers
    g = gold("mv_forward_full_cfg1.npz")
    dev = torch.device("cuda", 0)
    mv = configs.build_mv_model(1, device=dev, dtype=torch.bfloat16, xformers=True).to(dt)      # the fixture's weights: filler rounded to bf16
    mv.noise_on_host = True
    inp = S.mv_inputs(frames=8, pano_hw=(32, 64), pers_hw=(16, 16), seed=1, sam_frames=16)
    inp = {k: (v.to(torch.bfloat16) if torch.is_floating_point(v) and k not in S.FP32_INPUTS else v) for k, v in inp.items()}
    cams = S.icosahedron_cameras(90, 128)
    dinp = S.cast_mv_inputs(inp, dev, dt)
    o_pers, o_pano = g["pers"].float(), g["pano"].float()
    cal_pano, cal_pers = (float(v) for v in g["storage_only_bf16" if dt == torch.bfloat16 else "storage_only_fp16"])
    errs = {"storage_only_pano": cal_pano, "storage_only_pers": cal_pers}
    outs = {}
    for name, min_tokens in (("production_routing", layers.ROUTE_MIN_TOKENS), ("mfma_routing_from_16k_tokens", 16384)):
        saved = layers.ROUTE_MIN_TOKENS
        layers.ROUTE_MIN_TOKENS = min_tokens
        try:
            torch.manual_seed(7)
            random.seed(7)
            pers, pano = mv(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True, **dinp)
            outs[name] = (pers.float().cpu(), pano.float().cpu())
        finally:
            layers.ROUTE_MIN_TOKENS = saved
    for name, (pers, pano) in outs.items():
        errs[name + "_pano"], errs[name + "_pers"] = rel(pano, o_pano), rel(pers, o_pers)
        # no single view / frame may be off: the worst (view, frame) slice of the perspective prediction
        d = (pers - o_pers).flatten(3).norm(dim=(2, 3)) / o_pers.flatten(3).norm(dim=(2, 3))
        errs[name + "_worst_view"] = float(d.max())
    errs["routings_agree"] = rel(outs["mfma_routing_from_16k_tokens"][1], outs["production_routing"][1])
    del mv
    torch.cuda.empty_cache()
    vae = configs.build_vae(1, device=dev, dtype=torch.bfloat16).to(dt)
    z = torch.randn(1, 4, 32, 64, generator=torch.Generator().manual_seed(9)).to(torch.bfloat16)
    errs["vae_decode_full_width"] = rel(vae.decode(z.to(dev, dt)).sample, g["vae_decode"].float())
    _record(f"full_width_cfg1_step_{str(dt).split('.')[-1]}", **errs)
    for name in outs:
        assert errs[name + "_pano"] <= 1.25 * cal_pano + 2e-4, errs
        assert errs[name + "_pers"] <= 1.25 * cal_pers + 2e-4, errs
        assert errs[name + "_worst_view"] <= 2 * cal_pers + 2e-4, errs
    assert errs["vae_decode_full_width"] < (2e-2 if dt == torch.bfloat16 else 3e-3), errs


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_full_width_cfg2_step_vs_reference_fixture_in_every_launch_mode(dt):
    """The BENCHMARKED configuration itself (BASELINE cfg2: full width, 16 frames, 64x128 panorama latent + 20 views of 32x32,
    CFG batch 2 -- src/models/MVGenModel.py:59-481 at the shapes of configs/prompt-dual.yaml:59-72):
    (1) one dual-branch forward against what the REAL reference computed in fp32 for the same bf16-rounded weights, inputs and
        seeds (tests/golden/mv_forward_full_cfg2.npz, oracle/tools/gen_golden.py mvfull2: 17 minutes of host time there), within
        1.12x of what 16-bit storage alone costs on this network (the fixture's calibration; observed 1.04 - 1.06x);
    (2) the same forward issued eagerly with the panorama branch on the side stream (opt-in mode; recorded, not asserted);
    (3) one whole denoising step (forward + CFG + DDIM) replayed from the captured two-stream hipGraph -- the launch mode
        bench.py times -- against the step issued eagerly on one stream from the same latents and RNG states."""
    from imagine360_amd.graph_step import GraphedDenoiseStep
    g = gold("mv_forward_full_cfg2.npz")
    dev = torch.device("cuda", 0)
    mv = configs.build_mv_model(1, device=dev, dtype=torch.bfloat16, xformers=True).to(dt)      # the fixture's weights: filler rounded to bf16
    mv.noise_on_host = True
    inp = S.mv_inputs(frames=16, pano_hw=(64, 128), pers_hw=(32, 32), seed=1, sam_frames=16)
    inp = {k: (v.to(torch.bfloat16) if torch.is_floating_point(v) and k not in S.FP32_INPUTS else v) for k, v in inp.items()}
    cams = S.icosahedron_cameras(90, 256)
    dinp = S.cast_mv_inputs(inp, dev, dt)
    dcams = S.icosahedron_cameras(90, 256, device=dev)
    views = [int(v) for v in g["pers_view_index"]] if "pers_view_index" in g else list(range(20))
    o_pers, o_pano = g["pers"].float(), g["pano"].float()
    cal_pano, cal_pers = (float(v) for v in g["storage_only_bf16" if dt == torch.bfloat16 else "storage_only_fp16"])
    errs = {"storage_only_pano": cal_pano, "storage_only_pers": cal_pers}
    outs = {}
    mv.dual_stream_eager = True          # informational: the eager two-stream forward (opt-in; the default issues eager steps on one stream)
    for dual in (False, True):
        mv.dual_stream = dual
        torch.manual_seed(7)
        random.seed(7)
        pers, pano = mv(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True, **dinp)
        torch.cuda.synchronize()
        outs[dual] = (pers.clone(), pano.clone())
    pers, pano = outs[False][0].float().cpu()[:, views], outs[False][1].float().cpu()
    errs["pano"], errs["pers"] = rel(pano, o_pano), rel(pers, o_pers)
    d = (pers - o_pers).flatten(3).norm(dim=(2, 3)) / o_pers.flatten(3).norm(dim=(2, 3))
    errs["worst_view"] = float(d.max())
    dp = (pano - o_pano).permute(0, 2, 1, 3, 4).flatten(2).norm(dim=2) / o_pano.permute(0, 2, 1, 3, 4).flatten(2).norm(dim=2)
    errs["worst_pano_frame"] = float(dp.max())
    errs["eager_two_streams_bit_identical"] = bool(torch.equal(outs[False][0], outs[True][0]) and torch.equal(outs[False][1], outs[True][1]))
    mv.dual_stream_eager = False
    # (3) the timed launch mode: captured two-stream graph vs eager one-stream step, IP-adapter noise from the device generator
    mv.noise_on_host = False
    sch = DDIMScheduler(**configs.NOISE_SCHEDULER_KWARGS)
    sch.set_timesteps(25)
    ts = sch._timesteps_host
    ginp = {k: v.clone() for k, v in dinp.items() if k != "timestep"}
    p0, q0 = ginp["pano_latent"][:1, :4].clone(), ginp["latents"][:1, :, :4].clone()
    mv.dual_stream = True
    gs = GraphedDenoiseStep(mv, sch, ginp, dcams, p0, q0, 7.5, warmup=1)
    py_state, dev_state = random.getstate(), torch.cuda.get_rng_state(dev)
    gs.step(ts[3])
    torch.cuda.synchronize()
    g_pano, g_pers = gs.pano_lat.clone(), gs.pers_lat.clone()
    del gs
    mv.dual_stream = False
    random.setstate(py_state)
    torch.cuda.set_rng_state(dev_state, dev)
    einp = {k: v.clone() for k, v in dinp.items()}
    einp["pano_latent"][:, :4] = p0
    einp["latents"][:, :, :4] = q0
    einp["timestep"] = torch.tensor([ts[3]], dtype=torch.int64, device=dev)
    pp, pn = mv(cameras=dcams, use_fps_condition=True, use_ip_plus_cross_attention=True, **einp)
    e_pano = sch.fused_cfg_step(pn[0:1], pn[1:2], 7.5, ts[3], p0)
    e_pers = sch.fused_cfg_step(pp[0:1], pp[1:2], 7.5, ts[3], q0)
    errs["graph_two_streams_vs_eager_pano"], errs["graph_two_streams_vs_eager_pers"] = rel(g_pano, e_pano), rel(g_pers, e_pers)
    errs["graph_bit_identical"] = bool(torch.equal(g_pano, e_pano) and torch.equal(g_pers, e_pers))
    errs["step_moved_the_latent"] = rel(g_pano, p0)
    _record(f"full_width_cfg2_step_{str(dt).split('.')[-1]}", **errs)
    # (1.12x the storage-only error -- observed 1.04 - 1.06x -- so that a regression of one kernel family cannot hide)
    assert errs["pano"] <= 1.12 * cal_pano + 1e-4 and errs["pers"] <= 1.12 * cal_pers + 1e-4, errs
    assert errs["worst_view"] <= 2 * cal_pers + 2e-4 and errs["worst_pano_frame"] <= 2 * cal_pano + 2e-4, errs
    assert errs["graph_bit_identical"], errs                 # (bit identity, which is what every recorded run showed)
    assert errs["step_moved_the_latent"] > 1e-2, errs


# ------------------------------------------------------------------ whole steps at BASELINE cfg4 / cfg5 size
_BIG = {"cfg4": dict(frames=48, pano_hw=(64, 128), pers_hw=(32, 32), px=256, dt=torch.bfloat16),
        "cfg5": dict(frames=16, pano_hw=(128, 256), pers_hw=(64, 64), px=512, dt=torch.float16)}


@pytest.mark.parametrize("name", ["cfg4", "cfg5"])
def test_whole_denoising_step_at_cfg4_and_cfg5_size(name):
    """ONE WHOLE full-width denoising step (src/models/MVGenModel.py:59-481 + CFG + DDIM) at the sizes of BASELINE cfg4 (48 frames
    of 512 x 1024, bf16: the 17 - 64 frame temporal kernels, 3x the token counts of cfg2) and cfg5 (16 frames of 1024 x 2048, fp16:
    32 768-token panorama self-attention, 8192 x 20 480 WarpAttn masks, 4x the token counts) on one GPU.  No CPU reference fits
    these sizes (1288 TFLOP per step); what is asserted is size-independent:
    (1) the step replayed from the captured two-stream hipGraph == the step issued eagerly on one stream, BIT FOR BIT, from the
        same latents and RNG states (graph capture, routing thresholds, mask / PE tables and every kernel at these shapes);
    (2) CFG-row independence: with the conditioning of row 0 copied into row 1 the two rows of every prediction agree -- bit for
        bit at cfg4 size; at cfg5 size hipBLASLt's stream-K kernels accumulate the two rows' tiles in position-dependent orders
        and ~60 layers decorrelate those last-bit differences up to the storage error of the format (1.3e-3 / 1.7e-3 observed in
        fp16, like two runs with different GEMM solutions), hence a bound of a few storage errors instead of equality: a row
        read from the wrong place is an O(1) error;
    (3) longitude rotation: with the seven WarpAttn blocks at their reference initialisation (zero output projections,
        src/modules/transformer.py:30-32, 55-57: exact identities) the panorama branch is the circularly padded UNet alone, and
        rolling the input latent along W rolls the prediction -- up to the 16-bit rounding of two different evaluations and the
        reference's own padded GroupNorm statistics (pad_pano(2) weighs four columns twice, MVGenModel.py:276-281, so the
        reference is not exactly equivariant either).  A wrong wrap / seam at any of the 4 resolutions shows as a spike in the
        per-column error profile."""
    from imagine360_amd.graph_step import GraphedDenoiseStep
    w = _BIG[name]
    dt, dev = w["dt"], torch.device("cuda", 0)
    tag = f"whole_step_{name}_{str(dt).split('.')[-1]}"
    mv = configs.build_mv_model(1, device=dev, dtype=dt, xformers=True)
    inp = S.mv_inputs(frames=w["frames"], pano_hw=w["pano_hw"], pers_hw=w["pers_hw"], seed=1, sam_frames=max(16, w["frames"]), dtype=dt, device=dev)
    cams = S.icosahedron_cameras(90, w["px"], device=dev)
    sch = DDIMScheduler(**configs.NOISE_SCHEDULER_KWARGS)
    sch.set_timesteps(25)
    ts = sch._timesteps_host
    kw = dict(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True)
    errs = {}
    # ---- (1) captured two-stream graph == eager one-stream step
    ginp = {k: v.clone() for k, v in inp.items() if k != "timestep"}
    p0, q0 = ginp["pano_latent"][:1, :4].clone(), ginp["latents"][:1, :, :4].clone()
    mv.dual_stream = True
    gs = GraphedDenoiseStep(mv, sch, ginp, cams, p0, q0, 7.5, warmup=1)
    py_state, dev_state = random.getstate(), torch.cuda.get_rng_state(dev)
    gs.step(ts[3])
    torch.cuda.synchronize()
    g_pano, g_pers = gs.pano_lat.clone(), gs.pers_lat.clone()
    del gs, ginp
    torch.cuda.empty_cache()
    mv.dual_stream = False
    random.setstate(py_state)
    torch.cuda.set_rng_state(dev_state, dev)
    einp = {k: v.clone() for k, v in inp.items()}
    einp["pano_latent"][:, :4] = p0
    einp["latents"][:, :, :4] = q0
    einp["timestep"] = torch.tensor([ts[3]], dtype=torch.int64, device=dev)
    pp, pn = mv(**kw, **einp)
    e_pano = sch.fused_cfg_step(pn[0:1], pn[1:2], 7.5, ts[3], p0)
    e_pers = sch.fused_cfg_step(pp[0:1], pp[1:2], 7.5, ts[3], q0)
    errs["graph_two_streams_vs_eager_pano"], errs["graph_two_streams_vs_eager_pers"] = rel(g_pano, e_pano), rel(g_pers, e_pers)
    errs["graph_bit_identical"] = bool(torch.equal(g_pano, e_pano) and torch.equal(g_pers, e_pers))
    errs["finite"] = bool(torch.isfinite(g_pano.float()).all() and torch.isfinite(g_pers.float()).all())
    errs["step_moved_the_latent"] = rel(g_pano, p0)
    del pp, pn, g_pers, e_pers
    # ---- (2) CFG-row independence (IP-adapter noise off: it is drawn per row)
    mv._ip_noise = lambda like: torch.zeros_like(like)
    m = einp["latents"].shape[1]
    rinp = {}
    for k, v in einp.items():
        if k != "timestep" and v.dim() >= 1 and v.shape[0] in (2, 2 * m):
            half = v.shape[0] // 2
            # (the perspective SAM features are ONE tensor expanded over the views, stride 0: keep them a view)
            v = v[:1].expand(v.shape) if (half == 1 and 0 in v.stride()) else torch.cat([v[:half], v[:half]]).contiguous()
        rinp[k] = v
    pp, pn = mv(**kw, **rinp)
    errs["cfg_rows_pano"], errs["cfg_rows_pers"] = rel(pn[1], pn[0]), rel(pp[1], pp[0])
    del pp, pn, rinp
    # ---- (3) longitude rotation of the panorama branch with the WarpAttn blocks at their (identity) initialisation
    for blk in list(mv.cp_blocks_encoder) + [mv.cp_blocks_mid] + list(mv.cp_blocks_decoder):
        for prm in (blk.transformer.attn1.to_out.weight, blk.transformer.attn1.to_out.bias, blk.transformer.ff.net[2].weight, blk.transformer.ff.net[2].bias):
            prm.zero_()                    # (in place ON the parameter: bumps its version, the packed-weight caches rebuild)
    W = w["pano_hw"][1]
    shift = (W // 32 + 1) * 8              # a multiple of 8: the three stride-2 downsamplers sample on a grid of period 8 latent columns
    _, pn_a = mv(**kw, **einp)
    pn_a = pn_a.clone()
    rolled = dict(einp, pano_latent=torch.roll(einp["pano_latent"], shift, dims=-1).contiguous())
    _, pn_b = mv(**kw, **rolled)
    want = torch.roll(pn_a, shift, dims=-1)
    errs["rotation_rel"] = rel(pn_b, want)
    col = ((pn_b.float() - want.float()) ** 2).sum(dim=(0, 1, 2, 3)).sqrt() / (want.float() ** 2).sum(dim=(0, 1, 2, 3)).sqrt()
    errs["rotation_worst_column_over_median"] = float(col.max() / col.median())
    errs["rotation_moved_the_prediction"] = rel(pn_b, pn_a)
    _record(tag, **errs)
    half = dt == torch.float16
    assert errs["finite"] and errs["graph_bit_identical"], errs
    assert errs["step_moved_the_latent"] > 3e-3, errs
    assert errs["cfg_rows_pano"] < (4e-3 if half else 2.5e-2) and errs["cfg_rows_pers"] < (4e-3 if half else 2.5e-2), errs
    assert errs["rotation_rel"] < (4e-3 if half else 2.5e-2), errs             # observed 1.4e-3 (fp16, cfg5) / 9.1e-3 (bf16, cfg4): the storage error of two evaluations
    assert errs["rotation_worst_column_over_median"] < 2.5 and errs["rotation_moved_the_prediction"] > 0.1, errs      # (the roll is not a no-op: 0.41 / 0.51 observed)


def test_cfg5_sized_kernels_fp16():
    """BASELINE cfg5 (1024x2048 equirect, fp16) tensor sizes through the kernels: panorama level-0 self-attention with
    32 768 tokens (d 64), WarpAttn level-1 attention 8192 x 20 480 with the shared bias (d 32), a 128 x 256 x 320
    convolution with circular wrap -- each against the fp32 oracle on a slice of the output (the full score matrices do
    not fit a CPU reference)."""
    import torch.nn.functional as F
    from im360_oracle import unet as OU
    dt, dev = torch.float16, "cuda"
    g = torch.Generator().manual_seed(41)
    errs = {}
    # pano L0 self-attention: 2 frames x 5 heads, 32768 tokens
    B, H, D, N = 2, 5, 64, 32768
    q, k, v = (_q(torch.randn(B, N, H * D, generator=g), dt) for _ in range(3))
    o = K.attention(q.to(dev, dt), k.to(dev, dt), v.to(dev, dt), H)
    rows = torch.cat([torch.arange(0, 64), torch.arange(16000, 16064), torch.arange(N - 64, N)])
    errs["attn_32768"] = rel(o[:, rows], OU.sdpa(q[:, rows], k, v, H))
    # WarpAttn L1: equirect queries 64x128 = 8192, keys 20 views x 32x32 = 20480, shared bias in [-1, 1]
    B, H, D, Nq, Nk = 1, 10, 32, 8192, 20480
    q, k, v = _q(torch.randn(B, Nq, H * D, generator=g), dt), _q(torch.randn(B, Nk, H * D, generator=g), dt), _q(torch.randn(B, Nk, H * D, generator=g), dt)
    bias = _q(torch.rand(Nq, Nk, generator=g) * 2 - 1, dt)
    o = K.attention(q.to(dev, dt), k.to(dev, dt), v.to(dev, dt), H, bias=bias.to(dev, dt))
    rows = torch.cat([torch.arange(0, 96), torch.arange(Nq - 96, Nq)])
    errs["warp_8192x20480"] = rel(o[:, rows], OU.sdpa(q[:, rows], k, v, H, bias=bias[rows]))
    # pano L0 convolution 128 x 256 x 320 -> 320, circular W (4 images = 512 tiles of 256 pixels)
    x = _q(torch.randn(4, 128, 256, 320, generator=g), dt)
    w = _q(torch.randn(320, 320, 3, 3, generator=g) * (9 * 320) ** -0.5, dt)
    bsv = _q(torch.randn(320, generator=g) * 0.1, dt)
    y = K.conv2d(x.to(dev, dt), K.pack_conv_weight(w.to(dev, dt)), 320, bias=bsv.to(dev, dt), wrap=True)
    xr = x[:1].permute(0, 3, 1, 2)
    ref = F.conv2d(torch.cat([xr[..., -1:], xr, xr[..., :1]], dim=-1), w, bsv, padding=(1, 0)).permute(0, 2, 3, 1)
    errs["conv_128x256"] = rel(y[:1], ref)
    _record("cfg5_sized_kernels_fp16", **errs)
    assert max(errs.values()) < 3e-3, errs


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_temporal_attention_48_frames_8192_pixels(dt):
    """BASELINE cfg4: 48 frames (the long-sequence kernel path), level-0 panorama pixel count, d = 40."""
    from im360_oracle import unet as OU
    B, Fr, P, heads, d = 1, 48, 8192, 8, 40
    C = heads * d
    g = torch.Generator().manual_seed(43)
    qkv = _q(torch.randn(B * Fr * P, 3 * C, generator=g), dt)
    out = K.temporal_attention(qkv.to("cuda", dt), B, Fr, P, heads)
    sel = torch.cat([torch.arange(0, 64), torch.arange(5000, 5064), torch.arange(P - 64, P)])
    t = qkv.reshape(B, Fr, P, 3 * C)[:, :, sel].permute(0, 2, 1, 3).reshape(B * len(sel), Fr, 3 * C)
    ref = OU.sdpa(t[..., :C], t[..., C:2 * C], t[..., 2 * C:], heads).reshape(B, len(sel), Fr, C).permute(0, 2, 1, 3)
    err = rel(out.reshape(B, Fr, P, C)[:, :, sel], ref)
    _record(f"temporal_48f_8192px_{str(dt).split('.')[-1]}", rel=err)
    assert err < (1e-2 if dt == torch.bfloat16 else 3e-3)


def test_graph_and_eager_pipelines_agree_from_the_same_seeds():
    """The default pipeline (device RNG, one captured hipGraph per step, latents_dtype left at its float16 default on a
    bfloat16 model) and the eager pipeline see the same noise / coin streams for the same seeds: building the graph
    consumes no randomness."""
    from imagine360_amd.pipeline import AnimationPipeline
    dt, dev = torch.bfloat16, torch.device("cuda", 0)
    mv = configs.build_mv_model(5, device=dev, dtype=dt, xformers=True)
    vae = configs.build_vae(4, device=dev, dtype=dt)
    vb = S.video_batch(frames=16, pano_hw=(256, 512), seed=2)
    cond = S.conditioning(frames=16, seed=2)
    vids, lats = [], []
    for use_graph in (True, False):
        pipe = AnimationPipeline(vae, None, None, mv.unet, mv.pano_unet, mv, DDIMScheduler(**configs.NOISE_SCHEDULER_KWARGS), None, "SAM").to(dev)
        pipe._no_progress, pipe.use_graph = True, use_graph
        torch.manual_seed(33)
        random.seed(33)
        vids.append(pipe("synthetic", num_inference_steps=3, guidance_scale_text=7.5, negative_prompt="", video_batch=vb,
                         use_outpaint=True, use_ip_plus_cross_attention=True, use_fps_condition=True, ip_plus_condition="video",
                         prompt_embeds=(cond["text_pano"], cond["text_pers"]), sam_features=(cond["sam_pano"], cond["sam_pers"])).videos)
        lats.append(pipe.last_latents[0].float().cpu())
    err_l, err_v = rel(lats[0], lats[1]), rel(vids[0], vids[1])
    _record("graph_vs_eager_pipeline", latents=err_l, video=err_v)
    assert lats[0].dtype == torch.float32 and err_l < 1e-5 and err_v < 1e-5


@pytest.mark.parametrize("dt,early", [(torch.bfloat16, (1.2e-2, 2.0e-2, 6.5e-2)), (torch.float16, (1.6e-3, 2.8e-3, 1.0e-2))])
def test_pipeline_25_steps_vs_reference_fixture(dt, early):
    """The FULL 25-step DDIM loop (CFG 7.5) + VAE decode at channels / 10 against the latent trajectory and video the REAL
    reference produced on CPU in fp32 for the same seeds (tests/golden/pipeline25_w10.npz, oracle/tools/gen_golden.py
    pipeline25).

    With random-init weights the guided recurrence is EXPANSIVE: the fixture records that two fp32 evaluations of it
    (oracle vs reference, both CPU fp32) drift apart from 4e-7 at step 0 to 3.5e-3 at step 24, a x8500 amplification of
    rounding noise.  A 16-bit evaluation starts from ~6e-3 (bf16), so pointwise agreement at step 24 is not attainable by
    any 16-bit implementation; what IS checkable, and is asserted here:
      (1) the early steps, where the error is still the per-step kernel error, meet a stated tolerance;
      (2) at every recorded step the 16-bit error grows no faster than the recurrence itself amplifies fp32 rounding
          noise: err16(t) / drift32(t) <= 1.5 * err16(0) / drift32(0)  (a step that were wrong only late in the schedule
          -- other alphas, other timestep embeddings -- would break this);
      (3) the decoded video is finite, in range, and its per-frame mean / std match the reference's.
    Observed errors are written to gpurun_out/parity_observed.json."""
    import os
    from helpers import GOLDEN
    if not os.path.isfile(os.path.join(GOLDEN, "pipeline25_w10.npz")):
        pytest.skip("fixture not generated")
    from imagine360_amd.pipeline import AnimationPipeline
    dev = torch.device("cuda", 0)
    mv = configs.build_mv_model(10, device=dev, dtype=dt, xformers=False, motion_heads=4)     # temporal head dim 8 at 32 channels
    vae = configs.build_vae(4, device=dev, dtype=dt)
    pipe = AnimationPipeline(vae, None, None, mv.unet, mv.pano_unet, mv, DDIMScheduler(**configs.NOISE_SCHEDULER_KWARGS), None, "SAM").to(dev)
    pipe.rng, pipe._no_progress = "host", True
    vb = S.video_batch(frames=16, pano_hw=(256, 512), seed=0)
    cond = S.conditioning(frames=16, seed=0)
    g = gold("pipeline25_w10.npz")
    trace = []
    torch.manual_seed(21)
    random.seed(21)
    vid = pipe("synthetic", num_inference_steps=25, guidance_scale_text=7.5, negative_prompt="", latents_dtype=dt, video_batch=vb,
               use_outpaint=True, use_ip_plus_cross_attention=True, use_fps_condition=True, ip_plus_condition="video",
               prompt_embeds=(cond["text_pano"], cond["text_pers"]), sam_features=(cond["sam_pano"], cond["sam_pers"]),
               trace=trace).videos
    assert len(trace) == 25 and torch.isfinite(vid).all()
    assert float(vid.min()) >= 0.0 and float(vid.max()) <= 1.0
    steps = (0, 1, 4, 9, 14, 19, 24)
    drift = g["oracle_vs_reference_rel_l2_per_step"].double()
    lat = {i: rel(trace[i], g[f"pano_latent_{i}"]) for i in steps}
    errs = {f"latent_step_{i}": lat[i] for i in steps}
    errs.update({f"amplification_vs_fp32_step_{i}": lat[i] / float(drift[i]) for i in steps})
    errs["fp32_oracle_vs_reference_step_0"], errs["fp32_oracle_vs_reference_step_24"] = float(drift[0]), float(drift[24])
    errs["video_pointwise"] = rel(vid[:, :, ::3, ::4, ::4], g["video_sub"])
    v = vid.float().cpu()
    stats = torch.stack([v.mean(dim=(0, 1, 3, 4)), v.std(dim=(0, 1, 3, 4))])
    errs["video_frame_mean_max_abs"] = float((stats[0] - g["video_frame_stats"][0]).abs().max())
    errs["video_frame_std_max_abs"] = float((stats[1] - g["video_frame_stats"][1]).abs().max())
    _record(f"pipeline_25_steps_{str(dt).split('.')[-1]}", **errs)
    for i, tol in zip((0, 1, 4), early):                                                     # (1)
        assert lat[i] < tol, errs
    base = lat[0] / float(drift[0])
    for i in steps:                                                                           # (2)
        assert lat[i] / float(drift[i]) <= 1.5 * base, (i, errs)
    assert errs["video_frame_mean_max_abs"] < 1e-2 and errs["video_frame_std_max_abs"] < 1e-2, errs   # (3)


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 2e-2), (torch.float16, 4e-3)])
def test_ops_vs_reference_fixture(dt, tol):
    """Per-op outputs of the REAL reference (tests/golden/ops_w5.npz) against the product blocks on the GPU: the hoisted
    IP-adapter conditioning (TemporalProjection + Resampler), a panorama ResnetBlock3D, the spatial
    transformer (xformers semantics), the motion module, the circular down / up samplers and conv_in."""
    from helpers import op_inputs
    from imagine360_amd.layers import from_cl, to_cl
    dev = torch.device("cuda", 0)
    mv = configs.build_mv_model(5, device=dev, dtype=dt, xformers=True)
    g, I, un = gold("ops_w5.npz"), op_inputs(), mv.pano_unet
    d = lambda t: t.to(dev, dt)
    x, f = to_cl(d(I["x"]))
    errs = {"ip_tokens": rel(un.ip_tokens_clean(d(I["feat"])), g["ip_tokens"]),
            "resnet_pano": rel(from_cl(un.down_blocks[0].resnets[0].forward_cl(x, d(I["emb"]), f, pano=True), f), g["resnet_pano"]),
            "spatial_xf": rel(un.down_blocks[0].attentions[0](d(I["x"]), encoder_hidden_states=d(I["ctx"])).sample, g["spatial_xf"]),
            "motion": rel(un.down_blocks[0].motion_modules[0](d(I["x"]), d(I["emb"]), d(I["ctx"])), g["motion"]),
            "down_pano": rel(from_cl(un.down_blocks[0].downsamplers[0].forward_cl(x, pano=True), f), g["down_pano"])}
    x3, _ = to_cl(d(I["x3"]))
    errs["up_pano"] = rel(from_cl(un.up_blocks[0].upsamplers[0].forward_cl(x3, pano=True), f), g["up_pano"])
    l9, _ = to_cl(d(I["lat9"]))
    errs["conv_in_pano"] = rel(from_cl(un.conv_in_cl(l9, pano=True), f), g["conv_in_pano"])
    _record(f"ops_w5_{str(dt).split('.')[-1]}", **errs)
    assert max(errs.values()) < tol, errs


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 2.5e-2), (torch.float16, 3.5e-3)])
def test_single_branch_unet_forward_on_the_kernels(dt, tol):
    """SURVEY row a21 on the GPU: ``UNet3DConditionModel.forward`` (the kept single-branch API: reference layout in / out,
    IP-adapter conditioning and fps embedding inside) through the HIP kernels against the SAME host code on exact fp32
    stand-in kernels (tests/_emu_kernels.py; that pairing is pinned on the real reference's forward to 2e-6 by the CPU
    tier).  The bound is the 16-bit storage error of a ~60-layer forward."""
    import _emu_kernels as E
    dev = torch.device("cuda", 0)
    un = configs.build_mv_model(5, device=dev, dtype=dt, xformers=True).pano_unet
    ref = configs.build_mv_model(5, device="cpu", dtype=dt, xformers=True).pano_unet.float()       # same 16-bit weights, fp32 math
    g = torch.Generator().manual_seed(17)
    x = _q(torch.randn(2, 9, 8, 16, 32, generator=g), dt)
    ctx = _q(torch.randn(2, 77, 1024, generator=g), dt)
    feat = _q(torch.randn(2, 16, 4096, 256, generator=g), dt)
    kw = dict(use_ip_plus_cross_attention=True, use_fps_condition=True, fps_tensor=torch.tensor([8.0]))
    out = un(x.to(dev, dt), 981, ctx.to(dev, dt), reference_images_clip_feat=feat.to(dev, dt), **kw).sample
    with E.patched_kernels():
        want = ref(x, 981, ctx, reference_images_clip_feat=feat, **kw).sample
    err = rel(out, want)
    _record(f"unet_forward_w5_{str(dt).split('.')[-1]}", rel=err)
    assert out.shape == (2, 4, 8, 16, 32) and torch.isfinite(out.float()).all() and err < tol, err


def test_cross_view_masks_and_tables_on_the_device_match_the_reference_fixture():
    """SURVEY row a12 on the GPU: the cached per-resolution geometry WarpAttn uploads once -- both additive mask matrices
    (normal and antipodal), as handed to the attention kernel (16-bit, and the packed fp16 * log2(e) form), and the
    spherical positional tables -- against the REAL reference's masks (tests/golden/masks.npz) and the host-built tables."""
    from imagine360_amd import pano_geometry as G
    from imagine360_amd.mv_model import WarpAttn
    g = gold("masks.npz")
    dev = torch.device("cuda", 0)
    ph, eh, m = 8, 16, 20
    cams = {k: v[0] for k, v in S.icosahedron_cameras(90, ph * 8).items()}
    blk = WarpAttn(64).to(dev)
    ne, npx = eh * 2 * eh, ph * ph
    for tag in ("normal", "oppo"):
        b_e2p, b_p2e, pers_pe, equi_pe, packed = blk.geometry(ph, ph, eh, 2 * eh, cams, tag == "oppo", dev, torch.bfloat16)
        ref_e2p = g[f"pers_{tag}_{ph}"].reshape(m, ne, npx).permute(1, 0, 2).reshape(ne, m * npx)
        ref_p2e = g[f"equi_{tag}_{ph}"].reshape(m * npx, ne)
        assert packed and b_e2p.dtype == torch.float16 and b_e2p.is_cuda and b_e2p.shape == (ne, m * npx)
        # packed form = fp16((bf16(mask) + shift) * log2 e) (round 6: shifted so that the background is zero): undo both, allow the two roundings
        xt = blk.geometry_extra(ph, ph, eh, 2 * eh, cams, tag == "oppo", dev, torch.bfloat16)
        assert (b_e2p.float().cpu() / 1.4426950408889634 - xt["shift_e2p"] - ref_e2p).abs().max() < 6e-3
        assert (b_p2e.float().cpu() / 1.4426950408889634 - xt["shift_p2e"] - ref_p2e).abs().max() < 6e-3
        # the block maps: a clear bit <=> the 32 x 32 block of the packed matrix is all zero
        for mat, bm in ((b_e2p, xt["blocks_e2p"]), (b_p2e, xt["blocks_p2e"])):
            nq, nk = mat.shape
            nz = torch.zeros((-(-nq // 32) * 32, -(-nk // 32) * 32), dtype=torch.bool, device=dev)
            nz[:nq, :nk] = mat != 0
            want = nz.view(-1, 32, nz.shape[1] // 32, 32).any(3).any(1).cpu()
            words = bm.cpu().to(torch.int64) & 0xffffffff
            got = ((words[:, :, None] >> torch.arange(32)) & 1).reshape(words.shape[0], -1)[:, :want.shape[1]].bool()
            assert torch.equal(got, want)
            assert float(want.float().mean()) < 0.6          # most blocks are background
        pc, ec = G.spherical_coords(ph, ph, eh, 2 * eh, cams)
        assert torch.equal(pc, g[f"pers_coords_{ph}"]) and torch.equal(ec, g[f"equi_coords_{ph}"])
        from im360_oracle import geometry as OG
        assert (pers_pe.float().cpu() - OG.spherical_pe(pc, 16).reshape(-1, 64)).abs().max() < 5e-3     # bf16 table of values in [-1, 1]
        assert (equi_pe.float().cpu() - OG.spherical_pe(ec, 16).reshape(-1, 64)).abs().max() < 5e-3
    # the cache returns the resident tensors (no rebuild, no upload) on the second call
    again = blk.geometry(ph, ph, eh, 2 * eh, cams, False, dev, torch.bfloat16)
    assert again[0].data_ptr() == blk.geometry(ph, ph, eh, 2 * eh, cams, False, dev, torch.bfloat16)[0].data_ptr()


def test_cross_view_masks_at_the_cfg5_level_1_size_on_the_device():
    """BASELINE cfg5's largest WarpAttn resolution (equirect 64 x 128, 20 views of 32 x 32; 8192 x 20 480 entries per matrix):
    the masks built ON THE DEVICE (pano_geometry.cross_view_bias) against the REAL get_merged_masks' fixture, then the cached
    geometry as the attention kernel receives it for fp16 (packed fp16 * log2 e matrices, both variants) and the spherical
    positional tables at 640 channels (160 frequencies, top frequency 5000^(159 / 64)) against the oracle's fp32 tables."""
    from helpers import check_masks_64x128x32
    from im360_oracle import geometry as OG
    from imagine360_amd import pano_geometry as G
    from imagine360_amd.mv_model import WarpAttn
    dev = torch.device("cuda", 0)
    cams = {k: v[0] for k, v in S.icosahedron_cameras(90, 512).items()}
    g = gold("masks_64x128x32.npz")
    blk = WarpAttn(640).to(dev)
    obs = {}
    for tag in ("normal", "oppo"):
        b_e2p, b_p2e = G.cross_view_bias(32, 32, 64, 128, cams, tag == "oppo", dev)
        assert b_e2p.is_cuda
        obs.update({f"{tag}_{k}": v for k, v in check_masks_64x128x32(tag, b_e2p, b_p2e).items()})
        del b_e2p, b_p2e
        k_e2p, k_p2e, pers_pe, equi_pe, packed = blk.geometry(32, 32, 64, 128, cams, tag == "oppo", dev, torch.float16)
        assert packed and k_e2p.dtype == torch.float16 and k_e2p.shape == (8192, 20480) and k_p2e.shape == (20480, 8192)
        for name, mat, rows in (("e2p", k_e2p, g["rows_e2p"]), ("p2e", k_p2e, g["rows_p2e"])):
            # packed form = fp16((fp16(mask) + shift) * log2 e): undo scale and shift, allow the two fp16 roundings
            shift = blk.geometry_extra(32, 32, 64, 128, cams, tag == "oppo", dev, torch.float16)[f"shift_{name}"]
            err = float((mat[rows.to(dev)].float().cpu() / 1.4426950408889634 - shift - g[f"{name}_{tag}_rows"].float()).abs().max())
            obs[f"{tag}_{name}_packed_rows_max_abs"] = err
            assert err < 2e-3, (tag, name, err)
    pc, ec = G.spherical_coords(32, 32, 64, 128, cams)
    obs["pers_pe_max_abs"] = float((pers_pe.float().cpu() - OG.spherical_pe(pc, 160).reshape(-1, 640)).abs().max())
    obs["equi_pe_max_abs"] = float((equi_pe.float().cpu() - OG.spherical_pe(ec, 160).reshape(-1, 640)).abs().max())
    _record("masks_64x128x32_on_the_device", **obs)
    assert obs["pers_pe_max_abs"] < 1e-3 and obs["equi_pe_max_abs"] < 1e-3, obs          # fp16 table of values in [-1, 1]


def test_warp_attn_block_at_cfg5_level_1_size_vs_oracle():
    """ONE WarpAttn block at BASELINE cfg5's level-1 size -- equirect 64 x 128, 20 views of 32 x 32, 640
    channels (20 heads of 32), one frame, both mask variants -- through the product (positional tables, LayerNorm + PE, fused QKV,
    the 8192 x 20 480 and 20 480 x 8192 attentions with the shifted masks and their block maps, output projection, LayerNorm-folded
    GEGLU feed-forward) against the fp32 oracle (im360_oracle.mv.warp_attn, attn_perspano.py:22-99) on the same weights and inputs.
    The oracle gets the masks the product built on the device; the test above pins exactly those on the REAL get_merged_masks."""
    from im360_oracle import mv as OMV
    from imagine360_amd import pano_geometry as G
    from imagine360_amd.mv_model import WarpAttn
    dt, dev = torch.float16, torch.device("cuda", 0)
    ph, eh, m, c = 32, 64, 20, 640
    cams = {k: v[0] for k, v in S.icosahedron_cameras(90, 512).items()}
    g = torch.Generator().manual_seed(17)
    blk = WarpAttn(c)
    with torch.no_grad():
        for name, prm in blk.named_parameters():             # (the reference zero-initialises the output projections: fill everything)
            if prm.dim() >= 2:
                prm.copy_(torch.randn(prm.shape, generator=g) * prm.shape[-1] ** -0.5)
            elif name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("norm3.weight"):
                prm.copy_(1.0 + 0.1 * torch.randn(prm.shape, generator=g))
            else:
                prm.copy_(0.05 * torch.randn(prm.shape, generator=g))
        blk = blk.to(dt)                                     # the weights both sides use are the fp16-rounded ones
    sd = {"w." + k: v.float() for k, v in blk.state_dict().items()}
    pers = _q(torch.randn(m, ph, ph, c, generator=g), dt)    # (b m) f = 20 images of one frame, channels-last
    equi = _q(torch.randn(1, eh, 2 * eh, c, generator=g), dt)
    gblk = blk.to(dev)
    obs = {}
    for tag, opp in (("normal", False), ("oppo", True)):
        gp, ge = gblk.forward_cl(pers.to(dev, dt), equi.to(dev, dt), cams, 1, opposite=opp)
        x = gblk.geometry_extra(ph, ph, eh, 2 * eh, cams, opp, dev, dt)
        assert x["blocks_e2p"].shape == (eh * 2 * eh // 32, m * ph * ph // 1024) and x["blocks_p2e"].shape == (m * ph * ph // 32, eh * 2 * eh // 1024)
        b_e2p, b_p2e = G.cross_view_bias(ph, ph, eh, 2 * eh, cams, opp, dev)
        masks = (b_e2p.float().cpu().reshape(eh, 2 * eh, m, ph, ph).permute(2, 0, 1, 3, 4), b_p2e.float().cpu().reshape(m, ph, ph, eh, 2 * eh))
        del b_e2p, b_p2e
        op, oe = OMV.warp_attn(sd, "w.", pers.float().permute(0, 3, 1, 2).unsqueeze(2), equi.float().permute(0, 3, 1, 2).unsqueeze(2), cams, opposite=opp, masks=masks)
        obs[f"{tag}_pers"] = rel(gp, op[:, :, 0].permute(0, 2, 3, 1))
        obs[f"{tag}_equi"] = rel(ge, oe[:, :, 0].permute(0, 2, 3, 1))
        # no 32-row block of tokens off by more than a few times the tensor's error (a wrong mask block / skipped foreground block is O(1))
        obs[f"{tag}_equi_worst_block"] = float(((ge.float().cpu() - oe[:, :, 0].permute(0, 2, 3, 1)).reshape(-1, 32, c).norm(dim=(1, 2))
                                                / oe[:, :, 0].permute(0, 2, 3, 1).reshape(-1, 32, c).norm(dim=(1, 2))).max())
    _record("warp_attn_block_64x128x32", **obs)
    # observed on MI355X: 4.4e-4 everywhere (tensor and worst 32-row block): the bounds are 3 x that, not a tolerance picked by hand
    assert max(v for k, v in obs.items() if not k.endswith("worst_block")) < 1.3e-3, obs
    assert max(v for k, v in obs.items() if k.endswith("worst_block")) < 1.5e-3, obs


def test_preprocessing_warps_vs_oracle():
    """SURVEY row N3 on the GPU: im360_remap_cubic_wrap_u8 == the oracle's restatement of cv2.remap(INTER_CUBIC, BORDER_WRAP)
    bit for bit -- random maps incl. coordinates outside the image, exact .5 / integer positions and 1 / 3 / 4 channels --
    and the script-level helpers built on it (process_equi, pers2pano_frames, Equirectangular / Perspective, get_anchor_target)
    == the same compositions of the oracle.  (The maps themselves are pinned on the real reference in the CPU tests; the
    bicubic arithmetic is parity-unpinned, see the oracle header.)"""
    import numpy as np
    from im360_oracle import preprocess as OPP
    from imagine360_amd import preprocess as PP
    rng = np.random.default_rng(3)
    for C in (1, 3, 4):
        img = rng.integers(0, 256, (2, 37, 53, C), dtype=np.uint8)
        mx = (rng.random((3, 20, 31)) * 80 - 15).astype(np.float32)
        my = (rng.random((3, 20, 31)) * 60 - 12).astype(np.float32)
        mx[0, 0, :8] = [0.0, 0.5, 1.5, 52.0, 52.984375, -0.015625, 53.0, -1.0]
        my[0, 0, :8] = [0.0, 0.5, 2.5, 36.0, 36.5, -0.5, 37.0, -4.0]
        got = PP.remap(torch.from_numpy(img).cuda(), torch.from_numpy(mx).cuda(), torch.from_numpy(my).cuda()).cpu().numpy()
        for n in range(2):
            for m in range(3):
                assert np.array_equal(got[n, m], OPP.remap_cubic_wrap_u8(img[n], mx[m], my[m])), (C, n, m)
    pano = torch.from_numpy(rng.random((2, 3, 64, 128)).astype(np.float32) * 2 - 1)
    thetas, phis = np.array([0.0, 72.0, -108.0, 180.0]), np.array([0.0, 52.6, -10.8, 90.0])
    out = PP.process_equi(pano, torch.from_numpy(thetas)[None], torch.from_numpy(phis)[None], pers_resolution=32)
    ref = OPP.process_equi(pano.numpy(), thetas, phis, pers_resolution=32)
    assert out.shape == (2, 4, 3, 32, 32) and np.array_equal(out.numpy(), ref)
    valid = PP.process_equi(pano.abs(), thetas, phis, pers_resolution=32, back_norm=False)
    assert np.array_equal(valid.numpy(), OPP.process_equi(pano.abs().numpy(), thetas, phis, pers_resolution=32, back_norm=False))
    frames = rng.integers(0, 256, (3, 24, 40, 3), dtype=np.uint8)
    pf, pm = PP.pers2pano_frames(frames, [0.0, 17.5, 0.0], pano_H=48, pano_W=96)
    rf, rm = OPP.pers2pano_frames(frames, [0.0, 17.5, 0.0], pano_h=48, pano_w=96)
    assert np.array_equal(pf, rf) and np.array_equal(pm, rm)
    one = PP.Equirectangular(frames[0]).GetPerspective(90, 30.0, -20.0, 16, 16)
    assert np.array_equal(one, OPP.get_perspective(frames[0], 90, 30.0, -20.0, 16, 16))
    canvas, mask = PP.Perspective(frames[1], 90, 0, 10.0).GetEquirec(48, 96)
    rc, rmask = OPP.get_equirec(frames[1], 90, 0, 10.0, 48, 96)
    assert np.array_equal(canvas, rc) and np.array_equal(mask, rmask)
    vid = torch.from_numpy(rng.random((2, 3, 64, 128)).astype(np.float32) * 2 - 1).cuda()
    a, ap, tgt, masks, rel_pos, pitchs = PP.get_anchor_target(vid, [5.0, -12.0])
    assert a.shape == (1, 2, 3, 256, 256) and ap.shape == (1, 2, 3, 32, 32) and masks.shape == (1, 2, 1, 64, 128)
    assert rel_pos.shape == (1, 2, 6) and pitchs.shape == (1, 2) and torch.equal(tgt, vid[None])
    for i, ph in enumerate([5.0, -12.0]):
        fr = ((vid[i].permute(1, 2, 0).cpu().numpy() + 1) / 2 * 255).astype(np.uint8)
        assert np.array_equal(((ap[0, i].permute(1, 2, 0).cpu().numpy() + 1) * 127.5).round().astype(np.uint8), OPP.get_perspective(fr, 90, 0, ph, 32, 32))
        _, _, m = OPP.p2e_maps(90, 0, ph, 32, 32, 64, 128)
        assert np.array_equal(masks[0, i, 0].cpu().numpy(), (1 - m).astype(np.float32))
        top, left, rw, rh = OPP.get_maxrec_cord(m)
        assert rel_pos[0, i].tolist() == [int(32 - (2 * top + rh) / 2), int(64 - (2 * left + rw) / 2), rh, rw, 64, 128]
