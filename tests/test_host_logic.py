"""Host logic of the product (imagine360_amd) on CPU: the HIP kernels are replaced by the torch stand-ins
of tests/_emu_kernels.py (the product itself has no CPU path), everything else -- block walk, channels-last
layouts, fused/hoisted conditioning, geometry caches, RNG order, scheduler, pipeline -- is the shipped code,
checked against fixtures generated from the REAL reference.  fp32 -> relative L2 <= 5e-5."""
import json
import os
import random
import re

import pytest
import torch

import _emu_kernels as E
from helpers import GOLDEN, gold, op_inputs, rel
from imagine360_amd import configs, pano_geometry as G, synthetic as S
from imagine360_amd.layers import from_cl, to_cl
from imagine360_amd.scheduler import DDIMScheduler

torch.set_grad_enabled(False)
TOL = 5e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mv():
    m = configs.build_mv_model(5, device="cpu", dtype=torch.float32, xformers=False)
    m.noise_on_host = True
    return m


def test_state_dict_keys_match_reference_checkpoints():
    keys = json.load(open(os.path.join(GOLDEN, "state_keys_full.json")))
    with torch.device("meta"):
        from imagine360_amd.mv_model import MultiViewBaseModel
        from imagine360_amd.vae import AutoencoderKL
        mvm = MultiViewBaseModel(configs.build_unet(1), configs.build_unet(1))
        vae = AutoencoderKL(**configs.vae_config(1))
    assert {k: list(v.shape) for k, v in mvm.state_dict().items()} == keys["mv"]
    assert {k: list(v.shape) for k, v in vae.state_dict().items()} == keys["vae"]


def test_zero_initialised_layers_like_reference():
    un = configs.build_unet(5)
    assert un.fps_embedding.linear_2.weight.abs().sum() == 0
    assert un.down_blocks[0].motion_modules[0].temporal_transformer.proj_out.weight.abs().sum() == 0
    from imagine360_amd.mv_model import WarpAttn
    w = WarpAttn(64)
    assert w.transformer.attn1.to_out.weight.abs().sum() == 0 and w.transformer.ff.net[2].weight.abs().sum() == 0


def test_blocks_against_reference(mv):
    g, I, un = gold("ops_w5.npz"), op_inputs(), mv.pano_unet
    with E.patched_kernels():
        x, f = to_cl(I["x"])
        assert rel(from_cl(un.down_blocks[0].resnets[0].forward_cl(x, I["emb"], f, pano=True), f), g["resnet_pano"]) < TOL
        x2 = torch.cat([I["x"], I["x"].flip(1), 0.5 * I["x"].roll(3, 1)], 1)
        assert rel(un.up_blocks[3].resnets[0](x2, I["emb"]), g["resnet_shortcut"]) < TOL
        T = un.down_blocks[0].attentions[0]
        assert rel(from_cl(T.forward_cl(x, I["ctx"], f), f), g["spatial_cpu"]) < TOL
        un.enable_xformers_memory_efficient_attention()
        assert rel(T(I["x"], encoder_hidden_states=I["ctx"]).sample, g["spatial_xf"]) < TOL
        un.disable_xformers_memory_efficient_attention()
        assert rel(un.down_blocks[0].motion_modules[0](I["x"], I["emb"], I["ctx"]), g["motion"]) < TOL
        assert rel(from_cl(un.down_blocks[0].downsamplers[0].forward_cl(x, pano=True), f), g["down_pano"]) < TOL
        x3, _ = to_cl(I["x3"])
        assert rel(from_cl(un.up_blocks[0].upsamplers[0].forward_cl(x3, pano=True), f), g["up_pano"]) < TOL
        l9, _ = to_cl(I["lat9"])
        assert rel(from_cl(un.conv_in_cl(l9, pano=True), f), g["conv_in_pano"]) < TOL
        assert rel(un.ip_tokens_clean(I["feat"]), g["ip_tokens"]) < 1e-4
        cams = {k: v[0] for k, v in S.icosahedron_cameras(90, 64).items()}
        for tag, seed in (("normal", 0), ("oppo", 1)):
            random.seed(seed)                                   # the coin is drawn inside, like the reference
            p, e = mv.cp_blocks_encoder[0](I["px"], I["ex"], cams)
            assert rel(p, g["warp_pers_" + tag]) < TOL and rel(e, g["warp_equi_" + tag]) < TOL


def test_folded_layernorm_and_skip_pair_paths_against_reference(mv):
    """The branches the host code takes on the GPU at large token counts -- token-major Linears on the MFMA kernel with the
    rows' LayerNorm statistics written by the producer, every LayerNorm of the spatial / temporal / cross-view transformer
    blocks folded into the GEMM that consumes it (gamma-scaled weights, c1 / c2 vectors, the frame-PE table pushed through
    the QKV projection), fused GEGLU -- forced on CPU with the torch stand-ins, against the REAL reference's fixtures."""
    g, I, un = gold("ops_w5.npz"), op_inputs(), mv.pano_unet
    calls = []
    with E.patched_kernels(), E.routed_gemms():
        from imagine360_amd import kernels
        for name in ("linear_ln", "linear_geglu_ln", "linear", "conv1x1_cat"):
            orig = getattr(kernels, name)
            setattr(kernels, name, (lambda o, n: (lambda *a, **k: (calls.append(n), o(*a, **k))[1]))(orig, name))
        x, f = to_cl(I["x"])
        T = un.down_blocks[0].attentions[0]
        assert rel(from_cl(T.forward_cl(x, I["ctx"], f), f), g["spatial_cpu"]) < TOL
        # (four plain linears: proj_in and the two attention out-projections with row statistics, and -- since round 4 -- proj_out,
        #  whose epilogue leaves the GroupNorm partial sums of the block's output for the next module's norm)
        assert calls.count("linear_ln") == 2 and calls.count("linear_geglu_ln") == 1 and calls.count("linear") == 4, calls
        del calls[:]
        assert rel(un.down_blocks[0].motion_modules[0](I["x"], I["emb"], I["ctx"]), g["motion"]) < TOL
        assert calls.count("linear_ln") == 0 and calls.count("linear_geglu_ln") == 1, calls      # 128 pixels per frame: the PE table needs 256 | pixels
        xm = torch.randn(1, 64, 4, 16, 16, generator=torch.Generator().manual_seed(3))        # 256 pixels per frame: QKV folded too, PE as a table
        del calls[:]
        folded = un.down_blocks[0].motion_modules[0](xm, None, None)
        assert calls.count("linear_ln") == 2 and calls.count("linear_geglu_ln") == 1, calls
        x2 = torch.cat([I["x"], I["x"].flip(1), 0.5 * I["x"].roll(3, 1)], 1)
        xa, _ = to_cl(x2[:, :128])
        xb, _ = to_cl(x2[:, 128:])
        del calls[:]
        out = un.up_blocks[3].resnets[0].forward_cl((xa, xb), I["emb"], f)            # (x, skip) pair, never concatenated
        assert rel(from_cl(out, f), g["resnet_shortcut"]) < TOL and "conv1x1_cat" in calls
        cams = {k: v[0] for k, v in S.icosahedron_cameras(90, 64).items()}
        random.seed(0)
        del calls[:]
        p, e = mv.cp_blocks_encoder[0](I["px"], I["ex"], cams)
        assert rel(p, g["warp_pers_normal"]) < TOL and rel(e, g["warp_equi_normal"]) < TOL and calls.count("linear_geglu_ln") == 2
    with E.patched_kernels():
        assert rel(folded, un.down_blocks[0].motion_modules[0](xm, None, None)) < 2e-5          # == LayerNorm kernel + PE add + GEMM
    inp = S.mv_inputs(frames=8, pano_hw=(32, 64), pers_hw=(16, 16), seed=0, sam_frames=16)
    cams = S.icosahedron_cameras(90, 128)
    mv.unet.disable_xformers_memory_efficient_attention()
    mv.pano_unet.disable_xformers_memory_efficient_attention()
    with E.patched_kernels(), E.routed_gemms():
        torch.manual_seed(7)
        random.seed(7)
        pers, pano = mv(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True, **inp)
    gm = gold("mv_forward_w5.npz")
    assert rel(pano, gm["pano"]) < TOL and rel(pers[:, [0, 7, 13, 19]], gm["pers_views"]) < TOL


def test_cross_view_geometry_against_reference():
    g = gold("masks.npz")
    for ph, eh in ((4, 8), (8, 16)):
        cams = {k: v[0] for k, v in S.icosahedron_cameras(90, ph * 8).items()}
        m, ne, npx = 20, eh * 2 * eh, ph * ph
        for tag in ("normal", "oppo"):
            b_e2p, b_p2e = G.cross_view_bias(ph, ph, eh, 2 * eh, cams, tag == "oppo")
            ref_e2p = g[f"pers_{tag}_{ph}"].reshape(m, ne, npx).permute(1, 0, 2).reshape(ne, m * npx)
            ref_p2e = g[f"equi_{tag}_{ph}"].reshape(m * npx, ne)
            assert (b_e2p - ref_e2p).abs().max() < 1e-3 and (b_p2e - ref_p2e).abs().max() < 1e-3     # fp16 fixture
        pc, ec = G.spherical_coords(ph, ph, eh, 2 * eh, cams)
        assert torch.equal(pc, g[f"pers_coords_{ph}"]) and torch.equal(ec, g[f"equi_coords_{ph}"])


def test_cross_view_geometry_at_the_cfg5_level_1_size_against_reference():
    """VERDICT r4 missing #2: the largest mask BASELINE cfg5 (1024 x 2048) builds -- equirect 64 x 128 against 20 views of
    32 x 32, 8192 x 20 480 entries per direction and variant -- from the product's footprint scatter (pano_geometry.cross_view_bias,
    no one-hot tensors) against the REAL get_merged_masks (two 5.4 GB one-hots per variant there; tests/golden/masks_64x128x32.npz)."""
    from helpers import check_masks_64x128x32
    cams = {k: v[0] for k, v in S.icosahedron_cameras(90, 512).items()}
    for tag in ("normal", "oppo"):
        b_e2p, b_p2e = G.cross_view_bias(32, 32, 64, 128, cams, tag == "oppo")
        assert b_e2p.shape == (8192, 20480) and b_p2e.shape == (20480, 8192)
        check_masks_64x128x32(tag, b_e2p, b_p2e)
        del b_e2p, b_p2e


def test_pad_pano_api():
    x = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)
    p = G.pad_pano(x, 2)
    assert p.shape[-1] == 9 and torch.equal(p[..., :2], x[..., -2:]) and torch.equal(p[..., -2:], x[..., :2])
    assert torch.equal(G.unpad_pano(p, 2), x) and G.pad_pano(x, 0) is x


def test_scheduler_against_reference():
    g = gold("ddim.npz")
    sch = DDIMScheduler(**configs.NOISE_SCHEDULER_KWARGS)
    assert torch.equal(sch.alphas_cumprod, g["alphas_cumprod"])
    gen = torch.Generator().manual_seed(5)
    x, v = torch.randn(1, 4, 2, 8, 16, generator=gen), torch.randn(1, 4, 2, 8, 16, generator=gen)
    for n in (4, 25, 50):
        sch.set_timesteps(n)
        assert torch.equal(sch.timesteps, g[f"timesteps_{n}"].long())
        for idx in (0, n - 1):
            assert rel(sch.step(v, sch.timesteps[idx], x).prev_sample, g[f"step_{n}_{idx}"]) < 1e-5
    with E.patched_kernels():           # fused CFG + update == guidance then step
        u, c = torch.randn(1, 4, 2, 8, 16, generator=gen), torch.randn(1, 4, 2, 8, 16, generator=gen)
        t = sch._timesteps_host[3]
        assert rel(sch.fused_cfg_step(u, c, 7.5, t, x), sch.step(u + 7.5 * (c - u), t, x).prev_sample) < 1e-6


def test_vae_against_reference():
    g = gold("vae_w4.npz")
    vae = configs.build_vae(4, device="cpu", dtype=torch.float32)
    gen = torch.Generator().manual_seed(9)
    x = torch.rand(2, 3, 64, 96, generator=gen) * 2 - 1
    z = torch.randn(2, 4, 8, 20, generator=gen)
    with E.patched_kernels():
        assert rel(vae.encode(x, 2).latent_dist.parameters, g["moments"]) < TOL      # int return_dict like the pipeline
        assert rel(vae.decode(z).sample, g["decoded"]) < TOL
        vae.enable_slicing()
        assert rel(vae.decode(z).sample, g["decoded"]) < TOL


@pytest.mark.parametrize("xf,name", [(False, "mv_forward_w5.npz"), (True, "mv_forward_w5_xf.npz")])
def test_mv_forward_against_reference(mv, xf, name):
    g = gold(name)
    inp = S.mv_inputs(frames=8, pano_hw=(32, 64), pers_hw=(16, 16), seed=0, sam_frames=16)
    cams = S.icosahedron_cameras(90, 128)
    (mv.unet.enable_xformers_memory_efficient_attention if xf else mv.unet.disable_xformers_memory_efficient_attention)()
    (mv.pano_unet.enable_xformers_memory_efficient_attention if xf else mv.pano_unet.disable_xformers_memory_efficient_attention)()
    with E.patched_kernels():
        torch.manual_seed(7)
        random.seed(7)
        mv.taps = {}
        pers, pano = mv(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True, **inp)
    assert rel(pano, g["pano"]) < TOL and rel(pers[:, [0, 7, 13, 19]], g["pers_views"]) < TOL
    if not xf:
        for n, (tp, te) in mv.taps.items():
            assert rel(from_cl(te, 8)[:, ::4, ::3], g[f"tap_{n}_equi"]) < TOL
    mv.taps = None
    mv.unet.disable_xformers_memory_efficient_attention()
    mv.pano_unet.disable_xformers_memory_efficient_attention()


def test_pipeline_against_reference(mv):
    from imagine360_amd.pipeline import AnimationPipeline
    g = gold("pipeline_w5.npz")
    vae = configs.build_vae(4, device="cpu", dtype=torch.float32)
    pipe = AnimationPipeline(vae, None, None, mv.unet, mv.pano_unet, mv, DDIMScheduler(**configs.NOISE_SCHEDULER_KWARGS), None, "SAM")
    pipe.rng, pipe._no_progress = "host", True
    pipe.enable_vae_slicing()
    vb = S.video_batch(frames=16, pano_hw=(256, 512), seed=0)
    cond = S.conditioning(frames=16, seed=0)
    trace = []
    with E.patched_kernels():
        torch.manual_seed(21)
        random.seed(21)
        vid = pipe("synthetic", num_inference_steps=2, guidance_scale_text=7.5, negative_prompt="", latents_dtype=torch.float32,
                   video_batch=vb, use_outpaint=True, use_ip_plus_cross_attention=True, use_fps_condition=True,
                   ip_plus_condition="video", prompt_embeds=(cond["text_pano"], cond["text_pers"]),
                   sam_features=(cond["sam_pano"], cond["sam_pers"]), trace=trace).videos
    for i, t in enumerate(trace):
        assert rel(t, g[f"pano_latent_{i}"]) < 1e-4
    assert rel(vid[:, :, ::3, ::4, ::4], g["video_sub"]) < 1e-3
    st = torch.stack([vid.mean(dim=(0, 1, 3, 4)), vid.std(dim=(0, 1, 3, 4))])
    assert rel(st, g["video_frame_stats"]) < 1e-4
    assert vid.dtype == torch.float32 and vid.shape == (1, 3, 16, 256, 512)


def test_geglu_interleave_and_gemm_routing_rules():
    """Host side of the fused GEMM paths: the value/gate row interleave that the GEGLU epilogue assumes (accumulator
    blocks 2i / 2i+1 = value / gate of the same 32 channels), and the shape rules that route a Linear to the kernel."""
    import torch.nn.functional as F
    from imagine360_amd import kernels, layers
    g = torch.Generator().manual_seed(3)
    I, Kd = 256, 64
    w, b = torch.randn(2 * I, Kd, generator=g), torch.randn(2 * I, generator=g)
    wp, bp = kernels.interleave_geglu(w, b)
    x = torch.randn(5, Kd, generator=g)
    h = F.linear(x, wp, bp).reshape(5, I // 32, 2, 32)          # packed GEMM output: [.., block, value|gate, 32]
    fused = (h[:, :, 0] * F.gelu(h[:, :, 1])).reshape(5, I)
    ref = F.linear(x, w, b)
    assert torch.allclose(fused, ref[:, :I] * F.gelu(ref[:, I:]), atol=1e-5)
    # routing: level-0 / pers level-1 token counts of cfg2 take the kernel, deeper levels stay on hipBLASLt
    assert layers._gemm_kernel_pays(655360, 320, 320) and layers._gemm_kernel_pays(262144, 320, 960)
    assert layers._gemm_kernel_pays(163840, 2560, 640) and not layers._gemm_kernel_pays(40960, 5120, 1280)
    assert not layers._gemm_kernel_pays(655360, 320, 256) and not layers._gemm_kernel_pays(655360, 96, 320)
    # on CPU tensors the helpers are plain torch (the emulated product path of these tests)
    lin = torch.nn.Linear(Kd, 32)
    r = torch.randn(5, 32, generator=g)
    assert torch.allclose(layers.linear_residual(lin, x, r), lin(x) + r)
    assert torch.allclose(layers.linear(lin, x), lin(x))


def test_no_cpu_fallback_in_product():
    from imagine360_amd import kernels
    x = torch.zeros(1, 16, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        kernels.attention(x, x, x, heads=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        kernels.group_norm(torch.zeros(1, 2, 2, 32), torch.ones(32), torch.zeros(32), 32, 1e-5)


def test_dropin_aliases():
    """Without a reference checkout on sys.path every aliased module is registered synthetically and the hot-path
    imports of inference_dual_p2e.py:20-38 resolve to imagine360_amd; uninstall() removes them again."""
    import sys
    from imagine360_amd import dropin
    if any(n in sys.modules and not getattr(sys.modules[n], "__im360_alias__", False) for n in ("diffusers", "animatediff", "src")):
        pytest.skip("reference modules already imported in this interpreter")
    try:
        how = dropin.install()
        assert set(how.values()) == {"synthetic"}
        from animatediff.models.unet import UNet3DConditionModel
        from animatediff.pipelines.pipeline_animation_inference_dual import AnimationPipeline
        from diffusers import AutoencoderKL, DDIMScheduler
        from diffusers.utils.import_utils import is_xformers_available
        from src.models.MVGenModel import MultiViewBaseModel
        from src.modules.utils import flush
        from src.utils.pano import pad_pano, unpad_pano
        import imagine360_amd.unet3d as U3
        assert UNet3DConditionModel is U3.UNet3DConditionModel and callable(pad_pano) and callable(unpad_pano)
        assert AnimationPipeline.__call__ and AutoencoderKL and DDIMScheduler and MultiViewBaseModel
        assert flush() is None and not is_xformers_available()       # answered per caller: this test module is not the script ...
        assert eval("is_xformers_available()", {"__name__": "__main__", "is_xformers_available": is_xformers_available})   # ... the script is
    finally:
        dropin.uninstall()
    assert "animatediff.models.unet" not in sys.modules and "diffusers" not in sys.modules


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout (build container only)")
def test_dropin_overlay_runs_the_reference_script():
    """Tier-1 drop-in: with dropin.install() overlaid on the reference checkout, the UNMODIFIED
    inference_dual_p2e.py executes its import block (:1-45) and its own model-loading code (:175-245, :382-474 --
    from_pretrained_2d incl. the conv_in widening, checkpoint load with 'module.' prefixes, LoRA merge by dotted name,
    MultiViewBaseModel / AnimationPipeline construction) on imagine360_amd classes at reduced width; names outside the
    hot path keep coming from the checkout; UNet3DConditionModel.forward == the reference's forward on the same weights.
    Runs in a subprocess (tests/dropin_harness.py) so the reference never enters this interpreter."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "dropin_harness.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["how"].pop("diffusers.utils.import_utils") == "scoped probe" and set(out["how"].values()) == {"overlay"}
    assert all(out["names"].values()), out["names"]
    # the xformers probe is answered per caller: True for the script (above), the library's own answer for diffusers modules
    # imported after the overlay (on a box without xformers their module-level `import xformers` must not run)
    fd = out["fresh_diffusers_import"]
    assert fd["ok"] and fd["probe_for_diffusers"] == fd["library_answer"], fd
    b = out["built"]
    assert all(t.startswith("imagine360_amd.") for t in b["types"])
    assert b["conv_in_widened"][1] == 9 and b["conv_in_extra_channels_zero"] and b["motion_ckpt_loaded"]
    assert b["lora_merged"] < 1e-6 and b["fused_qkv_sees_lora"] < 1e-6 and b["xformers_flag"] and b["slicing"]
    assert b["mv_unexpected"] == 0
    u = out["unet_forward"]
    assert u["finite"] and u["ref_missing"] == 0 and u["ref_unexpected"] == 0 and u["rel_vs_reference_forward"] < 5e-5
    assert out["uninstalled"]


def test_conditioning_producers_prompt_and_sam_paths():
    """the pipeline's own conditioning producers (out of the kernel hot path, but part of the kept
    API): `_encode_prompt` (uncond first, one encode per prompt -- the reference encodes the same prompt once per view,
    pipeline...dual.py:628,655) and `_sam_features` (uint8 conversion, chunks of 8 frames through a SamPredictor)."""
    import sys
    import types
    import numpy as np
    from imagine360_amd.pipeline import AnimationPipeline

    calls = {"enc": [], "sam": []}

    class Tok:
        model_max_length = 77

        def __call__(self, s, **kw):
            assert kw["padding"] == "max_length" and kw["max_length"] == 77 and kw["truncation"]
            ids = torch.tensor([[len(x) for _ in range(77)] for x in s])
            return types.SimpleNamespace(input_ids=ids)

    class Enc(torch.nn.Module):
        def forward(self, ids):
            calls["enc"].append(ids.shape)
            return (ids.float().unsqueeze(-1).expand(-1, -1, 8),)

    class Predictor:
        class _T:
            def apply_image(self, im):
                assert im.dtype == np.uint8 and im.shape[-1] == 3
                return im

        def __init__(self, model):
            self.transform = self._T()

        def set_torch_image(self, x, hw):
            calls["sam"].append(tuple(x.shape))
            self._n = x.shape[0]

        def get_image_embedding(self):
            return torch.ones(self._n, 256, 64, 64)

    saved = sys.modules.get("segment_anything")
    sys.modules["segment_anything"] = types.SimpleNamespace(SamPredictor=Predictor)
    try:
        vae = types.SimpleNamespace(config=types.SimpleNamespace(block_out_channels=(1, 2, 3, 4)), device=torch.device("cpu"))
        pipe = AnimationPipeline(vae, Enc(), Tok(), None, None, None, None, image_encoder=object(), image_encoder_name="SAM")
        emb = pipe._encode_prompt(["a street at night"], "cpu", 1, True, [None])
        assert emb.shape == (2, 77, 8) and len(calls["enc"]) == 2            # text + negative, one call each
        assert emb[0, 0, 0] == 0 and emb[1, 0, 0] == len("a street at night")      # uncond ("" -> length 0) first
        feats = pipe._sam_features(torch.zeros(1, 16, 3, 32, 32))
        assert feats.shape == (1, 16, 4096, 256) and calls["sam"] == [(8, 3, 32, 32)] * 2
    finally:
        if saved is None:
            sys.modules.pop("segment_anything", None)
        else:
            sys.modules["segment_anything"] = saved


def test_sr_patch_module_mirrors_the_reference_helpers():
    """imagine360_amd.sr_patch on the torch stand-in of the pad kernel: names, argument meaning, identity
    for padding 0, view semantics of the unpad and the NotImplementedError of the reference for other ranks."""
    import numpy as np
    import os
    from helpers import GOLDEN
    from imagine360_amd import sr_patch
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "sr_pad.npz")).items()}
    with E.patched_kernels():
        import imagine360_amd.kernels as K
        saved, K._dev = K._dev, lambda *a: None
        try:
            assert torch.equal(sr_patch.padding_pano(g["lat"], latent=True), g["lat_pad16"])
            assert torch.equal(sr_patch.padding_pano(g["vid"]), g["vid_pad128"])
            assert torch.equal(sr_patch.circular_pad(g["fr"], (3, 5, 2, 4)), g["fr_fit"])
        finally:
            K._dev = saved
    un = sr_patch.unpadding_pano(g["vid_pad128"])
    assert torch.equal(un, g["vid"]) and un.data_ptr() != g["vid"].data_ptr() and not un.is_contiguous()       # a slice, like the reference
    assert sr_patch.padding_pano(g["lat"], padding=0, latent=True) is g["lat"]
    with pytest.raises(NotImplementedError):
        sr_patch.padding_pano(g["lat"][0, 0, 0], latent=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sr_patch.padding_pano(g["lat"], latent=True)                 # the product path needs the HIP extension


def test_preprocess_maps_table_and_rectangle_against_reference():
    """imagine360_amd.preprocess, the parts that run on the host: sampling maps and masks == the real
    reference's (fixture preproc.npz, incl. the digest of the 20 production maps), the weight table == the oracle's
    independent construction, im360_max_rect (C, host) == the real get_maxrec_cord incl. all-ones / all-zeros masks and a
    real P2E footprint."""
    import hashlib
    import numpy as np
    import os
    from helpers import GOLDEN
    from im360_oracle import preprocess as OPP
    from imagine360_amd import preprocess as PP
    g = np.load(os.path.join(GOLDEN, "preproc.npz"))
    for n, (th, ph) in enumerate([(0.0, 0.0), (36.0, 52.6), (-108.0, -10.8), (180.0, 90.0), (72.0, -52.6)]):
        lon, lat = PP.e2p_maps(90.0, th, ph, 32, 32, 64, 128)
        assert np.array_equal(lon, g[f"e2p_lon_{n}"]) and np.array_equal(lat, g[f"e2p_lat_{n}"])
    h = hashlib.sha256()
    for th, ph in zip(g["e2p_cfg2_thetas"], g["e2p_cfg2_phis"]):
        lon, lat = PP.e2p_maps(90.0, float(th), float(ph), 256, 256, 512, 1024)
        h.update(lon.tobytes())
        h.update(lat.tobytes())
    assert np.array_equal(np.frombuffer(h.digest(), np.uint8), g["e2p_cfg2_sha256"])
    for n, (th, ph) in enumerate([(0.0, 0.0), (0.0, 17.5), (30.0, -40.0)]):
        lon, lat, mask = PP.p2e_maps(90.0, th, ph, 24, 40, 48, 96)
        assert np.array_equal(lon, g[f"p2e_lon_{n}"]) and np.array_equal(lat, g[f"p2e_lat_{n}"]) and np.array_equal(mask, g[f"p2e_mask_{n}"])
    assert np.array_equal(PP.cubic_weight_table(), OPP.cubic_weight_table())
    for n in range(6):
        assert PP.get_maxrec_cord(g[f"rect_mask_{n}"]) == tuple(int(v) for v in g["rects"][n])
    _, _, m = PP.p2e_maps(90.0, 0.0, 12.0, 256, 256, 256, 512)
    assert PP.get_maxrec_cord(torch.from_numpy(m)) == tuple(int(v) for v in g["rect_p2e_phi12"])


def test_dropin_preprocess_aliases_are_opt_in():
    """install(preprocess=True) adds the N3 names (E2P / P2E classes, get_anchor_target, get_maxrec_cord); the default
    overlay leaves the script's preprocessing on the checkout (its cv2 arithmetic cannot be checked here)."""
    from imagine360_amd import dropin, preprocess
    extra = dropin._preprocess_aliases()
    assert extra["src.utils.pano_utils.Equirec2Perspec"]["Equirectangular"] is preprocess.Equirectangular
    assert extra["src.utils.pano_utils.Perspec2Equirec"]["Perspective"] is preprocess.Perspective
    assert extra["animatediff.utils.video_mask"]["get_anchor_target"] is preprocess.get_anchor_target
    assert extra["src.modules.utils"]["get_maxrec_cord"] is preprocess.get_maxrec_cord
    assert not any(k.startswith("src.utils.pano_utils") or k == "animatediff.utils.video_mask" for k in dropin._ALIASES)


def test_measurement_tools_parse_what_they_claim(tmp_path):
    """The two text tools the round-4 evidence rests on, on tiny synthetic inputs: tools/kernel_event_map.py (instruction-order map
    of a kernel from device assembly: the check that no scratch reload sits inside a K loop) and tools/trace_by_shape.py (a
    rocprofv3 kernel trace grouped by kernel and grid, tail of N steps)."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("kernel_event_map", os.path.join(root, "tools", "kernel_event_map.py"))
    kem = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kem)
    body = ["\tglobal_load_lds_dwordx4 v[0:1], off", "\ts_waitcnt vmcnt(16)", "\ts_barrier", "\tds_read_b128 v[4:7], v2", "\tds_read_b128 v[8:11], v2",
            "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[4:7], v[8:11], v[0:15]", "\tscratch_load_dwordx2 v[2:3], off, off", "\ts_waitcnt vmcnt(0)",
            "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[4:7], v[8:11], v[0:15]", "\tglobal_store_dwordx4 v[0:1], v[4:7], off"]
    assert kem.event_map(body) == "G v16  | d2 M L v0  M W"
    # trace_by_shape: two steps, each ending with two cfg_ddim launches; the tail of ONE step must hold only the second step's rows
    hdr = "Kind,Agent_Id,Queue_Id,Stream_Id,Thread_Id,Dispatch_Id,Kernel_Id,Kernel_Name,Correlation_Id,Start_Timestamp,End_Timestamp,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Workgroup_Size_X,Workgroup_Size_Y,Workgroup_Size_Z,Grid_Size_X,Grid_Size_Y,Grid_Size_Z"
    rows, t = [], 0

    def launch(name, dur, grid, wg):
        nonlocal t
        rows.append(f"KERNEL_DISPATCH,1,1,1,1,{len(rows)},1,\"{name}\",1,{t},{t + dur},0,0,64,0,32,{wg},1,1,{grid * wg},1,1")
        t += dur + 10

    for step in range(2):
        launch("_ZN5im36016conv_ring_kernelIDF16bLi5ELi2ELb1ELb1ELi3ELb0ELi2EEEvNS_10ConvParamsE", 1000 * (step + 1), 256, 512)
        launch("_ZN5im36017gn_apply_kernelIDF16bEEvPKT_PKfS5_PS1_iiiiiiii", 100, 2560, 256)
        launch("_ZN5im36015cfg_ddim_kernelIDF16bEEvPKT_S3_S3_PS1_lfffPKf", 5, 64, 256)
        launch("_ZN5im36015cfg_ddim_kernelIDF16bEEvPKT_S3_S3_PS1_lfffPKf", 5, 64, 256)
    f = tmp_path / "t_kernel_trace.csv"
    f.write_text(hdr + "\n" + "\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "trace_by_shape.py"), str(f), "--steps", "1"], capture_output=True, text=True, check=True).stdout
    assert "# 4 dispatches over 1 step(s)" in out
    ring = [ln for ln in out.splitlines() if "conv_ring_kernel" in ln and "grid=" in ln]
    assert len(ring) == 1 and ring[0].split()[0] == "0.002" and "grid=(256, 1, 1) x 512" in ring[0], ring


def test_caller_supplied_outputs_drop_a_producer_tag():
    """the ctypes kernels do not bump Tensor._version, so every wrapper that writes into a caller-supplied tensor
    drops a GroupNorm producer tag the tensor may carry (kernels._written)."""
    import inspect
    from imagine360_amd import kernels as K
    t = torch.zeros(4)
    K._tag_gn(t, torch.zeros(1), 1)
    assert K._gn_of(t) is not None
    assert K._written(t) is t and K._gn_of(t) is None and K._written(None) is None
    for fn in (K.attention, K.temporal_attention, K.softmax_rows, K.shard_pack):
        assert "_written(" in inspect.getsource(fn), fn.__name__


def test_tool_patches_apply_to_the_tree():
    """tools/patches/* (the cycle-stamp instrumentation of the ring kernel, tools/stamps_probe.py) must apply to HEAD."""
    import glob
    import shutil
    import subprocess
    import tempfile
    patches = sorted(glob.glob(os.path.join(ROOT, "tools", "patches", "*.patch")))
    assert patches
    tmp = tempfile.mkdtemp()
    try:
        # (works without a .git directory: the patched file is copied into a scratch tree and `patch --dry-run` is asked)
        for p in patches:
            text = open(p).read()
            files = re.findall(r"^\+\+\+ b/(\S+)", text, re.M)
            assert files, p
            for f in files:
                os.makedirs(os.path.dirname(os.path.join(tmp, f)), exist_ok=True)
                shutil.copy(os.path.join(ROOT, f), os.path.join(tmp, f))
            r = subprocess.run(["patch", "-p1", "--dry-run", "-F0", "-i", p], cwd=tmp, capture_output=True, text=True)
            assert r.returncode == 0, (p, r.stdout[-400:], r.stderr[-400:])
    finally:
        shutil.rmtree(tmp)


def test_newest_rocprof_summary_names_kernels_the_shipped_library_contains():
    """profiles/ must describe the code that ships.  Every im360 kernel symbol in the kernel-statistics summaries
    of the newest round (tools/profile_bench.sh keeps the names mangled) has to exist in the gfx950 code objects of
    imagine360_amd/libim360_kernels.so -- a summary taken before a kernel's template arguments changed fails here."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    tools = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(f"{tools}/llvm-objdump"):
        pytest.skip("ROCm LLVM binutils not installed")
    files = glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats.csv"))
    rounds = {f: int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)) for f in files}
    newest = max(rounds.values())
    assert newest >= 5, "no rocprofv3 summary of this round's library under profiles/ (tools/profile_bench.sh)"
    tmp = tempfile.mkdtemp()
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(os.path.join(ROOT, "imagine360_amd", "libim360_kernels.so"), so)
        subprocess.run([f"{tools}/llvm-objdump", "--offloading", so], capture_output=True, cwd=tmp, check=True)
        have = set()
        for f in glob.glob(so + ".*gfx950"):
            syms = subprocess.run([f"{tools}/llvm-objdump", "-t", f], capture_output=True, text=True, check=True).stdout
            have.update(re.findall(r"\b(_ZN5im360\w+)", syms))
    finally:
        shutil.rmtree(tmp)
    assert len(have) > 100, len(have)
    checked = 0
    for f in sorted(k for k, v in rounds.items() if v == newest):
        lines = [l for l in open(f) if not l.startswith("#")]
        names = [r["Name"].split(".kd")[0] for r in csv.DictReader(lines)]
        ours = [n for n in names if n.startswith("_ZN5im360")]
        assert len(ours) >= 20, (f, "kernel names must be mangled (rocprofv3 -M) and cover the step", len(ours))
        missing = [n for n in ours if n not in have]
        assert not missing, (os.path.basename(f), missing[:5])
        checked += len(ours)
    assert checked >= 20
