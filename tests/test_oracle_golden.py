"""The CPU oracle (oracle/im360_oracle) against the fixtures generated from the REAL reference
(oracle/tools/gen_golden.py, run in the authoring container with /root/reference imported).
Tolerance: fp32 re-association only -> relative L2 <= 2e-5 (masks are stored as fp16: 1e-3)."""
import random

import pytest
import torch

from helpers import gold, op_inputs, rel
from im360_oracle import ddim as OD, geometry as OG, mv as OMV, pipeline as OP, unet as OU, vae as OV
from im360_oracle.cfg import sd21_unet_cfg, sd21_vae_cfg
from imagine360_amd import synthetic as S
from imagine360_amd.weights import fill_state_dict_

torch.set_grad_enabled(False)
TOL = 2e-5


@pytest.fixture(scope="module")
def mv_sd():
    from imagine360_amd import configs
    mv = configs.build_mv_model(5, device="cpu", dtype=torch.float32, xformers=False)
    return dict(mv.state_dict())


@pytest.fixture(scope="module")
def vae_sd():
    from imagine360_amd import configs
    return dict(configs.build_vae(4, device="cpu", dtype=torch.float32).state_dict())


def test_ops_against_reference(mv_sd):
    g, I, sd, P = gold("ops_w5.npz"), op_inputs(), mv_sd, "pano_unet."
    pad, unpad = OG.pad_pano, OG.unpad_pano
    assert rel(unpad(OU.resnet_block(sd, P + "down_blocks.0.resnets.0.", pad(I["x"], 2), I["emb"]), 2), g["resnet_pano"]) < TOL
    x2 = torch.cat([I["x"], I["x"].flip(1), 0.5 * I["x"].roll(3, 1)], 1)
    assert rel(OU.resnet_block(sd, P + "up_blocks.3.resnets.0.", x2, I["emb"]), g["resnet_shortcut"]) < TOL
    assert rel(OU.spatial_transformer(sd, P + "down_blocks.0.attentions.0.", I["x"], I["ctx"], 1, 64), g["spatial_cpu"]) < TOL
    assert rel(OU.spatial_transformer(sd, P + "down_blocks.0.attentions.0.", I["x"], I["ctx"], 1, 64, xformers=True), g["spatial_xf"]) < TOL
    assert rel(OU.motion_module(sd, P + "down_blocks.0.motion_modules.0.", I["x"]), g["motion"]) < TOL
    assert rel(unpad(OU.downsample(sd, P + "down_blocks.0.downsamplers.0.", pad(I["x"], 2)), 1), g["down_pano"]) < TOL
    assert rel(unpad(OU.upsample(sd, P + "up_blocks.0.upsamplers.0.", pad(I["x3"], 1)), 2), g["up_pano"]) < TOL
    assert rel(unpad(OU.conv2d_frames(sd, P + "conv_in.", pad(I["lat9"], 1)), 1), g["conv_in_pano"]) < TOL
    assert rel(OMV.ip_tokens_clean(sd, P, sd21_unet_cfg(5), I["feat"]), g["ip_tokens"]) < 1e-4
    cams = {k: v[0] for k, v in S.icosahedron_cameras(90, 64).items()}
    for tag in ("normal", "oppo"):
        p, e = OMV.warp_attn(sd, "cp_blocks_encoder.0.", I["px"], I["ex"], cams, opposite=(tag == "oppo"))
        assert rel(p, g["warp_pers_" + tag]) < TOL and rel(e, g["warp_equi_" + tag]) < TOL


def test_masks_coords_pe_against_reference():
    g = gold("masks.npz")
    for ph, eh in ((4, 8), (8, 16)):
        cams = {k: v[0] for k, v in S.icosahedron_cameras(90, ph * 8).items()}
        for tag in ("normal", "oppo"):
            pm, em = OG.merged_masks(ph, ph, eh, 2 * eh, cams, tag == "oppo")
            assert (pm - g[f"pers_{tag}_{ph}"]).abs().max() < 1e-3 and (em - g[f"equi_{tag}_{ph}"]).abs().max() < 1e-3
            assert pm.min() >= -1 and pm.max() <= 1
        pc, ec = OG.coords(ph, ph, eh, 2 * eh, cams)
        assert torch.equal(pc, g[f"pers_coords_{ph}"]) and torch.equal(ec, g[f"equi_coords_{ph}"])
    _, ec = OG.coords(4, 4, 8, 16, {k: v[0] for k, v in S.icosahedron_cameras(90, 32).items()})
    for nf in (16, 80, 160):
        assert torch.equal(OG.spherical_pe(ec, nf)[::3, ::5], g[f"equi_pe_{nf}"])


def test_masks_at_the_cfg5_level_1_size_against_reference():
    """The oracle's one-hot construction at equirect 64 x 128 / 20 views of 32 x 32 (the largest mask of BASELINE cfg5) against the
    REAL get_merged_masks' fixture; the antipodal variant here (gen_golden.py masks5 asserted both variants equal to the
    reference, max abs 0, while writing the fixture; one of them keeps this test at ~20 s and 6 GB of host memory)."""
    from helpers import check_masks_64x128x32
    cams = {k: v[0] for k, v in S.icosahedron_cameras(90, 512).items()}
    pm, em = OG.merged_masks(32, 32, 64, 128, cams, True)
    e2p = pm.reshape(20, 8192, 1024).permute(1, 0, 2).reshape(8192, 20480)
    del pm
    check_masks_64x128x32("oppo", e2p, em.reshape(20480, 8192))


def test_ddim_against_reference():
    g = gold("ddim.npz")
    acp = OD.alphas_cumprod()
    assert torch.equal(acp, g["alphas_cumprod"])
    gen = torch.Generator().manual_seed(5)
    x, v = torch.randn(1, 4, 2, 8, 16, generator=gen), torch.randn(1, 4, 2, 8, 16, generator=gen)
    for n in (4, 25, 50):
        ts = OD.timesteps(n)
        assert torch.equal(ts, g[f"timesteps_{n}"].long())
        for idx in (0, n - 1):
            assert rel(OD.step_v(v, ts[idx], x, acp, n), g[f"step_{n}_{idx}"]) < 1e-6


def test_vae_against_reference(vae_sd):
    g, cfg = gold("vae_w4.npz"), sd21_vae_cfg(4)
    gen = torch.Generator().manual_seed(9)
    x = torch.rand(2, 3, 64, 96, generator=gen) * 2 - 1
    z = torch.randn(2, 4, 8, 20, generator=gen)
    assert rel(OV.encode_moments(vae_sd, cfg, x), g["moments"]) < TOL
    assert rel(OV.decode(vae_sd, cfg, z), g["decoded"]) < TOL


@pytest.mark.parametrize("xf,name", [(False, "mv_forward_w5.npz"), (True, "mv_forward_w5_xf.npz")])
def test_mv_forward_against_reference(mv_sd, xf, name):
    g = gold(name)
    cfg = sd21_unet_cfg(5)
    cfg.xformers = xf
    inp = S.mv_inputs(frames=8, pano_hw=(32, 64), pers_hw=(16, 16), seed=0, sam_frames=16)
    cams = S.icosahedron_cameras(90, 128)
    torch.manual_seed(7)
    random.seed(7)
    taps = {}
    pers, pano = OMV.mv_forward(mv_sd, cfg, inp["latents"], inp["pano_latent"], inp["timestep"], inp["prompt_embd"],
                                inp["pano_prompt_embd"], cams, inp["fps_tensor_pano"], inp["fps_tensor_pers"],
                                inp["reference_images_clip_feat_pano"], inp["reference_images_clip_feat_pers"],
                                inp["relative_position_tensor"], inp["pitchs_tensor"], taps=taps, mask_cache={})
    assert rel(pano, g["pano"]) < TOL and rel(pers[:, [0, 7, 13, 19]], g["pers_views"]) < TOL
    if not xf:
        for n, (tp, te) in taps.items():
            assert rel(te[:, ::4, ::3], g[f"tap_{n}_equi"]) < TOL


def test_pipeline_against_reference(mv_sd, vae_sd):
    g = gold("pipeline_w5.npz")
    vb = S.video_batch(frames=16, pano_hw=(256, 512), seed=0)
    cond = S.conditioning(frames=16, seed=0)
    torch.manual_seed(21)
    random.seed(21)
    trace = []
    vid, _, _ = OP.run(mv_sd, sd21_unet_cfg(5), vae_sd, sd21_vae_cfg(4), vb, cond["text_pano"], cond["text_pers"],
                       cond["sam_pano"], cond["sam_pers"], num_inference_steps=2, trace=trace)
    for i, t in enumerate(trace):
        assert rel(t, g[f"pano_latent_{i}"]) < 1e-4
    assert rel(vid[:, :, ::3, ::4, ::4], g["video_sub"]) < 1e-3          # fixture stored as fp16
    assert vid.min() >= 0 and vid.max() <= 1 and vid.shape == (1, 3, 16, 256, 512)


def test_sr_close_loop_pad_against_reference():
    """the oracle's padding_pano / unpadding_pano / circular_pad == the real src/utils/pano.py pad_pano /
    unpad_pano as sr/video_to_video_model.py:16-29, 99, 160-162 calls them (tests/golden/sr_pad.npz), bit-exact."""
    import numpy as np
    import os
    from helpers import GOLDEN
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "sr_pad.npz")).items()}     # keep fp16 / fp32 as stored
    assert torch.equal(OG.padding_pano(g["lat"], 16, latent=True), g["lat_pad16"])
    assert torch.equal(OG.padding_pano(g["vid"], 16, latent=False), g["vid_pad128"])
    assert torch.equal(OG.unpadding_pano(g["vid_pad128"], 16, latent=False), g["vid_unpad128"])
    assert torch.equal(OG.circular_pad(g["fr"], (3, 5, 2, 4)), g["fr_fit"])
    assert OG.padding_pano(g["lat"], 0, latent=True) is g["lat"]
    with pytest.raises(NotImplementedError):
        OG.padding_pano(g["lat"][0, 0, 0], 16, latent=True)


def test_remap_restatement_tracks_an_independent_bicubic():
    """cv2 is not in this image, so the oracle's restatement of cv2.remap(INTER_CUBIC, BORDER_WRAP) (and with it the HIP
    kernel, which the GPU tier holds to it bit for bit) cannot be pinned on OpenCV itself.  It CAN be held against an
    independent implementation of the same published algorithm that IS here: torch's bicubic grid_sample (Keys kernel,
    A = -0.75, float arithmetic) on a circularly padded image.  At coordinates that are exact multiples of 1/32 pixel -- where
    OpenCV's coordinate quantisation is exact, leaving only its 2^-15 weight table and the final rounding -- the two agree to
    1 LSB everywhere and exactly on > 99 % of the samples, wrap-around taps in both axes included: kernel, tap alignment,
    border mode and rounding are the published ones.  At arbitrary coordinates on a smooth image the 1/32-pixel quantisation
    costs at most 2 LSB more."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from im360_oracle import preprocess as P
    rng = np.random.default_rng(0)
    H, W, h, w, pad = 40, 64, 50, 70, 4

    def torch_bicubic(img, mx, my):
        t = F.pad(torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None], (pad, pad, pad, pad), mode="circular")
        gx = (torch.from_numpy(mx.astype(np.float64)) + pad) / (W + 2 * pad - 1) * 2 - 1
        gy = (torch.from_numpy(my.astype(np.float64)) + pad) / (H + 2 * pad - 1) * 2 - 1
        out = F.grid_sample(t, torch.stack([gx, gy], -1)[None], mode="bicubic", padding_mode="border", align_corners=True)
        return out[0].permute(1, 2, 0).numpy()

    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)                          # white noise: the hardest image
    mx = (rng.integers(-48, (W + 1) * 32, (h, w)) / 32.0).astype(np.float32)       # from -1.5 to W + 1: wraps on both sides
    my = (rng.integers(-48, (H + 1) * 32, (h, w)) / 32.0).astype(np.float32)
    d = np.abs(P.remap_cubic_wrap_u8(img, mx, my).astype(np.int64) - np.clip(np.rint(torch_bicubic(img, mx, my)), 0, 255).astype(np.int64))
    assert d.max() <= 1 and (d == 0).mean() > 0.99, (d.max(), (d == 0).mean())
    yy, xx = np.mgrid[0:H, 0:W]
    smooth = np.stack([127.5 + 120 * np.sin(2 * np.pi * (xx / W + c * yy / H)) * np.cos(2 * np.pi * yy / H) for c in (1, 2, 3)], -1)
    smooth = np.clip(np.rint(smooth), 0, 255).astype(np.uint8)                     # periodic in both axes, like a panorama in W
    fx = rng.uniform(-1.5, W + 1, (h, w)).astype(np.float32)
    fy = rng.uniform(-1.5, H + 1, (h, w)).astype(np.float32)
    d = np.abs(P.remap_cubic_wrap_u8(smooth, fx, fy).astype(np.int64) - np.clip(np.rint(torch_bicubic(smooth, fx, fy)), 0, 255).astype(np.int64))
    assert d.max() <= 3, d.max()


def test_preprocessing_geometry_against_reference():
    """the oracle's E2P / P2E sampling maps and masks == the maps the REAL Equirec2Perspec / Perspec2Equirec
    modules hand to cv2.remap, and its get_maxrec_cord == the real one (tests/golden/preproc.npz), bit-exact.  (cv2.remap's
    own arithmetic is parity-unpinned: see the oracle header.)"""
    import hashlib
    import numpy as np
    import os
    from helpers import GOLDEN
    from im360_oracle import preprocess as OPP
    g = np.load(os.path.join(GOLDEN, "preproc.npz"))
    for n, (th, ph) in enumerate([(0.0, 0.0), (36.0, 52.6), (-108.0, -10.8), (180.0, 90.0), (72.0, -52.6)]):
        lon, lat = OPP.e2p_maps(90, th, ph, 32, 32, 64, 128)
        assert np.array_equal(lon, g[f"e2p_lon_{n}"]) and np.array_equal(lat, g[f"e2p_lat_{n}"])
    h = hashlib.sha256()
    for th, ph in zip(g["e2p_cfg2_thetas"], g["e2p_cfg2_phis"]):
        lon, lat = OPP.e2p_maps(90, th, ph, 256, 256, 512, 1024)
        h.update(lon.tobytes())
        h.update(lat.tobytes())
    assert np.array_equal(np.frombuffer(h.digest(), np.uint8), g["e2p_cfg2_sha256"])
    for n, (th, ph) in enumerate([(0.0, 0.0), (0.0, 17.5), (30.0, -40.0)]):
        lon, lat, mask = OPP.p2e_maps(90, th, ph, 24, 40, 48, 96)
        assert np.array_equal(lon, g[f"p2e_lon_{n}"]) and np.array_equal(lat, g[f"p2e_lat_{n}"]) and np.array_equal(mask, g[f"p2e_mask_{n}"])
    for n in range(6):
        assert OPP.get_maxrec_cord(g[f"rect_mask_{n}"]) == tuple(int(v) for v in g["rects"][n])
    tab = OPP.cubic_weight_table()
    assert tab.shape == (1024, 16) and (tab.astype(np.int64).sum(1) == 32768).all()
    assert tab[0].reshape(4, 4)[1, 1] == 32767 and tab[0].reshape(4, 4)[2, 2] == 1      # integer position: saturated centre tap + the residue
    img = np.random.default_rng(0).integers(0, 256, (9, 11, 3), dtype=np.uint8)
    yy, xx = np.meshgrid(np.arange(9, dtype=np.float32), np.arange(11, dtype=np.float32), indexing="ij")
    assert np.array_equal(OPP.remap_cubic_wrap_u8(img, xx, yy), img)                    # identity map returns the image
