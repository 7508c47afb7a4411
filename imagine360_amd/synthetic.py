"""Synthetic, seeded inputs of the dual-branch denoising step and the
icosahedron camera rig the reference builds in ``get_cameras`` (inference_dual_p2e.py:79-110,
src/utils/pano.py:37-72, 103-119).  Used by bench.py, smoke(), tests and the golden generator."""
import math

import numpy as np
import torch


def icosahedron_angles():
    """Yaw/pitch (radians) of the 20 face centres of a regular icosahedron, in the reference's
    order: 5 top faces, 5 upper-middle, 5 lower-middle, 5 bottom (src/utils/pano.py:37-72)."""
    r_circ = math.sin(2 * math.pi / 5.0)
    r_in = math.sqrt(3) / 12.0 * (3 + math.sqrt(5))
    r_mid = math.cos(math.pi / 5.0)
    step = 2.0 * math.pi / 5.0
    top = math.pi / 2 - math.acos(r_in / r_circ)
    upper = top - 2 * math.acos(r_in / r_mid)
    thetas, phis = [], []
    for ring, (lat, off) in enumerate(((top, step / 2), (upper, step / 2), (-upper, 0.0), (-top, 0.0))):
        for k in range(5):
            thetas.append(-math.pi + off + k * step)
            phis.append(lat)
    return np.array(thetas), np.array(phis)


def _axis_angle(v):
    v = np.asarray(v, np.float64)
    th = np.linalg.norm(v)
    if th < 1e-15:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K


def icosahedron_cameras(fov=90, pers_resolution=256, device="cpu"):
    """Camera dict with the reference's keys/shapes: height,width,FoV,theta,phi [1,20] (degrees),
    R,K [1,20,3,3]."""
    th, ph = icosahedron_angles()
    th, ph = np.rad2deg(th), np.rad2deg(ph)
    Ks, Rs = [], []
    for t, p in zip(th, ph):
        f = 0.5 * pers_resolution / math.tan(0.5 * fov / 180.0 * math.pi)
        c = (pers_resolution - 1) / 2.0
        Ks.append(np.array([[f, 0, c], [0, f, c], [0, 0, 1]], np.float32))
        R1 = _axis_angle(np.array([0.0, 1.0, 0.0]) * np.radians(t))
        R2 = _axis_angle(R1 @ np.array([1.0, 0.0, 0.0]) * np.radians(p))
        Rs.append((R2 @ R1).astype(np.float32))
    full = lambda v: torch.from_numpy(np.full_like(th, v, dtype=int)).unsqueeze(0).to(device)
    return {
        "height": full(pers_resolution), "width": full(pers_resolution), "FoV": full(fov),
        "theta": torch.from_numpy(th).unsqueeze(0).to(device),
        "phi": torch.from_numpy(ph).unsqueeze(0).to(device),
        "R": torch.from_numpy(np.stack(Rs)).unsqueeze(0).to(device),
        "K": torch.from_numpy(np.stack(Ks)).unsqueeze(0).to(device),
    }


def mv_inputs(frames=16, pano_hw=(32, 64), pers_hw=(16, 16), views=20, seed=0, sam_frames=None,
              timestep=961, text_dim=1024, dtype=torch.float32, device="cpu"):
    """Inputs of one CFG-batched ``MultiViewBaseModel.forward`` call (b=2), drawn from a CPU
    generator so every backend sees identical numbers."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    b, m, f = 2, views, frames
    sf = sam_frames or max(16, f)
    H, W = pano_hw
    h, w = pers_hw

    def with_mask(lat, mask, masked):
        return torch.cat([lat, mask, masked], dim=-4)

    pano = with_mask(rn(b, 4, f, H, W), (torch.rand(b, 1, f, H, W, generator=g) < 0.7).float(), rn(b, 4, f, H, W))
    pers = with_mask(rn(b, m, 4, f, h, w), (torch.rand(b, m, 1, f, h, w, generator=g) < 0.7).float(),
                     rn(b, m, 4, f, h, w))
    feat_pano = rn(b, sf, 4096, 256)
    feat_pers = rn(b, 1, sf, 4096, 256).expand(b, m, sf, 4096, 256)
    rel = torch.tensor([1.0, 1.0, 255.0, 255.0, float(H * 8), float(W * 8)]).repeat(b, f, 1)
    rel = rel + torch.randint(0, 8, (b, f, 6), generator=g).float()
    pitch = (torch.rand(b, f, generator=g) * 60.0 - 30.0)
    out = dict(
        latents=pers, pano_latent=pano, timestep=torch.tensor([timestep], dtype=torch.int64),
        prompt_embd=rn(b * m, 77, text_dim), pano_prompt_embd=rn(b, 77, text_dim),
        fps_tensor_pano=torch.full((b,), 8.0), fps_tensor_pers=torch.full((b, m), 8.0),
        reference_images_clip_feat_pano=feat_pano, reference_images_clip_feat_pers=feat_pers,
        relative_position_tensor=rel, pitchs_tensor=pitch)
    for k, v in out.items():
        if torch.is_floating_point(v) and k not in FP32_INPUTS:
            v = v.to(dtype)
        out[k] = v.to(device)
    return out


# Scalar conditioning (frame rate, crop rectangle, camera pitch in degrees): the pipeline hands these to the model as
# float32 whatever the latent dtype is (pipeline.py, like pipeline_animation_inference_dual.py:608-613) -- a pitch rounded to
# bf16 is a different camera (17.3 -> 17.25 degrees shifts its sinusoidal embedding by 5 %), not a rounding of the same one.
FP32_INPUTS = ("fps_tensor_pano", "fps_tensor_pers", "relative_position_tensor", "pitchs_tensor")


def cast_mv_inputs(inp, device, dtype):
    """``mv_inputs`` tensors -> device, the activations in ``dtype``, the scalar conditioning left in float32."""
    return {k: (v.to(device, dtype) if torch.is_floating_point(v) and k not in FP32_INPUTS else v.to(device)) for k, v in inp.items()}


def video_batch(frames=16, pano_hw=(256, 512), seed=0, fps=8, anchor_hw=(64, 64)):
    """Synthetic ``video_batch`` dict with the keys/shapes the reference pipeline consumes
    (inference_dual_p2e.py:546-564; pipeline_animation_inference_dual.py:605-620)."""
    g = torch.Generator().manual_seed(seed)
    H, W = pano_hw
    ps = H // 2
    f, m = frames, 20
    pano_pix = torch.rand(1, f, 3, H, W, generator=g) * 2 - 1
    pano_mask = torch.ones(1, f, 1, H, W)
    pano_mask[..., H // 4: 3 * H // 4, W // 3: 2 * W // 3] = 0.0        # known (outpainting anchor) region
    pers_pix = torch.rand(1, f, m, 3, ps, ps, generator=g) * 2 - 1
    pers_masks = (torch.rand(1, f, m, 1, ps // 8, ps // 8, generator=g) < 0.7).float()
    pers_masks = pers_masks.repeat_interleave(8, dim=-1).repeat_interleave(8, dim=-2)
    rel = torch.tensor([1.0, 1.0, 255.0, 255.0, float(H), float(W)]).repeat(f, 1)
    rel = rel + torch.randint(0, 8, (f, 6), generator=g).float()
    return {
        "videoid": "synthetic", "fps": fps, "video_length": f,
        "pano_pixel_values": pano_pix, "pano_mask": pano_mask,
        "pers_pixel_values": pers_pix, "pers_masks": pers_masks,
        "anchor_pixels_values": torch.rand(1, f, 3, *anchor_hw, generator=g) * 2 - 1,
        "anchor_pixels_values_pers": torch.rand(1, f, 3, *anchor_hw, generator=g) * 2 - 1,
        "relative_position": rel, "pitchs": torch.rand(f, generator=g) * 60.0 - 30.0,
        "cameras": icosahedron_cameras(90, ps), "pano_H": H, "pano_W": W, "pers_size": ps,
    }


def conditioning(frames=16, views=20, seed=0, text_dim=1024):
    """Stand-ins for the conditioning producers that are out of scope (CLIP text encoder, SAM):
    CFG-stacked text embeddings [2,77,d] / [2*views,77,d] and SAM features [1,F,4096,256] x2."""
    g = torch.Generator().manual_seed(seed + 1000)
    text_pano = torch.randn(2, 77, text_dim, generator=g)
    text_pers = text_pano.repeat_interleave(views, dim=0) + 0.01 * torch.randn(2 * views, 77, text_dim, generator=g)
    sam = torch.randn(2, frames, 4096, 256, generator=g)
    return dict(text_pano=text_pano, text_pers=text_pers, sam_pano=sam[0:1], sam_pers=sam[1:2])
