"""Expected xGMI traffic of the frame-sharded modes: bytes every rank SENDS per
denoising step through FrameShard.frames_to_pixels / pixels_to_frames (imagine360_amd/dist.py; the reference's full-length
temporal attention, animatediff/models/motion_module.py:165-183, 250-331), so that the first SCALE line measured on a multi-GPU
node can be checked against them.  Two exchange placements:
  module    (default since round 5) one C-wide all-to-all behind the module's GroupNorm, one C-wide in front of its residual add
            (the whole temporal transformer runs on pixel-sharded rows: unet3d.TemporalTransformer3DModel._forward_pixel_sharded)
  attention (round 3 / 4) 3 C out + C back around each of the module's two attentions

    python tools/xgmi_bytes.py            # cfg4 on 4 GPUs (frames), cfg5 on 8 GPUs (cfg x frames = 2 x 4)
"""
LEVEL_C = (320, 640, 1280, 1280)
MODULES = (5, 5, 5, 1)          # motion modules the dual model RUNS per level and branch: 2 (down) + 3 (up) at levels 0 - 2; at the lowest
                                # resolution only the mid block's (DownBlock3D / UpBlock3D modules are skipped, src/models/MVGenModel.py:292-303, 426-443)
ATTN_PER_MODULE = 2             # attention_block_types = (Temporal_Self, Temporal_Self)
LINK_GBS = 153.0                # one xGMI link, per direction (MI355X platform: 7 links per GPU, fully connected 8-GPU node)


def per_step(frames, pano_hw, pers_hw, cfg_batch, world, views=20, elem=2, boundary="module"):
    fl = frames // world
    rows = []
    tot = 0.0
    for branch, (h, w), images in (("pano", pano_hw, cfg_batch), ("pers", pers_hw, cfg_batch * views)):
        for lvl, (c, nmod) in enumerate(zip(LEVEL_C, MODULES)):
            p = (h >> lvl) * (w >> lvl)
            tokens = images * fl * p                              # this rank's frame-sharded tokens
            there = tokens * (3 if boundary == "attention" else 1) * c * elem * (world - 1) / world   # q | k | v rows (attention) / normalised rows (module) to the ranks owning the other pixel ranges
            back = tokens * c * elem * (world - 1) / world        # attention output / proj_out rows back to the frame owners
            n = nmod * (ATTN_PER_MODULE if boundary == "attention" else 1)
            rows.append((branch, lvl, c, p, n, (there + back) / 1e6, n * (there + back) / 1e9))
            tot += n * (there + back)
    return rows, tot


def report(name, frames, pano_hw, pers_hw, cfg_batch, world, compute_ms_one_gpu, boundary="module"):
    rows, tot = per_step(frames, pano_hw, pers_hw, cfg_batch, world, boundary=boundary)
    print(f"== {name} [exchange at the {boundary} boundary]: {frames} frames over {world} ranks ({frames // world} per rank), CFG batch {cfg_batch} per rank")
    print("   branch level  chan  pixels  exchanges/step   MB sent per exchange pair   GB sent per step")
    for b, lvl, c, p, n, mb, gb in rows:
        print(f"   {b:5s}  L{lvl}   {c:5d} {p:7d}  {n:5d}            {mb:10.1f}               {gb:8.2f}")
    links = world - 1
    t_ms = tot / (links * LINK_GBS * 1e9) * 1e3
    print(f"   total sent per rank and step: {tot / 1e9:.1f} GB, spread point-to-point over {links} direct links: >= {t_ms:.0f} ms at {LINK_GBS:.0f} GB/s per link"
          f" (not overlapped with compute: the exchange sits " + ("between the GroupNorm and proj_in / between proj_out and the residual add)" if boundary == "module" else "between the QKV GEMM and the attention kernel)"))
    print(f"   compute per rank ~ {compute_ms_one_gpu / world:.0f} ms (single-GPU step {compute_ms_one_gpu:.0f} ms / {world}) -> expected step >= {compute_ms_one_gpu / world + t_ms:.0f} ms, "
          f"speed-up over one GPU <= {compute_ms_one_gpu / (compute_ms_one_gpu / world + t_ms):.2f}x of {world}x")
    print(f"   latent boundary: all-gather of {frames // world} frames of the panorama latent = {4 * (frames // world) * pano_hw[0] * pano_hw[1] * 2 / 1e6:.2f} MB per rank, once\n")


if __name__ == "__main__":
    # single-GPU step times of these shapes: profiles/r04_bench_other_configs.json (cfg4 897 ms, cfg5 1545 ms)
    for bd in ("module", "attention"):
        report("cfg4, --parallelism frames, 4 GPUs", 48, (64, 128), (32, 32), 2, 4, 897.0, bd)
        report("cfg5, --parallelism cfgxframes, 8 GPUs = 2 CFG halves x 4 frame shards", 16, (128, 256), (64, 64), 1, 4, 1545.0 / 2, bd)
    print("cfg5 adds one pairwise exchange of the two predictions per step (exchange_cfg_halves): "
          f"{(4 * 4 * 128 * 256 + 20 * 4 * 4 * 64 * 64) * 2 / 1e6:.1f} MB each way per rank pair")
    print("cfg3 (--parallelism samples): no per-step traffic; one all-gather of the 1 MB panorama latents at the end")
