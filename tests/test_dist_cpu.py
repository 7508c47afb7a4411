"""Multi-process paths on CPU (gloo, world_size 2): sample-parallel latent gather and the frame-chunk
sharding of the motion modules (all-to-all frame-sharded <-> pixel-sharded tokens)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

torch.set_grad_enabled(False)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    return dict(ret)


def _gather_job(rank, world):
    from imagine360_amd import dist as D
    lat = torch.full((1, 4, 2, 3, 5), float(rank + 1))
    g = D.gather_latents(lat)
    assert D.shard_samples(range(5)) == list(range(5))[rank::world]
    return [float(g[i].mean()) for i in range(world)] + [list(g.shape)]


def test_sample_parallel_latent_gather():
    out = _run(_gather_job)
    for r in range(2):
        assert out[r][:2] == [1.0, 2.0] and out[r][2] == [2, 1, 4, 2, 3, 5]


def _exchange_job(rank, world):
    import _emu_kernels as E
    from imagine360_amd.dist import FrameShard
    b, f, p, c = 2, 6, 7, 8                      # 7 pixels: exercises the zero-padded last pixel shard
    full = torch.arange(b * f * p * c, dtype=torch.float32).reshape(b, f, p, c)
    sh = FrameShard(f)
    loc = sh.take(full, 1).contiguous()
    with E.patched_kernels():
        pp = sh.pixels_per_rank(p)
        px = sh.frames_to_pixels(loc).reshape(f, b, pp, c)          # rows (frame, batch, pixel): this rank's pixels of ALL frames
        lo = rank * pp
        want = torch.zeros(b, f, pp, c)
        n = max(0, min(p, lo + pp) - lo)
        want[:, :, :n] = full[:, :, lo:lo + n]
        ok1 = torch.equal(px.permute(1, 0, 2, 3), want)
        buf = sh.pixel_result_buffer(loc, b, p, c)                  # the return trip's send buffer, same row order
        buf.copy_(px.reshape(-1, c))
        back = sh.pixels_to_frames(buf, b, p)
    ok2 = torch.equal(back, loc)
    ok3 = torch.equal(sh.gather_frames(loc, 1), full)
    return bool(ok1 and ok2 and ok3)


def test_frame_pixel_all_to_all_round_trip():
    out = _run(_exchange_job)
    assert out[0] and out[1]


def _motion_job(rank, world, boundary="attention"):
    import _emu_kernels as E
    from imagine360_amd.dist import FrameShard
    from imagine360_amd.layers import to_cl
    from imagine360_amd.unet3d import TemporalTransformer3DModel, VanillaTemporalModule, VersatileAttention
    from imagine360_amd.weights import fill_module_
    mm = VanillaTemporalModule(in_channels=64, num_attention_heads=8, num_transformer_block=1,
                               temporal_position_encoding=True, temporal_position_encoding_max_len=64)
    fill_module_(mm)
    g = torch.Generator().manual_seed(3)
    x5 = torch.randn(2, 64, 8, 3, 5, generator=g)                 # b c f h w, 8 frames, 15 pixels (ragged last pixel shard)
    sent = []
    with E.patched_kernels():
        full = mm(x5)
        sh = FrameShard(8, boundary=boundary)
        orig = sh.exchange
        sh.exchange = lambda send, tag: (sent.append(send.numel()), orig(send, tag))[1]
        holder = TemporalTransformer3DModel if boundary == "module" else VersatileAttention
        for mod in mm.modules():
            if isinstance(mod, holder):
                mod.frame_shard = sh
        xl, fl = to_cl(sh.take(x5, 2).contiguous())
        from imagine360_amd.layers import from_cl
        loc = from_cl(mm.forward_cl(xl, fl), fl)
    want = sh.take(full, 2)
    return [float((loc - want).abs().max() / want.abs().max()), len(sent), sum(sent)]


def _motion_job_module(rank, world):
    return _motion_job(rank, world, "module")


@pytest.mark.parametrize("boundary", ["attention", "module"])
def test_frame_sharded_motion_module_matches_unsharded(boundary):
    """One motion module, frames cut over two ranks: the exchange around every attention (round 3) and at the module
    boundary (round 5: GroupNorm frame-sharded -> one C-wide all-to-all -> proj_in ... proj_out on pixel-sharded rows of all
    frames -> one C-wide all-to-all -> residual add; animatediff/models/motion_module.py:158-185) both reproduce the
    unsharded module; the module boundary moves a quarter of the bytes in a quarter of the collectives."""
    out = _run(_motion_job if boundary == "attention" else _motion_job_module)
    b, fl, pp, c = 2, 4, 8, 64                                   # 15 pixels -> 8 per rank
    for r in range(2):
        err, n_exch, elems = out[r]
        assert err < 1e-5, out[r]
        if boundary == "module":
            assert n_exch == 2 and elems == 2 * (2 * fl * b * pp * c), out[r]              # C out, C back
        else:
            assert n_exch == 4 and elems == 2 * (2 * fl * b * pp * (3 * c + c)), out[r]    # (3 C out + C back) per attention, two attentions


def _mv_sharded_job(rank, world, boundary="module"):
    """Whole dual-branch forward (both UNets, 7 WarpAttn, all motion modules) with the frames cut over the ranks ==
    this rank's frames of the unsharded forward.  Reduced width (channels / 10), 4 frames, 256 x 512 panorama."""
    import random
    import _emu_kernels as E
    from imagine360_amd import configs, synthetic as S
    from imagine360_amd.dist import FrameShard, shard_mv_inputs
    mv = configs.build_mv_model(10, device="cpu", dtype=torch.float32, xformers=True)
    mv.noise_on_host = True
    frames = 4
    inp = S.mv_inputs(frames=frames, pano_hw=(32, 64), pers_hw=(16, 16), seed=5, sam_frames=16)
    cams = S.icosahedron_cameras(90, 128)
    kw = dict(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True)
    with E.patched_kernels():
        torch.manual_seed(3)
        random.seed(3)
        pers_full, pano_full = mv(**kw, **inp)
        sh = FrameShard(frames, boundary=boundary)
        n_exch = []
        orig = sh.exchange
        sh.exchange = lambda send, tag: (n_exch.append(tag), orig(send, tag))[1]
        mv.set_frame_shard(sh)
        from imagine360_amd.unet3d import TemporalTransformer3DModel, VersatileAttention
        holders = [type(m).__name__ for m in mv.modules() if isinstance(m, (TemporalTransformer3DModel, VersatileAttention)) and m.frame_shard is sh]
        torch.manual_seed(3)              # every rank replays the unsharded run's RNG stream (IP noise, WarpAttn coins)
        random.seed(3)
        pers_loc, pano_loc = mv(**kw, **shard_mv_inputs(inp, sh))
        mv.set_frame_shard(None)
        gathered = sh.gather_frames(pano_loc.contiguous(), 2)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    return [rel(pano_loc, sh.take(pano_full, 2)), rel(pers_loc, sh.take(pers_full, 3)), rel(gathered, pano_full),
            list(pano_loc.shape), list(pers_loc.shape), len(n_exch), sorted(set(holders))]


def _mv_sharded_job_attention(rank, world):
    return _mv_sharded_job(rank, world, "attention")


def _mv_two_communicators_job(rank, world):
    """The panorama UNet's all-to-alls on a second process group (dist.frame_shard_pair; what `dual_stream_shard` needs on a GPU):
    same result as with one communicator, and the motion modules of the two UNets really hold different groups."""
    import random
    import _emu_kernels as E
    from imagine360_amd import configs, synthetic as S
    from imagine360_amd.dist import FrameShard, frame_shard_pair, shard_mv_inputs
    from imagine360_amd.unet3d import TemporalTransformer3DModel as Holder
    mv = configs.build_mv_model(10, device="cpu", dtype=torch.float32, xformers=True)
    mv.noise_on_host = True
    frames = 4
    inp = S.mv_inputs(frames=frames, pano_hw=(32, 64), pers_hw=(16, 16), seed=5, sam_frames=16)
    cams = S.icosahedron_cameras(90, 128)
    kw = dict(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True)
    outs = []
    with E.patched_kernels():
        sh1 = FrameShard(frames)
        sh, psh = frame_shard_pair(frames)
        for shards in ((sh1, None), (sh, psh)):
            mv.set_frame_shard(*shards)
            torch.manual_seed(3)
            random.seed(3)
            outs.append(mv(**kw, **shard_mv_inputs(inp, shards[0])))
            two = mv._shard_two_comms
            groups = ({id(m.frame_shard.group) for m in mv.pano_unet.modules() if isinstance(m, Holder)},
                      {id(m.frame_shard.group) for m in mv.unet.modules() if isinstance(m, Holder)})
            mv.set_frame_shard(None)
    same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    return [same, two, len(groups[0]) == 1 and len(groups[1]) == 1 and groups[0] != groups[1], mv._sharded, mv._shard_two_comms]


def test_panorama_branch_on_its_own_communicator_matches_one_communicator():
    res = _run(_mv_two_communicators_job)
    assert len(res) == 2
    for same, two, distinct, still_sharded, still_two in res.values():
        assert same and two and distinct and not still_sharded and not still_two


@pytest.mark.parametrize("boundary", ["module", "attention"])
def test_frame_sharded_mv_forward_matches_unsharded(boundary):
    """The model runs 32 motion modules per step (16 per UNet: the DownBlock3D / UpBlock3D ones are skipped like in the
    reference, src/models/MVGenModel.py:292-303, 426-443): 64 all-to-alls with the exchange at the module boundary (the
    default), 128 around the attentions."""
    out = _run(_mv_sharded_job if boundary == "module" else _mv_sharded_job_attention)
    for r in range(2):
        assert out[r][0] < 1e-5 and out[r][1] < 1e-5 and out[r][2] < 1e-5, out[r]
        assert out[r][3] == [2, 4, 2, 32, 64] and out[r][4] == [2, 20, 4, 2, 16, 16]
        assert out[r][5] == (64 if boundary == "module" else 128), out[r]
        assert out[r][6] == (["TemporalTransformer3DModel"] if boundary == "module" else ["VersatileAttention"]), out[r]


def _cfg_split_job(rank, world):
    """BASELINE config 5 layout at world size 2: rank 0 runs the unconditional CFG half, rank 1 the text half (frame
    shards of one rank each); after the pairwise exchange both hold the CFG-batched prediction of the unsplit model."""
    import random
    import _emu_kernels as E
    from imagine360_amd import configs, synthetic as S
    from imagine360_amd.dist import cfg_frame_layout, cfg_half_inputs, exchange_cfg_halves, shard_mv_inputs
    mv = configs.build_mv_model(10, device="cpu", dtype=torch.float32, xformers=True)
    mv.noise_on_host = True
    inp = S.mv_inputs(frames=2, pano_hw=(32, 64), pers_hw=(16, 16), seed=6, sam_frames=16)
    cams = S.icosahedron_cameras(90, 128)
    kw = dict(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True)
    half, shard, pair = cfg_frame_layout(2)
    assert half == rank and shard.world == 1 and shard.local == 2
    with E.patched_kernels():
        torch.manual_seed(4)
        random.seed(4)
        pers_full, pano_full = mv(**kw, **inp)
        mv.set_frame_shard(shard)
        torch.manual_seed(4)
        random.seed(4)
        # the IP-adapter noise is drawn for the CFG batch: draw the full batch's stream on both ranks, keep this half
        mv._ip_noise_half = (half, 2)
        pers_h, pano_h = mv(**kw, **shard_mv_inputs(cfg_half_inputs(inp, half), shard))
        mv._ip_noise_half = None
        mv.set_frame_shard(None)
        pano_both, pers_both = exchange_cfg_halves(pano_h, pair), exchange_cfg_halves(pers_h, pair)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    return [rel(pano_both, pano_full), rel(pers_both, pers_full), list(pano_h.shape)]


def test_cfg_halves_on_two_rank_groups_match_the_cfg_batched_forward():
    out = _run(_cfg_split_job)
    for r in range(2):
        assert out[r][0] < 1e-5 and out[r][1] < 1e-5, out[r]
        assert out[r][2] == [1, 4, 2, 32, 64]


def _pipeline_sharded_job(rank, world):
    """The whole pipeline call (noise, masked-latent VAE encode of the LOCAL frames only, 2 DDIM steps with the motion
    modules' all-to-alls, decode, frame gather) frame-sharded over 2 ranks == the unsharded call from the same seeds; and a
    call that raises inside the loop leaves the model unsharded."""
    import random
    import _emu_kernels as E
    from imagine360_amd import configs, synthetic as S
    from imagine360_amd.dist import FrameShard
    from imagine360_amd.pipeline import AnimationPipeline
    from imagine360_amd.scheduler import DDIMScheduler
    from imagine360_amd.unet3d import TemporalTransformer3DModel, VersatileAttention
    holders = (TemporalTransformer3DModel, VersatileAttention)
    mv = configs.build_mv_model(10, device="cpu", dtype=torch.float32, xformers=True, motion_heads=4)
    vae = configs.build_vae(4, device="cpu", dtype=torch.float32)
    pipe = AnimationPipeline(vae, None, None, mv.unet, mv.pano_unet, mv, DDIMScheduler(**configs.NOISE_SCHEDULER_KWARGS), None, "SAM")
    pipe.rng, pipe._no_progress, pipe.use_graph = "host", True, False
    frames = 4
    vb = S.video_batch(frames=frames, pano_hw=(128, 256), seed=3)
    cond = S.conditioning(frames=16, seed=3)
    kw = dict(num_inference_steps=2, guidance_scale_text=7.5, negative_prompt="", latents_dtype=torch.float32, video_batch=vb,
              use_outpaint=True, use_ip_plus_cross_attention=True, use_fps_condition=True, ip_plus_condition="video",
              prompt_embeds=(cond["text_pano"], cond["text_pers"]), sam_features=(cond["sam_pano"], cond["sam_pers"]))
    with E.patched_kernels():
        torch.manual_seed(9)
        random.seed(9)
        full = pipe("synthetic", **kw).videos
        full_lat = pipe.last_latents[0].clone()
        encoded = []
        orig = vae.encode
        vae.encode = lambda x, *a, **k: (encoded.append(x.shape[0]), orig(x, *a, **k))[1]
        torch.manual_seed(9)
        random.seed(9)
        sh = FrameShard(frames)
        part = pipe("synthetic", frame_shard=sh, **kw).videos
        vae.encode = orig
        part_lat = pipe.last_latents[0]
        unsharded_after = all(m.frame_shard is None for m in mv.modules() if isinstance(m, holders))
        raised = False
        try:
            pipe("synthetic", frame_shard=sh, **dict(kw, callback=lambda *a: 1 / 0))
        except ZeroDivisionError:
            raised = True
        clean_after_error = all(m.frame_shard is None for m in mv.modules() if isinstance(m, holders))
    rel = lambda a, b: float((a - b).norm() / b.norm())
    return [rel(part, full), rel(part_lat, full_lat), sum(encoded), unsharded_after, raised and clean_after_error]


def test_frame_sharded_pipeline_call_matches_unsharded():
    out = _run(_pipeline_sharded_job)
    for r in range(2):
        assert out[r][0] < 1e-5 and out[r][1] < 1e-5, out[r]
        assert out[r][2] == 4 + 4 * 20 // 2, out[r]         # images through the VAE encoder: one 8-image chunk holds all 4 panorama frames; half of the 80 views
        assert out[r][3] and out[r][4], out[r]


# ------------------------------------------------------------------ bench.py's own multi-rank plumbing
def _bench_main(rank, world, mode, extra=()):
    import sys
    import _emu_kernels as E
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    os.environ["LOCAL_RANK"] = str(rank)
    with E.patched_kernels():
        return bench.main(["--gpus", str(world), "--backend", "gloo", "--steps", "1", "--warmup", "1", "--workload", "cfg1",
                           "--width-div", "10", "--parallelism", mode, *extra])


def _bench_samples(rank, world):
    return _bench_main(rank, world, "samples")


def _bench_frames(rank, world):
    # (with the opt-in second communicator for the panorama UNet: the flag's plumbing in bench.py; cfgxframes covers the default)
    return _bench_main(rank, world, "frames", ("--dual-stream-shard", "1"))


def _bench_cfgxframes(rank, world):
    return _bench_main(rank, world, "cfgxframes")


@pytest.mark.parametrize("mode", ["samples", "frames", "cfgxframes"])
def test_bench_script_runs_its_multi_rank_path_on_gloo(mode):
    """`bench.py --gpus 2 --backend gloo`: the benchmark's OWN rank bookkeeping, input sharding, per-step exchanges, latent
    gather, max-over-ranks timing and JSON line, on CPU tensors with the emulated kernels (a plumbing check, flagged invalid as
    a measurement) -- so that the first run on a multi-GPU node does not also debut this code."""
    out = _run({"samples": _bench_samples, "frames": _bench_frames, "cfgxframes": _bench_cfgxframes}[mode])
    line = out[0]
    assert out[1] is None and line is not None
    assert line["plumbing_check"] is True and line["valid"] is False and line["n_gpus"] == 2
    assert line["scaling"] == ("weak" if mode == "samples" else "strong") and line["steps"] == 1
    assert line["value"] > 0 and line["config"]["outputs_finite"] is True
    want = {"samples": "sample-parallel x2", "frames": "frame-chunk sharding x2 (4 frames per GPU)", "cfgxframes": "CFG halves x frame chunks (2 x 1)"}[mode]
    assert line["config"]["parallelism"] == want


def test_bench_script_respawn_command_line(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run on 127.0.0.1."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import argparse
    import bench
    seen = {}
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench._respawn(argparse.Namespace(gpus=4))
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
