#!/usr/bin/env python
"""Benchmark of the dual-branch denoising hot path (BASELINE.json metric: denoising steps/sec).

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: re-executes itself under
                                                              torch.distributed.run, one rank per GPU over RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one CFG-batched MultiViewBaseModel.forward (both UNets + 7 WarpAttn) + CFG combine + the two
DDIM updates, on synthetic inputs already resident in HBM.

  --parallelism samples (default, BASELINE config 3): one independent sample per GPU, weights replicated, no per-step
      collective, one all-gather of the final panorama latents inside the timed region.  Weak scaling.
  --parallelism frames (BASELINE config 4, and 5 at N <= frames): ONE sample, contiguous frame chunks per GPU; every
      motion-module attention exchanges tokens with one all-to-all each way (imagine360_amd.dist.FrameShard), the
      latents are all-gathered along the frame axis at the end.  Strong scaling.
  --parallelism cfgxframes (BASELINE config 5 on 8 GPUs): the two CFG halves on two rank groups, frames sharded inside
      each group; per step the halves exchange their predictions pairwise for the CFG combine.  Strong scaling.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from imagine360_amd import configs, flops, kernels, synthetic  # noqa: E402
from imagine360_amd.scheduler import DDIMScheduler  # noqa: E402

# committed measurements this script quotes (one place for the round tag): the PMC traffic summary and the direct CPU step
HBM_TRAFFIC_JSON = "r06_hbm_traffic.json"          # tools/hbm_traffic.sh on the shipped kernels (non-temporal epilogue stores included)
CPU_BASELINE_DIRECT = "r06_cpu_baseline_cfg2_direct.json"      # tools/cpu_baseline.py --cfg2-threads 32
MFMA_PEAK_TFLOPS = 2500.0       # dense bf16/fp16 MFMA peak of MI355X (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0           # HBM3E peak (MI355X_MICROARCH.md; ~6300 GB/s is what a copy kernel reaches)
WORKLOADS = {
    "cfg2": dict(frames=16, pano_hw=(64, 128), pers_hw=(32, 32), pers_px=256,
                 desc="BASELINE cfg2: 16-frame 512x1024 equirect (pano latent 4x16x64x128 + 20 views 4x16x32x32), "
                      "CFG batch 2, full-width random-init UNets, DDIM step"),
    "cfg1": dict(frames=8, pano_hw=(32, 64), pers_hw=(16, 16), pers_px=128,
                 desc="BASELINE cfg1 shapes: 8-frame 256x512 equirect, CFG batch 2"),
    "cfg4": dict(frames=48, pano_hw=(64, 128), pers_hw=(32, 32), pers_px=256,
                 desc="BASELINE cfg4 shapes: 48-frame 512x1024 equirect, CFG batch 2 (frame-chunk sharding)"),
    "cfg5": dict(frames=16, pano_hw=(128, 256), pers_hw=(64, 64), pers_px=512,
                 desc="BASELINE cfg5 shapes: 16-frame 1024x2048 equirect, CFG batch 2"),
}


def _best_threads():
    """Thread count that runs a miniature of the oracle step fastest on this host: a 3x3 conv, GroupNorm, a 1024-token
    attention and a token-major Linear at level-0 sizes (oversubscribing a 256-thread box makes the small per-frame ops
    of this path 3x slower; a plain GEMM probe once picked 64 threads where the real step was 45 % slower than on 32)."""
    import torch.nn.functional as F
    cores = os.cpu_count() or 1
    best, best_t = cores, float("inf")
    x, w = torch.randn(16, 320, 32, 32), torch.randn(320, 320, 3, 3)
    q, tok, wl = torch.randn(16, 5, 1024, 64), torch.randn(16384, 320), torch.randn(1280, 320)

    def mini():
        h = F.conv2d(F.silu(F.group_norm(x, 32)), w, padding=1)
        o = F.scaled_dot_product_attention(q, q, q)
        return h.sum() + o.sum() + F.linear(tok, wl).sum()

    for nt in sorted({min(cores, c) for c in (16, 32, 64, 128, cores)}):
        torch.set_num_threads(nt)
        mini()
        t0 = time.time()
        for _ in range(3):
            mini()
        t = time.time() - t0
        if t < best_t:
            best, best_t = nt, t
    torch.set_num_threads(best)
    return best


def cpu_baseline_step(mv, args, workload="cfg1", threads="probe"):
    """The oracle (CPU restatement of the reference path, fp32) timed on ONE FULL-WIDTH denoising step on the host cores of
    this box, with the GPU model's weights (``mv``; None: the same filler weights built on the CPU).
    ``workload`` "cfg1" (default: 8 frames of 256x512, the reference's own CPU-runnable case, 31.9 TFLOP, ~1 min): the figure
    for the benchmarked workload is the measured cfg1 rate scaled by the analytic FLOP ratio, stated separately; "cfg2": the
    benchmarked workload itself, measured directly (~7 minutes of host time -- `--cpu-baseline
    direct` / tools/cpu_baseline.py, run once per round and committed under profiles/).
    ``threads``: "probe" (the count a miniature of the step runs fastest on), "all" (os.cpu_count()) or a number."""
    import random
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from im360_oracle import mv as OMV
    from im360_oracle.cfg import sd21_unet_cfg
    prev_threads = torch.get_num_threads()        # restored before returning
    if threads == "probe":
        nthreads = _best_threads()
    else:
        nthreads = (os.cpu_count() or 1) if threads == "all" else int(threads)
        torch.set_num_threads(nthreads)
    w1 = WORKLOADS[workload]
    cfg = sd21_unet_cfg(args.width_div)
    cfg.xformers = True
    if mv is None:
        mv = configs.build_mv_model(args.width_div, device="cpu", dtype=torch.float32, xformers=True)
    sd = {k: v.detach().float().cpu() for k, v in mv.state_dict().items()}
    inp = synthetic.mv_inputs(frames=w1["frames"], pano_hw=w1["pano_hw"], pers_hw=w1["pers_hw"], seed=1, sam_frames=16)
    cams = synthetic.icosahedron_cameras(90, w1["pers_px"])
    torch.manual_seed(0)
    random.seed(0)
    # the step-invariant work the GPU path hoists out of the loop (IP-adapter conditioning, cross-view masks) is timed
    # inside the oracle step so that it can be reported separately
    from im360_oracle import geometry as OGm
    hoisted = {"ip_adapter_conditioning_s": 0.0, "cross_view_masks_s": 0.0}

    def timed(fn, key):
        def wrapper(*a, **k):
            t = time.time()
            try:
                return fn(*a, **k)
            finally:
                hoisted[key] += time.time() - t
        return wrapper

    orig_ip, orig_masks = OMV.ip_tokens_clean, OGm.merged_masks
    OMV.ip_tokens_clean, OGm.merged_masks = timed(orig_ip, "ip_adapter_conditioning_s"), timed(orig_masks, "cross_view_masks_s")
    t0 = time.time()
    try:
        with torch.no_grad():
            o_pers, o_pano = OMV.mv_forward(sd, cfg, inp["latents"], inp["pano_latent"], inp["timestep"], inp["prompt_embd"],
                                            inp["pano_prompt_embd"], cams, inp["fps_tensor_pano"], inp["fps_tensor_pers"],
                                            inp["reference_images_clip_feat_pano"], inp["reference_images_clip_feat_pers"],
                                            inp["relative_position_tensor"], inp["pitchs_tensor"], mask_cache={})
    finally:
        OMV.ip_tokens_clean, OGm.merged_masks = orig_ip, orig_masks
    dt = time.time() - t0
    torch.set_num_threads(prev_threads)
    assert torch.isfinite(o_pano).all()
    dt_loop = dt - sum(hoisted.values())
    boc = tuple(mv.unet.config.block_out_channels)
    f1 = flops.step_flops(frames=w1["frames"], pano_hw=w1["pano_hw"], pers_hw=w1["pers_hw"], block_out_channels=boc)
    w = WORKLOADS[args.workload]
    fw = flops.step_flops(frames=w["frames"], pano_hw=w["pano_hw"], pers_hw=w["pers_hw"], block_out_channels=boc)
    direct = workload == args.workload
    cpu_model = ""
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"value": (1.0 / dt) * f1 / fw, "unit": "denoising steps/sec", "cores": nthreads, "host_cores": os.cpu_count(), "cpu_model": cpu_model,
            "threads_chosen_by": threads, "kind": "port",
            "hoisted_work_inside_the_measured_step": dict(hoisted, per_step_work_s=dt_loop,
                                                        value_counting_only_per_step_work=(1.0 / dt_loop) * f1 / fw),
            "sample": f"ONE full-width oracle step of BASELINE {workload} ({w1['frames']} frames, {w1['pano_hw'][0] * 8}x{w1['pano_hw'][1] * 8} equirect, CFG batch 2, "
                      f"{f1 / 1e12:.1f} TFLOP incl. the IP-adapter conditioning and mask building the GPU path hoists): "
                      f"measured {dt:.1f} s on {nthreads} threads of the box's {os.cpu_count()} logical cores = {f1 / dt / 1e12:.3f} TFLOP/s",
            f"measured_{workload}_s_per_step": dt, f"measured_{workload}_steps_per_s": 1.0 / dt,
            "extrapolation": ("none: the benchmarked workload itself was timed" if direct else
                              f"value = measured {workload} steps/s x ({f1 / 1e12:.1f} / {fw / 1e12:.1f}) analytic FLOP ratio to {args.workload}")}


def cpu_baseline_sample(args):
    """Bounded alternative (--cpu-baseline sample): the oracle's ResnetBlock3D + spatial transformer + motion module at
    FULL width at UNet levels 0 and 2 on 8 frames of 12 views (~10 s), scaled by the analytic FLOP ratio."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from im360_oracle import unet as OU
    from imagine360_amd.mv_model import MultiViewBaseModel
    from imagine360_amd.weights import filler_tensor
    nthreads = _best_threads()
    with torch.device("meta"):
        meta = MultiViewBaseModel(configs.build_unet(1), configs.build_unet(1)).state_dict()
    frames, ctx_n, views = 8, 141, 12
    samples = [("pano_unet.down_blocks.0.", 320, 5, 32, 32, "resnets.0."), ("pano_unet.down_blocks.2.", 1280, 20, 8, 8, "resnets.1.")]
    tot_flops, tot_time, parts = 0.0, 0.0, []
    with torch.no_grad():
        for pre, c, heads, h, w, res in samples:
            sd = {k: filler_tensor(k, v.shape) for k, v in meta.items() if k.startswith(pre)}
            g = torch.Generator().manual_seed(0)
            x = torch.randn(views, c, frames, h, w, generator=g)
            emb = torch.randn(views, 1280, generator=g)
            ctx = torch.randn(views, ctx_n, 1024, generator=g)
            n = h * w * views
            fl = 2 * (2.0 * c * c * 9 * n * frames)
            fl += frames * n * (2.0 * c * c * 2 + 2.0 * c * c * 4 + 2.0 * c * c * 2 + 2.0 * c * 8 * c + 2.0 * 4 * c * c)
            fl += views * frames * ctx_n * 2.0 * 1024 * c * 2 + 4.0 * n * ctx_n * c * frames + 4.0 * n * (h * w) * c * frames
            fl += n * frames * (2.0 * c * c * 2 + 2 * (2.0 * c * c * 4) + 2.0 * c * 8 * c + 2.0 * 4 * c * c) + 2 * 4.0 * frames * frames * c * n
            t0 = time.time()
            y = OU.resnet_block(sd, pre + res, x, emb)
            y = OU.spatial_transformer(sd, pre + "attentions.0.", y, ctx, heads, 64, xformers=True)
            y = OU.motion_module(sd, pre + "motion_modules.0.", y)
            dt = time.time() - t0
            assert torch.isfinite(y).all()
            tot_flops += fl
            tot_time += dt
            parts.append(f"{c} ch {h}x{w}: {fl / 1e9:.0f} GF in {dt:.2f} s")
    w = WORKLOADS[args.workload]
    full_tf = flops.step_flops(frames=w["frames"], pano_hw=w["pano_hw"], pers_hw=w["pers_hw"]) / 1e12
    cpu_tflops = tot_flops / tot_time / 1e12
    return {"value": cpu_tflops / full_tf, "unit": "denoising steps/sec", "cores": nthreads, "host_cores": os.cpu_count(), "kind": "port",
            "sample": "oracle ResnetBlock3D + spatial transformer + motion module, full width, fp32, 8 frames of 12 views ("
                      + "; ".join(parts) + f") = {cpu_tflops:.3f} TFLOP/s on {nthreads} host threads; scaled to the "
                      f"{full_tf:.1f} TF step of {args.workload} by the analytic FLOP ratio"}


def _respawn(args):
    """`python bench.py --gpus N` without a launcher: run N ranks under torch.distributed.run (RCCL over xGMI)."""
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--parallelism", default="samples", choices=["samples", "frames", "cfgxframes"])
    ap.add_argument("--width-div", type=int, default=1, help="debug only: reduced-width model (INVALID as a benchmark)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--cpu-baseline", default="step", choices=["step", "sample", "none", "direct"],
                    help="step: one full-width cfg1 oracle step on the host cores (~1 min) scaled to the workload by the FLOP ratio; "
                         "direct: one oracle step of the benchmarked workload (--workload) itself (~9 min at cfg2); sample: bounded block sample (~10 s)")
    ap.add_argument("--cpu-threads", default="probe", help="host threads of the CPU baseline: probe (fastest on a miniature of the step), all (os.cpu_count()), or a number")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tuned-gemms", action="store_true", help="hipBLASLt default heuristic instead of the shipped solution table")
    ap.add_argument("--no-graph", action="store_true", help="issue every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--warp-streams", type=int, default=1, help="with --dual-stream: the two directions of every WarpAttn on the two streams too")
    ap.add_argument("--dual-stream", type=int, default=1, help="1 (default): the panorama branch between WarpAttn calls on a side stream (two parallel branches in the hipGraph); 0: one stream")
    ap.add_argument("--dual-stream-shard", type=int, default=0,
                    help="with --parallelism frames: 1 = a second communicator for the panorama UNet's all-to-alls so that the panorama "
                         "branch keeps its side stream under the shard (opt-in: never measured on more than one GPU)")
    ap.add_argument("--shard-boundary", default="module", choices=["module", "attention"],
                    help="frame modes: where the motion modules exchange tokens -- module (default: one C-wide all-to-all behind the "
                         "module's GroupNorm and one in front of its residual add, 2 C per token and module) or attention (round 3: "
                         "3 C out + C back around every temporal attention, 8 C per module)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: PLUMBING CHECK on CPU tensors (tests/test_dist_cpu.py): the rank bookkeeping, sharding, collectives and the JSON "
                         "line of this script with whatever `imagine360_amd.kernels` the caller installed; never a benchmark number")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", HBM_TRAFFIC_JSON),
                    help="rocprofv3 PMC summary (tools/hbm_traffic.sh) the roofline block quotes HBM traffic from")
    args = ap.parse_args(argv)
    if args.no_cpu_baseline:
        args.cpu_baseline = "none"
    plumbing = args.backend == "gloo"
    if plumbing:
        args.cpu_baseline, args.no_graph, args.no_tuned_gemms = "none", True, True

    if not plumbing and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _respawn(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} launched with WORLD_SIZE {world}: the two must agree")
    if plumbing:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    dist, own_group, out = None, False, None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        own_group = not dist.is_initialized()
        if own_group:                          # (the CPU-tier test's harness has a gloo group up already)
            if plumbing:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    dt = torch.float32 if plumbing else (torch.bfloat16 if args.dtype == "bf16" else torch.float16)
    w = WORKLOADS[args.workload]
    frames = w["frames"]
    mode = args.parallelism if world > 1 else "samples"
    torch.set_grad_enabled(False)
    if not plumbing:
        kernels.lib()
    tuned = False
    from imagine360_amd import tuning
    if not args.no_tuned_gemms:
        tuned = tuning.enable()
    mv = configs.build_mv_model(args.width_div, device=dev, dtype=dt, xformers=True)
    mv.dual_stream = bool(args.dual_stream)
    mv.warp_streams = bool(args.warp_streams)
    # samples: every rank its own sample (seed); frame modes: ONE sample, replicated conditioning, identical RNG streams
    seed = 1 + rank if mode == "samples" else 1
    inp = synthetic.mv_inputs(frames=frames, pano_hw=w["pano_hw"], pers_hw=w["pers_hw"], seed=seed,
                              sam_frames=max(16, frames), dtype=dt, device=dev)
    cams = synthetic.icosahedron_cameras(90, w["pers_px"], device=dev)
    shard, pair = None, None
    if mode != "samples":
        import random
        from imagine360_amd.dist import FrameShard, exchange_cfg_halves, shard_mv_inputs
        torch.manual_seed(1234)
        random.seed(1234)
        pano_shard = None
        if mode == "frames":
            if args.dual_stream_shard:
                from imagine360_amd.dist import frame_shard_pair
                shard, pano_shard = frame_shard_pair(frames, boundary=args.shard_boundary)
                mv.dual_stream_shard = True
            else:
                shard = FrameShard(frames, boundary=args.shard_boundary)
        else:
            from imagine360_amd.dist import cfg_frame_layout, cfg_half_inputs
            my_half, shard, pair = cfg_frame_layout(frames, boundary=args.shard_boundary)
            inp = cfg_half_inputs(inp, my_half)
            mv._ip_noise_half = (my_half, 2)
        inp = shard_mv_inputs(inp, shard)
        mv.set_frame_shard(shard, pano_shard)
    sch = DDIMScheduler(**configs.NOISE_SCHEDULER_KWARGS)
    nsteps_total = 25
    sch.set_timesteps(nsteps_total)
    ts_host = sch._timesteps_host
    ts_dev = [torch.tensor([t], dtype=torch.int64, device=dev) for t in ts_host]
    # CFG halves see the same latent (pipeline_animation_inference_dual.py:750-751); channels 0..3 are the noisy latent
    pano_in = inp["pano_latent"]
    pers_in = inp["latents"]
    pano_lat = pano_in[:1, :4].contiguous()
    pers_lat = pers_in[:1, :, :4].contiguous()
    guidance = 7.5

    def eager_step(i):
        nonlocal pano_lat, pers_lat
        t = i % nsteps_total
        pano_in[:, :4] = pano_lat
        pers_in[:, :, :4] = pers_lat
        pred_pers, pred_pano = mv(
            latents=pers_in, pano_latent=pano_in, timestep=ts_dev[t], prompt_embd=inp["prompt_embd"],
            pano_prompt_embd=inp["pano_prompt_embd"], cameras=cams, use_fps_condition=True,
            use_ip_plus_cross_attention=True, fps_tensor_pano=inp["fps_tensor_pano"],
            fps_tensor_pers=inp["fps_tensor_pers"],
            reference_images_clip_feat_pano=inp["reference_images_clip_feat_pano"],
            reference_images_clip_feat_pers=inp["reference_images_clip_feat_pers"],
            relative_position_tensor=inp["relative_position_tensor"], pitchs_tensor=inp["pitchs_tensor"])
        if pair is not None:
            # CFG combine across the two halves: each rank receives its partner's prediction of the same frames
            pred_pano, pred_pers = exchange_cfg_halves(pred_pano, pair), exchange_cfg_halves(pred_pers, pair)
        pano_lat = sch.fused_cfg_step(pred_pano[0:1], pred_pano[1:2], guidance, ts_host[t], pano_lat)
        pers_lat = sch.fused_cfg_step(pred_pers[0:1], pred_pers[1:2], guidance, ts_host[t], pers_lat)

    step = eager_step

    def barrier():
        if dist is not None:
            dist.barrier()
        if not plumbing:
            torch.cuda.synchronize()

    # (attn_warp / attn_x2: the biased WarpAttn launches and the two-set text + IP cross attention, timed apart and merged into "attn" below)
    prof_kinds = ["conv", "gemm", "attn", "attn_warp", "attn_x2", "temporal", "gn_stats", "gn_apply", "misc"]
    graphed = None
    if not args.no_graph and (shard is None or os.environ.get("IM360_GRAPH_SHARDED", "1") != "0"):
        # the step is ~3000 launches (+ 128 all-to-alls in the frame modes): capture it once (hipGraph) and replay, so the
        # host is out of the loop; the frame-sharded exchanges are RCCL stream operations on pre-sized buffers
        from imagine360_amd.graph_step import GraphedDenoiseStep
        try:
            graphed = GraphedDenoiseStep(mv, sch, inp, cams, pano_lat, pers_lat, guidance, warmup=1, cfg_pair=pair)
        except RuntimeError as e:           # capture refused (e.g. by another runtime thread): time the eager issue instead
            print(f"[rank {rank}] hipGraph capture failed, falling back to eager launches: {e}", file=sys.stderr)
            graphed = None
            torch.cuda.synchronize()
    if graphed is not None:
        def step(i):                                                       # noqa: F811
            graphed.step(ts_host[i % nsteps_total])

    for i in range(args.warmup):
        step(i)
    if dist is not None:          # untimed: first use of the boundary collective (RCCL channel setup is lazy)
        cur = pano_lat if graphed is None else graphed.pano_lat
        if shard is not None:
            shard.gather_frames(cur, 2)
        else:
            warm = [torch.empty_like(cur) for _ in range(world)]
            dist.all_gather(warm, cur)
            del warm
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if graphed is not None:
        pano_lat = graphed.pano_lat
    if dist is not None:          # latent boundary: gather every rank's panorama latent (1 MB each) / frame chunk
        if shard is not None:
            full_lat = shard.gather_frames(pano_lat, 2)
        else:
            gathered = [torch.empty_like(pano_lat) for _ in range(world)]
            dist.all_gather(gathered, pano_lat)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # per-kernel-class durations: the same K steps once more, issued eagerly with HIP events around every launch
    # of our kernels (a graph replay has no per-launch host hook); kernel durations do not depend on how they were launched
    eager_elapsed = None
    if rank == 0 and shard is None and not plumbing:
        if graphed is not None:
            pano_lat, pers_lat = graphed.pano_lat.clone(), graphed.pers_lat.clone()

            def step(i):                                                   # noqa: F811
                eager_step(i)
        kernels.prof_enable(prof_kinds)
        kernels.STATS, kernels.SHAPES = {}, {}
        was_dual, mv.dual_stream = mv.dual_stream, False      # one stream: a kernel's events then bracket that kernel alone
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        torch.cuda.synchronize()
        eager_elapsed = time.perf_counter() - t1
        mv.dual_stream = was_dual
        kernels.prof_enable([])
    finite = bool(torch.isfinite(pano_lat.float()).all().item())
    # parity of the BENCHMARKED launch mode at the benchmarked size (outside the timed region, one step): from identical
    # latents, RNG states and timestep, the captured two-stream hipGraph against the same step issued eagerly on one stream
    parity_check = None
    if rank == 0 and shard is None and graphed is not None:
        import random
        saved_counters = (kernels.STATS, kernels.SHAPES)          # (the profile pass's per-launch flop / byte counters: not these steps')
        kernels.STATS = kernels.SHAPES = None
        p0, q0 = graphed.pano_lat.clone(), graphed.pers_lat.clone()
        py_state, dev_state = random.getstate(), torch.cuda.get_rng_state(dev)
        i_chk = args.warmup + args.steps
        graphed.step(ts_host[i_chk % nsteps_total])
        torch.cuda.synchronize()
        g_pano, g_pers = graphed.pano_lat.clone(), graphed.pers_lat.clone()
        res = {}
        for name, dual in (("eager_one_stream", False), ("eager_two_streams", True)):
            random.setstate(py_state)
            torch.cuda.set_rng_state(dev_state, dev)
            pano_lat, pers_lat = p0.clone(), q0.clone()
            was_dual, mv.dual_stream, mv.dual_stream_eager = mv.dual_stream, dual, dual
            eager_step(i_chk)
            torch.cuda.synchronize()
            mv.dual_stream, mv.dual_stream_eager = was_dual, False
            res[name] = (pano_lat.clone(), pers_lat.clone())
        relf = lambda a, b: float(((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item())
        e_pano, e_pers = res["eager_one_stream"]
        parity_check = {
            "what": "one denoising step at the benchmarked size from identical latents / RNG states: the replayed hipGraph (the timed "
                    "launch mode) against the same step issued eagerly on ONE stream; relative L2 of the updated latents",
            "graph_dual_vs_eager_rel": max(relf(g_pano, e_pano), relf(g_pers, e_pers)),
            "graph_dual_vs_eager_pano_rel": relf(g_pano, e_pano), "graph_dual_vs_eager_pers_rel": relf(g_pers, e_pers),
            "graph_dual_vs_eager_bit_identical": bool(torch.equal(g_pano, e_pano) and torch.equal(g_pers, e_pers)),
            "eager_two_streams_vs_one_bit_identical": bool(torch.equal(res["eager_two_streams"][0], e_pano) and torch.equal(res["eager_two_streams"][1], e_pers)),
            "eager_two_streams_vs_one_rel": max(relf(res["eager_two_streams"][0], e_pano), relf(res["eager_two_streams"][1], e_pers)),
            "step_changed_the_latents_rel": relf(g_pano, p0),
        }
        kernels.STATS, kernels.SHAPES = saved_counters
        parity_check["status"] = ("bit-identical" if parity_check["graph_dual_vs_eager_bit_identical"] else
                                  "DIFFERS from the eager one-stream step (every recorded run was bit-identical: investigate)")
        if parity_check["graph_dual_vs_eager_rel"] > 1e-3:      # (bit-identical in practice; hipBLASLt may pick another solution under capture)
            raise SystemExit(f"bench.py: the benchmarked launch mode disagrees with the eager single-stream step: {parity_check}")

    if rank == 0:
        boc = tuple(mv.unet.config.block_out_channels)
        total, parts = flops.step_flops(frames=frames, pano_hw=w["pano_hw"], pers_hw=w["pers_hw"],
                                        block_out_channels=boc, breakdown=True)
        samples = world if mode == "samples" else 1
        steps_per_s = samples * args.steps / elapsed
        out = {
            **({"plumbing_check": True, "valid": False, "backend": "gloo (CPU tensors; NOT a benchmark: rank / shard / collective plumbing only)"} if plumbing else {}),
            "metric": "denoising steps/sec (dual-branch UNet, 16x512x1024 latent)",
            "value": steps_per_s, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak" if mode == "samples" else "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "tuned_gemm_table": tuned, "tuned_gemm_table_status": (tuning.STATUS if not args.no_tuned_gemms else {"applied": False, "reason": "--no-tuned-gemms"}),
            "launch": ("eager, one stream" if graphed is None else "hipGraph replay (one captured step)") + (", panorama branch on a side stream between the WarpAttn calls" if mv.dual_stream and (shard is None or (mv.dual_stream_shard and mv._shard_two_comms)) and graphed is not None else ""),
            "config": {"workload": w["desc"],
                       "parallelism": {"samples": f"sample-parallel x{world}" if world > 1 else "single GPU",
                                       "frames": f"frame-chunk sharding x{world} ({frames // max(world, 1)} frames per GPU)",
                                       "cfgxframes": f"CFG halves x frame chunks (2 x {world // 2})"}[mode],
                       "width_div": args.width_div, "ddim_steps_schedule": nsteps_total, "guidance": guidance,
                       "tflop_per_step": total / 1e12, "outputs_finite": finite},
            **({"shard_boundary": shard.boundary,
                "shard_exchange": {"validated_on_rccl": "world size 1 only (tests/test_dist_gpu.py::test_exchange_path_through_rccl_at_world_size_one_eager_and_captured: "
                                                        "pack -> all_to_all_single -> temporal kernel -> return trip, all_gather, CFG pair exchange on RCCL device "
                                                        "buffers, eager and captured in a hipGraph); never on more than one GPU"},
                "dual_stream_shard": {"enabled": bool(mv.dual_stream_shard and mv._shard_two_comms),
                                      "validated_on_rccl": False,
                                      "note": "two communicators on two forked streams work eagerly through RCCL at world size 1, but ENDING THE CAPTURE of a "
                                              "graph that holds both segfaults (round 6, tests/test_dist_gpu.py::test_two_communicators_on_two_streams_in_one_captured_graph, "
                                              "xfail): with this flag use --no-graph; a number measured with it is UNVALIDATED"}}
               if shard is not None else {}),
            "parity_check": parity_check,
            "whole_step_tflops": total / 1e12 * steps_per_s / world,          # per GPU
            "whole_step_frac_of_mfma_peak": total / 1e12 * steps_per_s / world / MFMA_PEAK_TFLOPS,
        }
        if eager_elapsed is not None:
            prof = {k: kernels.prof_collect(k) for k in prof_kinds}
            stats, kernels.STATS = kernels.STATS, None
            shapes, kernels.SHAPES = kernels.SHAPES, None
            out["eager_ms_per_step"] = 1e3 * eager_elapsed / args.steps
            out["profile_source"] = ("HIP events (on the launch stream) around every launch of our kernels during a separate eager pass of "
                                     "the same K steps, after the timed region; algorithmic flops / bytes counted per launch"
                                     + ("; that pass issues everything on ONE stream so that a kernel's events bracket it alone, whereas in "
                                        "the timed region the panorama branch's kernels run beside the perspective branch's (side stream): "
                                        "the per-class durations therefore add up to more than ms_per_step" if was_dual else ""))
            classes = {}
            # the attention class = all three kinds of launches, as in every earlier round; the parts are kept for the breakdown below
            attn_parts = {}
            for sub in ("attn", "attn_warp", "attn_x2"):
                attn_parts[sub] = (prof.get(sub, (0.0, 0)), list(stats.get(sub, [0.0, 0.0, 0, 0.0])))
            prof["attn"] = (sum(v[0][0] for v in attn_parts.values()), sum(v[0][1] for v in attn_parts.values()))
            stats["attn"] = [sum(v[1][i] for v in attn_parts.values()) for i in range(4)]
            for sub in ("attn_warp", "attn_x2"):
                prof.pop(sub, None)
                stats.pop(sub, None)
            for kname, (ms, n) in prof.items():
                fl, by, _, fx = stats.get(kname, [0.0, 0.0, 0, 0.0])
                t = ms * 1e-3
                # utilisation is priced on EXECUTED flops (the sub-pixel upsample convolutions run 4 / 9 of the reference
                # algorithm's multiplies); the algorithmic figure is reported next to it
                t_mfma, t_hbm = fx / (MFMA_PEAK_TFLOPS * 1e12), by / (HBM_PEAK_GBS * 1e9)
                c = {"ms_per_step": ms / args.steps, "launches_per_step": n / args.steps,
                     "algorithmic_tflop_per_step": fl / args.steps / 1e12, "executed_tflop_per_step": fx / args.steps / 1e12,
                     "algorithmic_gb_per_step": by / args.steps / 1e9}
                if t > 0:
                    c.update({"tflops": fx / t / 1e12, "algorithmic_tflops": fl / t / 1e12,
                              "frac_of_mfma_peak": fx / t / 1e12 / MFMA_PEAK_TFLOPS,
                              "gbs": by / t / 1e9, "frac_of_hbm_peak": by / t / 1e9 / HBM_PEAK_GBS,
                              "bound": "mfma" if t_mfma >= t_hbm else "hbm", "roofline_frac": max(t_mfma, t_hbm) / t})
                classes[kname] = c
            out["kernels"] = classes
            # the roofline block: the class of our kernels that takes the most time per step (conv_igemm / conv_ring as
            # convolution, or the same kernels as token-major GEMMs)
            dom = max(("conv", "gemm"), key=lambda k: classes[k]["ms_per_step"])
            c = classes[dom]
            names = {"conv": "conv_igemm_kernel (GroupNorm+SiLU'd 3x3/1x1 convolutions, implicit GEMM)",
                     "gemm": "conv_ring_kernel / conv_igemm_kernel as token-major GEMMs (Linear + bias/residual, fused GEGLU)"}
            traffic, tnote = None, "no PMC summary found"
            if os.path.isfile(args.traffic_json):
                try:
                    # measured traffic / algorithmic bytes per SHAPE CLASS (tools/hbm_traffic.py, at the perspective branch's
                    # token counts), weighted with THIS step's launch mix: every launch of the class contributes its own
                    # algorithmic bytes x the measured ratio of its shape key
                    tj = json.load(open(args.traffic_json))
                    ratio = {v["key"]: v["traffic_over_algorithmic"] for v in tj.get(dom, {}).values()
                             if isinstance(v, dict) and v.get("key") and "traffic_over_algorithmic" in v}
                    mix = {k[1]: v for k, v in shapes.items() if k[0] == dom}
                    n_all = sum(v[0] for v in mix.values())
                    by_all = sum(v[1] for v in mix.values())
                    by_cov = sum(v[1] for k, v in mix.items() if k in ratio)
                    if ratio and n_all:
                        traffic = sum(v[1] * ratio.get(k, 1.0) for k, v in mix.items()) / n_all
                        tnote = (f"launch-mix weighted: sum over the step's {len(mix)} {dom} shape classes of (algorithmic bytes x measured "
                                 f"traffic / algorithmic ratio of that class) / launches; ratios from {os.path.relpath(args.traffic_json, ROOT)} "
                                 "(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2 fetch correction); "
                                 f"{100.0 * by_cov / max(by_all, 1.0):.0f} % of the class's algorithmic bytes are in measured shape classes "
                                 f"(the rest counted at ratio 1); algorithmic bytes per launch: {by_all / n_all:.3e}")
                except (ValueError, OSError, KeyError) as e:
                    tnote = f"unreadable PMC summary: {e}"
            out["roofline"] = {"bound": "mfma", "kernel": names[dom], "class": dom,
                               "achieved": c.get("tflops", 0.0), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": c.get("frac_of_mfma_peak", 0.0), "traffic": traffic, "traffic_note": tnote,
                               "launches_per_step": c["launches_per_step"],
                               "avg_launch_ms": c["ms_per_step"] / max(c["launches_per_step"], 1e-9),
                               "algorithmic_tflop_per_step": c["algorithmic_tflop_per_step"],
                               "hbm_view": {"achieved_gbs": c.get("gbs", 0.0), "frac_of_hbm_peak": c.get("frac_of_hbm_peak", 0.0),
                                            "class_bound": c.get("bound"), "roofline_frac": c.get("roofline_frac")}}
            # what the vendor's own GEMM sustains on THIS box under the power cap (outside the timed region; ~40 ms): the
            # datasheet peak above stays the `peak`, this is context for reading `frac`
            try:
                ga, gb = (torch.randn(8192, 8192, device=dev, dtype=dt) for _ in range(2))
                for _ in range(15):
                    torch.matmul(ga, gb)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    torch.matmul(ga, gb)
                e1.record()
                torch.cuda.synchronize()
                out["roofline"]["practical_ceiling"] = {
                    "tflops": 30 * 2.0 * 8192 ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12,
                    "what": "hipBLASLt 8192^3 GEMM of the same dtype, 30 launches back to back after the timed region "
                            "(the chip runs at its power cap; see profiles/r02_clock_probe.txt)"}
                del ga, gb
            except RuntimeError as e:
                out["roofline"]["practical_ceiling"] = {"tflops": None, "what": f"not measured: {e}"}
            a = classes["attn"]
            out["attention"] = {"achieved": a.get("tflops", 0.0), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": a.get("frac_of_mfma_peak", 0.0),
                                "algorithmic_tflop_per_step": a["algorithmic_tflop_per_step"],
                                "note": "QK^T + PV flops of every attn_fwd launch (self, text + IP cross, WarpAttn) over their summed time"}
            # the same, by kind of launch: the two-set text + IP cross attention (77 + 64 keys, K / V resident in LDS) streams Q and O and
            # is bound by HBM, not by the matrix pipe -- its own roofline is the HBM one; `mfma_bound_kinds` leaves it out
            by_kind = {}
            for sub, label in (("attn", "self + single-set cross attention (d = 64)"), ("attn_warp", "WarpAttn cross-view attention (d = 32, shared soft mask)"),
                               ("attn_x2", "text + IP-adapter cross attention (two key / value sets, 77 + 64 keys)")):
                (ms, n), (fl, by, _, fx) = attn_parts[sub]
                if n:
                    t = ms * 1e-3
                    by_kind[sub] = {"what": label, "ms_per_step": ms / args.steps, "launches_per_step": n / args.steps,
                                    "tflops": fl / t / 1e12, "frac_of_mfma_peak": fl / t / 1e12 / MFMA_PEAK_TFLOPS,
                                    "gbs": by / t / 1e9, "frac_of_hbm_peak": by / t / 1e9 / HBM_PEAK_GBS,
                                    "bound": "mfma" if fl / (MFMA_PEAK_TFLOPS * 1e12) >= by / (HBM_PEAK_GBS * 1e9) else "hbm"}
            out["attention"]["by_kind"] = by_kind
            mf = [attn_parts[k] for k in ("attn", "attn_warp") if attn_parts[k][0][1]]
            if mf:
                t = sum(v[0][0] for v in mf) * 1e-3
                fl = sum(v[1][0] for v in mf)
                out["attention"]["mfma_bound_kinds"] = {"tflops": fl / t / 1e12, "frac": fl / t / 1e12 / MFMA_PEAK_TFLOPS,
                                                        "note": "self + WarpAttn launches only (the kinds whose roofline is the matrix pipe); `frac` above stays the figure over ALL launches"}
        if world == 1 and args.cpu_baseline != "none":
            if args.cpu_baseline == "sample":
                out["cpu_baseline"] = cpu_baseline_sample(args)
            else:
                out["cpu_baseline"] = cpu_baseline_step(mv, args, workload=args.workload if args.cpu_baseline == "direct" else "cfg1", threads=args.cpu_threads)
            # `value` is a MEASURED step of the benchmarked workload.  The run's own bounded measurement (one cfg1 step, ~1 min)
            # x the analytic FLOP ratio over-states the CPU by 28 %; the direct measurement of the benchmarked cfg2 step on a box of this
            # pool (tools/cpu_baseline.py, ~9 min, committed) is the figure of record, the in-run extrapolation rides along beside it.
            direct = os.path.join(ROOT, "profiles", CPU_BASELINE_DIRECT)
            if os.path.isfile(direct) and args.workload == "cfg2" and args.cpu_baseline != "direct":
                try:
                    d = json.load(open(direct))
                    cb = out["cpu_baseline"]
                    cb["in_run_extrapolation"] = {k: cb[k] for k in ("value", "sample", "extrapolation", "cores") if k in cb}
                    cb.update(value=d["value"], cores=d["cores"], sample=d["sample"] + f" [measured once per round on a box of this pool: profiles/{CPU_BASELINE_DIRECT}]",
                              extrapolation=d["extrapolation"], measured_cfg2_s_per_step=d.get("measured_cfg2_s_per_step"))
                except (ValueError, OSError, KeyError):
                    pass
            out["speedup_vs_cpu_baseline"] = steps_per_s / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        if own_group:
            dist.destroy_process_group()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
