#!/usr/bin/env python
"""Benchmark of the dual-branch denoising hot path (BASELINE.json metric: denoising steps/sec).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one CFG-batched MultiViewBaseModel.forward (both UNets + 7 WarpAttn) + CFG combine + the two
DDIM updates, on synthetic inputs already resident in HBM.  N > 1 = sample-parallel
(one independent sample per GPU, weights replicated, BASELINE config 3): weak scaling, no per-step
collective, one all-gather of the final panorama latents at the latent boundary inside the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from imagine360_amd import configs, flops, kernels, synthetic  # noqa: E402
from imagine360_amd.scheduler import DDIMScheduler  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0       # dense bf16/fp16 MFMA peak of MI355X (MI355X_MICROARCH.md)
WORKLOADS = {
    "cfg2": dict(frames=16, pano_hw=(64, 128), pers_hw=(32, 32), pers_px=256,
                 desc="BASELINE cfg2: 16-frame 512x1024 equirect (pano latent 4x16x64x128 + 20 views 4x16x32x32), "
                      "CFG batch 2, full-width random-init UNets, DDIM step"),
    "cfg1": dict(frames=8, pano_hw=(32, 64), pers_hw=(16, 16), pers_px=128,
                 desc="BASELINE cfg1 shapes: 8-frame 256x512 equirect, CFG batch 2"),
    "cfg5": dict(frames=16, pano_hw=(128, 256), pers_hw=(64, 64), pers_px=512,
                 desc="BASELINE cfg5 shapes: 16-frame 1024x2048 equirect, CFG batch 2"),
}


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference path, fp32, all host threads) timed on a BOUNDED sample and scaled
    to the benchmarked workload by the analytic FLOP ratio.  Sample = the three block types that carry ~95 % of a
    step's FLOPs (ResnetBlock3D, spatial Transformer3DModel, motion module) at FULL channel width at UNet levels 0
    (320 ch, 32x32) and 2 (1280 ch, 8x8) on 8 frames of 12 views -- full width because CPU GEMM/conv efficiency depends
    on the channel widths, a small batch because a whole cfg2 step would take ~10 minutes of host time."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from im360_oracle import unet as OU
    from imagine360_amd.mv_model import MultiViewBaseModel
    from imagine360_amd.weights import filler_tensor
    cores = os.cpu_count() or 1
    # give the CPU its best shot: pick the thread count that maximises fp32 GEMM throughput on this host
    # (oversubscribing a 256-thread box makes the small per-frame ops of this path 3x slower)
    best, best_t = cores, 0.0
    a, bm = torch.randn(4096, 1280), torch.randn(1280, 1280)
    for nt in sorted({min(cores, c) for c in (16, 32, 64, 128, cores)}):
        torch.set_num_threads(nt)
        torch.mm(a, bm)
        t0 = time.time()
        for _ in range(5):
            torch.mm(a, bm)
        r = 1.0 / (time.time() - t0)
        if r > best_t:
            best, best_t = nt, r
    torch.set_num_threads(best)
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
            fl = 2 * (2.0 * c * c * 9 * n * frames)                                                  # two 3x3 convs
            fl += frames * n * (2.0 * c * c * 2 + 2.0 * c * c * 4 + 2.0 * c * c * 2 + 2.0 * c * 8 * c + 2.0 * 4 * c * c)   # spatial GEMMs
            fl += views * frames * ctx_n * 2.0 * 1024 * c * 2 + 4.0 * n * ctx_n * c * frames + 4.0 * n * (h * w) * c * frames     # cross K/V, cross, self
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
    return {"value": cpu_tflops / full_tf, "unit": "denoising steps/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle ResnetBlock3D + spatial transformer + motion module, full width, fp32, 8 frames of 12 views ("
                      + "; ".join(parts) + f") = {cpu_tflops:.3f} TFLOP/s on {torch.get_num_threads()} host threads; scaled to the "
                      f"{full_tf:.1f} TF step of {args.workload} by the analytic FLOP ratio"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--width-div", type=int, default=1, help="debug only: reduced-width model (INVALID as a benchmark)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tuned-gemms", action="store_true", help="hipBLASLt default heuristic instead of the shipped solution table")
    ap.add_argument("--no-graph", action="store_true", help="issue every kernel from Python instead of replaying a hipGraph")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    w = WORKLOADS[args.workload]
    torch.set_grad_enabled(False)
    kernels.lib()
    tuned = False
    if not args.no_tuned_gemms:
        from imagine360_amd import tuning
        tuned = tuning.enable()
    mv = configs.build_mv_model(args.width_div, device=dev, dtype=dt, xformers=True)
    inp = synthetic.mv_inputs(frames=w["frames"], pano_hw=w["pano_hw"], pers_hw=w["pers_hw"], seed=1 + rank,
                              sam_frames=max(16, w["frames"]), dtype=dt, device=dev)
    cams = synthetic.icosahedron_cameras(90, w["pers_px"], device=dev)
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
        pano_lat = sch.fused_cfg_step(pred_pano[0:1], pred_pano[1:2], guidance, ts_host[t], pano_lat)
        pers_lat = sch.fused_cfg_step(pred_pers[0:1], pred_pers[1:2], guidance, ts_host[t], pers_lat)

    step = eager_step

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    prof_kinds = ["conv", "gemm", "attn", "temporal", "gn_stats", "gn_apply", "misc"]
    graphed = None
    if not args.no_graph:
        # the step is ~3000 launches: capture it once (hipGraph) and replay, so the host is out of the loop
        from imagine360_amd.graph_step import GraphedDenoiseStep
        try:
            graphed = GraphedDenoiseStep(mv, sch, inp, cams, pano_lat, pers_lat, guidance, warmup=1)
        except RuntimeError as e:           # capture refused (e.g. by another runtime thread): time the eager issue instead
            print(f"[rank {rank}] hipGraph capture failed, falling back to eager launches: {e}", file=sys.stderr)
            graphed = None
            torch.cuda.synchronize()
    if graphed is not None:
        def step(i):                                                       # noqa: F811
            graphed.step(ts_host[i % nsteps_total])

    for i in range(args.warmup):
        step(i)
    if dist is not None:          # untimed: first use of the collective (RCCL channel setup is lazy)
        warm = [torch.empty_like(pano_lat) for _ in range(world)]
        dist.all_gather(warm, pano_lat if graphed is None else graphed.pano_lat)
        del warm
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if graphed is not None:
        pano_lat = graphed.pano_lat
    if dist is not None:          # latent boundary: gather every rank's panorama latent (1 MB each)
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
    if rank == 0:
        if graphed is not None:
            pano_lat, pers_lat = graphed.pano_lat.clone(), graphed.pers_lat.clone()

            def step(i):                                                   # noqa: F811
                eager_step(i)
        kernels.prof_enable(prof_kinds)
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        torch.cuda.synchronize()
        eager_elapsed = time.perf_counter() - t1
        kernels.prof_enable([])
    finite = bool(torch.isfinite(pano_lat.float()).all().item())

    if rank == 0:
        prof = {k: kernels.prof_collect(k) for k in prof_kinds}
        boc = tuple(mv.unet.config.block_out_channels)
        total, parts = flops.step_flops(frames=w["frames"], pano_hw=w["pano_hw"], pers_hw=w["pers_hw"],
                                        block_out_channels=boc, breakdown=True)
        conv_flops = parts["pers conv"] + parts["pano conv"]
        attn_flops = parts["pers self-attn"] + parts["pano self-attn"] + parts["WarpAttn attn"]
        conv_ms, conv_n = prof["conv"]
        attn_ms, attn_n = prof["attn"]
        conv_tfs = conv_flops * args.steps / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        attn_tfs = attn_flops * args.steps / (attn_ms * 1e-3) / 1e12 if attn_ms > 0 else 0.0
        steps_per_s = world * args.steps / elapsed
        out = {
            "metric": "denoising steps/sec (dual-branch UNet, 16x512x1024 latent)",
            "value": steps_per_s, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "tuned_gemm_table": tuned,
            "launch": "eager" if graphed is None else "hipGraph replay (one captured step)",
            "eager_ms_per_step": 1e3 * eager_elapsed / args.steps,
            "config": {"workload": w["desc"], "parallelism": f"sample-parallel x{world}" if world > 1 else "single GPU",
                       "width_div": args.width_div, "ddim_steps_schedule": nsteps_total, "guidance": guidance,
                       "tflop_per_step": total / 1e12, "outputs_finite": finite},
            "roofline": {"bound": "mfma", "kernel": "conv_igemm_kernel (GroupNorm+SiLU'd 3x3/1x1 conv, implicit GEMM)",
                         "achieved": conv_tfs, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": conv_tfs / MFMA_PEAK_TFLOPS, "traffic": None,
                         "launches_per_step": conv_n / max(args.steps, 1), "avg_launch_ms": conv_ms / max(conv_n, 1),
                         "algorithmic_tflop_per_step": conv_flops / 1e12},
            "kernels": {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[1] / args.steps} for k, v in prof.items()},
            "attention": {"achieved": attn_tfs, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": attn_tfs / MFMA_PEAK_TFLOPS,
                          "algorithmic_tflop_per_step": attn_flops / 1e12,
                          "note": "self-attention + WarpAttn QK^T/PV flops over all attn_fwd launches (cross-attention launches "
                                  "included in the time, their small flops not counted)"},
            "whole_step_tflops": total / 1e12 * steps_per_s / world,
            "whole_step_frac_of_mfma_peak": total / 1e12 * steps_per_s / world / MFMA_PEAK_TFLOPS,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
            out["speedup_vs_cpu_baseline"] = steps_per_s / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
