#!/usr/bin/env python
"""End-to-end wall time of AnimationPipeline.__call__ at BASELINE cfg2 on ONE MI355X: full-width random-init UNets and VAE,
synthetic video batch and conditioning (the encoders are out of scope), 25 DDIM steps, CFG 7.5, decode to 512 x 1024
frames.  Prints the phases (HIP-synchronised) and the total; the second call shows the steady state (graph captured,
weights packed, geometry cached).

    python tools/pipeline_e2e.py [--steps 25] [--out gpurun_out/pipeline_e2e.txt]
"""
import argparse
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import configs, kernels, synthetic as S          # noqa: E402
from imagine360_amd.pipeline import AnimationPipeline                # noqa: E402
from imagine360_amd.scheduler import DDIMScheduler                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--out", default="gpurun_out/pipeline_e2e.txt")
    args = ap.parse_args()
    dev, dt = torch.device("cuda", 0), torch.bfloat16
    torch.set_grad_enabled(False)
    kernels.lib()
    from imagine360_amd import tuning
    tuning.enable()
    t0 = time.time()
    mv = configs.build_mv_model(1, device=dev, dtype=dt, xformers=True)
    vae = configs.build_vae(1, device=dev, dtype=dt)
    pipe = AnimationPipeline(vae, None, None, mv.unet, mv.pano_unet, mv, DDIMScheduler(**configs.NOISE_SCHEDULER_KWARGS), None, "SAM").to(dev)
    pipe._no_progress = True
    torch.cuda.synchronize()
    lines = [f"model build (random init, full width): {time.time() - t0:.1f} s"]
    # phase timers (HIP-synchronised wrappers around the pipeline's own methods)
    phases = {}

    def timed(obj, name, label):
        fn = getattr(obj, name)

        def wrap(*a, **k):
            torch.cuda.synchronize()
            t = time.time()
            r = fn(*a, **k)
            torch.cuda.synchronize()
            phases[label] = phases.get(label, 0.0) + time.time() - t
            return r
        setattr(obj, name, wrap)
    from imagine360_amd import graph_step
    timed(graph_step.GraphedDenoiseStep, "__init__", "graph warm-up + capture")
    timed(graph_step.GraphedDenoiseStep, "step", "25 graph replays")
    timed(pipe, "init_noise", "init_noise")
    timed(pipe, "_encode_chunks", "VAE encode")
    timed(pipe, "decode_latents", "VAE decode")
    vb = S.video_batch(frames=16, pano_hw=(512, 1024), seed=0)
    cond = S.conditioning(frames=16, seed=0)
    for call in range(2):
        torch.manual_seed(21)
        random.seed(21)
        torch.cuda.synchronize()
        t0 = time.time()
        vid = pipe("synthetic", num_inference_steps=args.steps, guidance_scale_text=7.5, negative_prompt="", latents_dtype=dt,
                   video_batch=vb, use_outpaint=True, use_ip_plus_cross_attention=True, use_fps_condition=True,
                   ip_plus_condition="video", prompt_embeds=(cond["text_pano"], cond["text_pers"]),
                   sam_features=(cond["sam_pano"], cond["sam_pers"])).videos
        torch.cuda.synchronize()
        dt_call = time.time() - t0
        ph = ", ".join(f"{k} {v:.2f} s" for k, v in phases.items())
        phases.clear()
        ok = bool(torch.isfinite(vid).all())
        lines.append(f"call {call}: {args.steps} steps + VAE encode of the panorama and 320 views + decode -> video {tuple(vid.shape)} "
                     f"finite={ok} in {dt_call:.2f} s ({dt_call / args.steps * 1e3:.0f} ms per step all-in); "
                     f"peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB; of which {ph}")
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
