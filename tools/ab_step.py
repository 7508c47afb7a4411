#!/usr/bin/env python
"""Interleaved A/B of WHOLE cfg2 denoising steps (the benchmarked launch mode: one captured hipGraph, panorama branch on a side
stream) under different tuning knobs / host settings: one model, one graph per variant (the knobs are baked in at capture), the
variants' replays alternated round by round so that clock ramps and box noise hit all of them alike; minimum and median ms per step,
and whether the updated latents are bit-identical to the first variant's.
    python tools/ab_step.py "base" "nt=1" "tattn_nt=1" "nt=1,tattn_nt=3" "GN_MODE=three" [--steps 6] [--rounds 4]
A variant is a comma-separated list of knob=value (kernels.KNOBS) or NAME=value for an upper-case attribute of imagine360_amd.kernels
/ imagine360_amd.layers (GN_MODE=three, ROUTE_MIN_TOKENS=32768, ...), or LIB=<path of another build of libim360_kernels.so>."""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imagine360_amd import configs, kernels, layers, synthetic, tuning  # noqa: E402
from imagine360_amd.graph_step import GraphedDenoiseStep  # noqa: E402
from imagine360_amd.scheduler import DDIMScheduler  # noqa: E402


def parse(spec):
    out = []
    for item in (spec.split(",") if spec not in ("base", "") else []):
        k, v = item.split("=")
        out.append((k, v))
    return out


def apply(settings, restore=None):
    """Set the variant's knobs / attributes; returns what to call to get the defaults back."""
    undo = []
    for k, v in settings:
        if k == "LIB":
            # another build of the kernel library (e.g. the previous commit's, kept under tools/experiments/): both are loaded side by
            # side; the variant's graph is captured with this one's kernels
            undo.append(("lib", kernels._lib, kernels._LIB_PATH))
            kernels._lib, kernels._LIB_PATH = None, os.path.abspath(v)
            kernels.lib()
        elif k in kernels.KNOBS:
            undo.append(("knob", k))
            kernels.tuning_set(k, int(v))
        else:
            mod = kernels if hasattr(kernels, k) else layers
            old = getattr(mod, k)
            undo.append(("attr", mod, k, old))
            setattr(mod, k, type(old)(v) if not isinstance(old, str) else v)
    return undo


DEFAULT_KNOBS = {}


def revert(undo):
    for u in undo:
        if u[0] == "lib":
            kernels._lib, kernels._LIB_PATH = u[1], u[2]
        elif u[0] == "knob":
            kernels.tuning_set(u[1], DEFAULT_KNOBS[u[1]])
        else:
            setattr(u[1], u[2], u[3])


def main():
    args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] not in ("--steps", "--rounds")]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 6
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 4
    variants = args or ["base"]
    dev, dt = torch.device("cuda", 0), torch.bfloat16
    torch.set_grad_enabled(False)
    kernels.lib()
    # the library's defaults (environment included) of every knob a variant touches
    import re
    src = open(os.path.join(os.path.dirname(os.path.abspath(kernels.__file__)), "csrc", "abi.cpp")).read()
    for name, env, dflt in re.findall(r"\{im360::KNOB_(\w+), \"(\w+)\", (-?\d+)\}", src):
        DEFAULT_KNOBS[name.lower()] = int(os.environ.get(env, dflt))
    tuning.enable()
    w = bench.WORKLOADS["cfg2"]
    mv = configs.build_mv_model(1, device=dev, dtype=dt, xformers=True)
    mv.dual_stream, mv.warp_streams = True, True
    inp = synthetic.mv_inputs(frames=w["frames"], pano_hw=w["pano_hw"], pers_hw=w["pers_hw"], seed=1, sam_frames=16, dtype=dt, device=dev)
    cams = synthetic.icosahedron_cameras(90, w["pers_px"], device=dev)
    sch = DDIMScheduler(**configs.NOISE_SCHEDULER_KWARGS)
    sch.set_timesteps(25)
    ts_host = [int(t) for t in sch._timesteps_host]
    pano_lat = inp["pano_latent"][:1, :4].contiguous()
    pers_lat = inp["latents"][:1, :, :4].contiguous()
    graphs, finals = [], []
    for spec in variants:
        undo = apply(parse(spec))
        try:
            gs = GraphedDenoiseStep(mv, sch, inp, cams, pano_lat, pers_lat, 7.5, warmup=1)
        finally:
            revert(undo)
        graphs.append(gs)
        torch.cuda.synchronize()
        print(f"captured: {spec}", file=sys.stderr, flush=True)
    times = [[] for _ in variants]
    for r in range(rounds + 1):
        for vi, gs in enumerate(graphs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                gs.step(ts_host[i % len(ts_host)])
            torch.cuda.synchronize()
            if r:                               # round 0 warms every graph up
                times[vi].append((time.perf_counter() - t0) / steps * 1e3)
    # identical starting latents, RNG state and coins -> compare one more step of each
    for gs in graphs:
        gs.pano_lat.copy_(pano_lat)
        gs.pers_lat.copy_(pers_lat)
        import random
        random.seed(3)
        torch.cuda.manual_seed(3)
        gs.step(ts_host[0])
        finals.append((gs.pano_lat.clone(), gs.pers_lat.clone()))
    for vi, spec in enumerate(variants):
        same = torch.equal(finals[vi][0], finals[0][0]) and torch.equal(finals[vi][1], finals[0][1])
        rel = float((finals[vi][1].float() - finals[0][1].float()).norm() / finals[0][1].float().norm())
        print(f"{spec:40s} min {min(times[vi]):8.2f} ms  median {statistics.median(times[vi]):8.2f} ms  ({len(times[vi])} rounds x {steps} steps)"
              f"  {'bit-identical to the first' if same else f'differs from the first: rel {rel:.2e}'}", flush=True)


if __name__ == "__main__":
    main()
