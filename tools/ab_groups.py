#!/usr/bin/env python
"""A/B of the persistent tile walk's cout groups (knob ring_groups: 0 = the launcher's rule, 1 / 2 / 4 / 8 forced) on the
token-major GEMM shapes whose weights exceed an XCD's L2 share: time per launch; same bits whatever the walk.
    python tools/ab_groups.py [--iters N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit, rn  # noqa: E402

iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
GROUPS = (0, 1, 2, 4, 8)


def run(name, fn, fl, rounds=4):
    """Round 5: the first variant timed after an idle gap runs at ramping clocks (consecutive 10-launch timings of the SAME
    configuration drifted by 16 % on one box), so the variants are timed INTERLEAVED, `rounds` times over, after a warm-up of
    the whole set; the minimum per variant is reported."""
    ref, same = None, {}
    for g in GROUPS:
        K.tuning_set("ring_groups", g)
        y = fn()
        y = (y[0] if isinstance(y, tuple) else y).clone()
        same[g] = True if ref is None else torch.equal(ref, y)
        ref = y if ref is None else ref
        timeit(fn, iters)
    best = {g: float("inf") for g in GROUPS}
    for _ in range(rounds):
        for g in GROUPS:
            K.tuning_set("ring_groups", g)
            best[g] = min(best[g], timeit(fn, iters))
    K.tuning_set("ring_groups", 0)
    print(f"{name:30s} " + " | ".join(f"g{g}: {best[g] * 1e3:6.3f} ms {fl / best[g] / 1e12:5.0f} TF/s{'' if same[g] else ' DIFFERS'}" for g in GROUPS), flush=True)


for name, M, Kd, N in [("pers L1 qkv", 163840, 640, 1920), ("pers L1 ff-out", 163840, 2560, 640), ("pers L1 out-proj", 163840, 640, 640),
                       ("pano L1 qkv", 65536, 640, 1920), ("pers L0 qkv", 655360, 320, 960), ("pers L0 ff-out", 655360, 1280, 320)]:
    x, w, b, r = rn(M, Kd), rn(N, Kd) * Kd ** -0.5, rn(N), rn(M, N)
    wp = K.pack_conv_weight(w.reshape(N, Kd, 1, 1))
    run("linear+res " + name, lambda: K.linear(x, wp, N, bias=b, res=r), 2.0 * M * Kd * N)
    del x, w, b, r, wp
for name, M, C in [("pers L1", 163840, 640), ("pano L1", 65536, 640), ("pers L0", 655360, 320), ("pano L0", 262144, 320)]:
    x, w, b = rn(M, C), rn(8 * C, C) * C ** -0.5, rn(8 * C)
    wp, bp = K.pack_geglu(w, b)
    run("geglu " + name, lambda: K.linear_geglu(x, wp, bp, 4 * C), 2.0 * M * C * 8 * C)
    del x, w, b, wp, bp
