#!/usr/bin/env python
"""Interleaved A/B of ONE tuning knob on the token-major GEMM / fused-GEGLU shapes of cfg2: time per launch (minimum over
rounds, variants alternated so that clock ramps hit all of them alike), TF/s, and bit identity against the first value.
    python tools/ab_knob.py conv_pf 0,1,2,5 [--iters N] [--rounds R] [--blas] [--only substr]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit, rn  # noqa: E402

knob = sys.argv[1]
values = [int(v) for v in sys.argv[2].split(",")]
iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""


def run(name, fn, fl):
    if only and only not in name:
        return
    ref, same = None, {}
    for v in values:
        K.tuning_set(knob, v)
        y = fn()
        y = (y[0] if isinstance(y, tuple) else y).clone()
        same[v] = True if ref is None else torch.equal(ref, y)
        ref = y if ref is None else ref
        timeit(fn, 3)
    best = {v: float("inf") for v in values}
    for _ in range(rounds):
        for v in values:
            K.tuning_set(knob, v)
            best[v] = min(best[v], timeit(fn, iters))
    K.tuning_set(knob, values[0])
    print(f"{name:30s} " + " | ".join(f"{v}: {best[v] * 1e3:6.3f} ms {fl / best[v] / 1e12:5.0f} TF/s{'' if same[v] else ' DIFFERS'}" for v in values), flush=True)


shapes = [("pers L0 qkv", 655360, 320, 960), ("pers L0 ff-out", 655360, 1280, 320), ("pers L0 proj", 655360, 320, 320),
          ("pers L1 qkv", 163840, 640, 1920), ("pers L1 ff-in", 163840, 640, 5120), ("pers L1 ff-out", 163840, 2560, 640),
          ("pers L2 ff-in", 40960, 1280, 10240), ("pers L2 ff-out", 40960, 5120, 1280), ("pano L0 ff-in", 262144, 320, 2560),
          ("pano L1 ff-out", 65536, 2560, 640)]
for name, M, Kd, N in shapes:
    if only and only not in name:
        continue
    x, w, b, r = rn(M, Kd), rn(N, Kd) * Kd ** -0.5, rn(N), rn(M, N)
    wp = K.pack_conv_weight(w.reshape(N, Kd, 1, 1))
    fl = 2.0 * M * Kd * N
    if "--blas" in sys.argv:
        t0 = timeit(lambda: F.linear(x, w, b), iters)
        print(f"{name:30s} hipBLASLt {t0 * 1e3:6.3f} ms {fl / t0 / 1e12:5.0f} TF/s")
    run("linear+res " + name, lambda: K.linear(x, wp, N, bias=b, res=r), fl)
    run("linear nores " + name, lambda: K.linear(x, wp, N, bias=b), fl)
    del x, w, b, r, wp
for name, M, C in [("pers L0", 655360, 320), ("pano L0", 262144, 320), ("pers L1", 163840, 640), ("pers L2", 40960, 1280)]:
    x, w, b = rn(M, C), rn(8 * C, C) * C ** -0.5, rn(8 * C)
    wp, bp = K.pack_geglu(w, b)
    run("geglu " + name, lambda: K.linear_geglu(x, wp, bp, 4 * C), 2.0 * M * C * 8 * C)
    del x, w, b, wp, bp
