#!/usr/bin/env python
"""Interleaved A/B of the GroupNorm paths on the image shapes of cfg2's step: rounds 2 - 5's statistics + finalize + apply
("three") against the one-launch normalisation from partial sums ("partials"; knob gn_apply = loads in flight / non-temporal
stores, knob gn_wgs = workgroups of the normalisation launch), with and without the producer's partial sums standing in for the statistics pass.  Time per call (minimum over rounds,
variants alternated), achieved GB/s on the algorithmic bytes of the normalisation pass (x read + y written), bit identity.
    python tools/ab_gn.py [--iters N] [--rounds R]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit, rn  # noqa: E402

iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3

# (name, N, H, W, C1, C2, pad, silu): resnet norm1 / norm2, the transformers' norm, the decoder's skip pairs
shapes = [("pers L0 resnet", 640, 32, 32, 320, 0, 0, True), ("pano L0 resnet pad", 32, 64, 128, 320, 0, 2, True), ("pano L0 norm", 32, 64, 128, 320, 0, 0, False),
          ("pers L1 resnet", 640, 16, 16, 640, 0, 0, True), ("pano L1 resnet pad", 32, 32, 64, 640, 0, 2, True), ("pers L2 resnet", 640, 8, 8, 1280, 0, 0, True),
          ("pers L0 skip pair", 640, 32, 32, 320, 320, 0, True), ("pers L1 skip pair", 640, 16, 16, 640, 640, 0, True), ("pano L0 skip pair pad", 32, 64, 128, 320, 320, 2, True),
          ("pers L3 resnet", 640, 4, 4, 1280, 0, 0, True)]
# (mode, knob gn_apply, knob gn_wgs)
variants = [("three", None, 0), ("partials", 0, 0), ("partials", 2, 0), ("partials", 2, 1024), ("partials", 2, 640), ("partials", 2, 512), ("partials", 3, 640), ("partials", 2, 256)]
for name, N, H, W, C1, C2, pad, silu in shapes:
    xa = rn(N, H, W, C1)
    xb = rn(N, H, W, C2) if C2 else None
    C = C1 + C2
    gamma, beta = 1 + 0.1 * rn(C), 0.1 * rn(C)
    x = xa if xb is None else (xa, xb)
    for tagged in (False, True):
        if tagged:
            if pad:
                continue
            # partial sums as a producer's epilogue leaves them: one slab per 256 pixels (H W % 256 == 0), else the statistics kernel's
            for t in (xa, xb):
                if t is None:
                    continue
                S = H * W // 256 if (H * W) % 256 == 0 else K.lib().im360_gn_num_slabs(N, H, W)
                if S == K.lib().im360_gn_num_slabs(N, H, W):
                    buf, _ = K.group_norm_partials(t)
                else:
                    buf = torch.stack([t.float().reshape(N, S, -1, t.shape[-1]).sum(2), (t.float() ** 2).reshape(N, S, -1, t.shape[-1]).sum(2)], dim=2).reshape(-1).contiguous()
                K._tag_gn(t, buf, S)

        def fn(mode, v, wgs):
            K.GN_MODE = mode
            if v is not None:
                K.tuning_set("gn_apply", v)
            K.tuning_set("gn_wgs", wgs)
            return K.group_norm(x, gamma, beta, 32, 1e-5, silu=silu, pad=pad)

        ref, same = None, {}
        for var in variants:
            y = fn(*var).clone()
            same[var] = True if ref is None else (torch.equal(ref, y) if xb is None else float((ref.float() - y.float()).abs().max()) < 0.07)
            ref = y if ref is None else ref
            timeit(lambda: fn(*var), 3)
        best = {k: float("inf") for k in variants}
        for _ in range(rounds):
            for var in variants:
                best[var] = min(best[var], timeit(lambda: fn(*var), iters))
        K.GN_MODE = "partials"
        K.tuning_set("gn_apply", 2)
        K.tuning_set("gn_wgs", 0)
        nbytes = 2.0 * N * H * (W + W + 2 * pad) * C
        print(f"{name:24s} {'producer sums' if tagged else 'own statistics':14s} " +
              " | ".join(f"{m[:4]}{'' if v is None else v}/{w}: {best[(m, v, w)] * 1e3:6.3f} {nbytes / best[(m, v, w)] / 1e9:5.0f}{'' if same[(m, v, w)] else ' DIFFERS'}" for m, v, w in variants), flush=True)
        for t in (xa, xb):
            if t is not None and hasattr(t, "_im360_gn"):
                del t._im360_gn
    del xa, xb, x
