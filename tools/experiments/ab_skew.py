#!/usr/bin/env python
"""Phase skew of the two-workgroups-per-CU GEGLU kernel (gemm_g4b_kernel, knob g4 bits 2 / 3; knob g4_skew = start delay of the second
half of the grid in 64-cycle units): the fused GEGLU projections of cfg2, plain and with the LayerNorm folded in, against the default
eight-wave kernel; interleaved timing, minimum over rounds, bit identity.
    python tools/ab_skew.py [--iters N] [--rounds R] [--skews 0,40,80,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit, rn  # noqa: E402

iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
skews = [int(v) for v in sys.argv[sys.argv.index("--skews") + 1].split(",")] if "--skews" in sys.argv else [0, 30, 60, 90, 120, 160, 220, 300]
for name, M, C in [("pers L0", 655360, 320), ("pano L0", 262144, 320), ("pers L1", 163840, 640), ("pano L1", 65536, 640)]:
    x, w, b = rn(M, C), rn(8 * C, C) * C ** -0.5, rn(8 * C)
    g, be = 1 + 0.1 * rn(C), 0.1 * rn(C)
    wp, bp = K.pack_geglu(w, b)
    wg, c1, c2 = K.fold_layer_norm(w, b, g, be)
    wpl, c1p = K.pack_geglu(wg, c1)
    c1p, c2p = c1p.contiguous(), K.interleave_geglu(wg, c2)[1].contiguous()
    # the rows' LayerNorm statistics as a producing GEMM's epilogue leaves them: (sum, sum of squares) per 160-column slice
    xs = x.float()
    p = C // 160
    stats = torch.stack([xs.reshape(M, p, 160).sum(-1), (xs * xs).reshape(M, p, 160).sum(-1)], dim=-1).contiguous()
    del xs
    fl = 2.0 * M * C * 8 * C
    for ln in (False, True):
        fn = (lambda: K.linear_geglu_ln(x, wpl, c1p, c2p, stats, 1e-5, 4 * C)) if ln else (lambda: K.linear_geglu(x, wp, bp, 4 * C))
        bit = 8 if ln else 4
        variants = [("default", 0, 0)] + [(f"g4b skew {s}", bit, s) for s in skews]
        ref, same, best = None, {}, {}
        for v in variants:
            K.tuning_set("g4", v[1])
            K.tuning_set("g4_skew", v[2])
            y = fn().clone()
            same[v] = True if ref is None else (torch.equal(ref, y) or float((ref.float() - y.float()).abs().max() / ref.float().abs().max()) < 1e-2)
            ref = y if ref is None else ref
            best[v] = float("inf")
        for _ in range(rounds):
            for v in variants:
                K.tuning_set("g4", v[1])
                K.tuning_set("g4_skew", v[2])
                best[v] = min(best[v], timeit(fn, iters))
        K.tuning_set("g4", 0)
        K.tuning_set("g4_skew", 0)
        print(f"geglu {name}{' +LN' if ln else '    '} " + " | ".join(f"{v[0].replace('g4b skew ', 's')}: {best[v] * 1e3:6.3f} {fl / best[v] / 1e12:5.0f}{'' if same[v] else ' DIFFERS'}" for v in variants), flush=True)
    del x, w, b, wp, bp, wpl, stats
