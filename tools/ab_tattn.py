#!/usr/bin/env python
"""Interleaved A/B of the temporal-attention kernel's non-temporal variants (knob tattn_nt: bit 0 stores, bit 1 loads) on cfg2's shapes.
    python tools/ab_tattn.py [--iters N] [--rounds R]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit, rn  # noqa: E402

iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
for name, B, Fr, P, C in [("pers L0", 40, 16, 1024, 320), ("pano L0", 2, 16, 8192, 320), ("pers L1", 40, 16, 256, 640), ("pano L1", 2, 16, 2048, 640),
                          ("pers L2", 40, 16, 64, 1280), ("pers L3", 40, 16, 16, 1280)]:
    qkv = rn(B * Fr * P, 3 * C)
    ref, same, best = None, {}, {}
    for v in (0, 1, 2, 3):
        K.tuning_set("tattn_nt", v)
        y = K.temporal_attention(qkv, B, Fr, P, 8).clone()
        same[v] = True if ref is None else torch.equal(ref, y)
        ref = y if ref is None else ref
        best[v] = float("inf")
    for _ in range(rounds):
        for v in (0, 1, 2, 3):
            K.tuning_set("tattn_nt", v)
            best[v] = min(best[v], timeit(lambda: K.temporal_attention(qkv, B, Fr, P, 8), iters))
    K.tuning_set("tattn_nt", 0)
    by = 4.0 * B * Fr * P * C * 2
    print(f"tattn {name:8s} " + " | ".join(f"nt {v}: {best[v] * 1e3:6.3f} ms {by / best[v] / 1e9:5.0f} GB/s{'' if same[v] else ' DIFFERS'}" for v in (0, 1, 2, 3)), flush=True)
