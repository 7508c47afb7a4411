#!/usr/bin/env python
"""Ablation of the four-wave GEMM tile's fused-GEGLU kernel (library built with -DIM360_G4_ABL; knob conv_dbg bits: 1 no load / write
stream, 2 no MFMA, 4 no fragment reads, 8 no epilogue): where a tile's time goes.  python tools/g4_ablate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit, rn  # noqa: E402

NAMES = {0: "full", 8: "no epilogue", 1: "no stream", 2: "no MFMA", 9: "MFMA + reads only", 10: "stream + reads only", 7: "epilogue + barriers only", 15: "skeleton"}
K.tuning_set("conv_ring", 12)
for name, M, C in [("geglu pers L0", 655360, 320), ("geglu pers L1", 163840, 640), ("geglu pers L2", 40960, 1280)]:
    x, w, b = rn(M, C), rn(8 * C, C) * C ** -0.5, rn(8 * C)
    wp, bp = K.pack_geglu(w, b)
    fn = lambda: K.linear_geglu(x, wp, bp, 4 * C)
    ideal = 2.0 * M * C * 8 * C / 2.5e15 * 1e3
    best = {v: float("inf") for v in NAMES}
    for v in NAMES:
        K.tuning_set("conv_dbg", v)
        timeit(fn, 3)
    for _ in range(3):
        for v in NAMES:
            K.tuning_set("conv_dbg", v)
            best[v] = min(best[v], timeit(fn, 10))
    K.tuning_set("conv_dbg", 0)
    print(f"{name} (ideal MFMA {ideal:.3f} ms): " + " | ".join(f"{NAMES[v]} {best[v] * 1e3:.3f}" for v in NAMES), flush=True)
    del x, w, b, wp, bp
K.tuning_set("conv_ring", 1)
