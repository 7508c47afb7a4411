#!/usr/bin/env python
"""Cycle stamps of the four-wave GEMM tile (library built with -DIM360_G4_ABL; knob conv_dbg 16 = stamps, 17 = stamps without the load /
write stream, 25 = stamps without stream and epilogue): shader-clock cycles per stage / epilogue of workgroup 0's wave 0."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import kernels as K  # noqa: E402
from tools.bench_kernels import rn  # noqa: E402

K.tuning_set("conv_ring", 12)
for name, M, C in [("geglu pers L0", 655360, 320), ("geglu pers L1", 163840, 640), ("geglu pers L2", 40960, 1280)]:
    x, w, b = rn(M, C), rn(8 * C, C) * C ** -0.5, rn(8 * C)
    wp, bp = K.pack_geglu(w, b)
    nph = C // 64
    for mode in (16, 17, 25):
        K.tuning_set("conv_dbg", mode)
        for _ in range(3):
            y = K.linear_geglu(x, wp, bp, 4 * C)
        torch.cuda.synchronize()
        st = y.view(torch.int32).flatten()[:256].cpu().tolist() if y.element_size() == 4 else y.flatten().view(torch.int32)[:256].cpu().tolist()
        st = [s & 0xffffffff for s in st if s != 0]
        d = [(b_ - a_) & 0xffffffff for a_, b_ in zip(st, st[1:])]
        per_tile = nph + 1 + (1 if nph % 2 else 0)          # stamps per tile: tile start (+ after the epilogue in the odd form), one per stage
        print(f"{name} mode {mode}: {len(st)} stamps; deltas of the first tiles:")
        for t in range(1, 4):
            seg = d[t * per_tile:(t + 1) * per_tile]
            print("    tile", t, seg)
    K.tuning_set("conv_dbg", 0)
    del x, w, b, wp, bp
K.tuning_set("conv_ring", 1)
